cd $GRAFT_REPO_ROOT; D=/dev/shm/e2e; mkdir -p $D
./tools/fqgen --profile 0 --reads 11200000 --seed 5 -o $D/a.fq
B=repaq_amd/bin/repaq_hip
for mb in 16 32 64 128 256; do
  TIMEFORMAT="compress batch_mb=$mb wall %R s"; time $B -c -i $D/a.fq -o $D/a.rfq --batch_mb $mb
  TIMEFORMAT="decompress batch_mb=$mb wall %R s"; time $B -d -i $D/a.rfq -o $D/b.fq --batch_mb $mb
done
cmp $D/a.fq $D/b.fq && echo OK
rm -rf $D
