# reader-thread / block-size sweep of the PE compress through the driver.  usage: bash tools/e2e_sweep2.sh
set -e
cd $GRAFT_REPO_ROOT
D=/dev/shm/e2e; mkdir -p $D
./tools/fqgen --profile 1 --reads 11200000 --seed 3 -o $D/r1.fq -O $D/r2.fq
B=repaq_amd/bin/repaq_hip
for f in "" "--io_threads 16" "--io_threads 32" "--io_threads 16 --block_mb 8" "--io_threads 16 --block_mb 32" "--io_threads 24 --batch_mb 512" "--io_threads 16 --batch_mb 128"; do
  for i in 1 2; do TIMEFORMAT="PE compress [$f] wall %R s"; time $B -c -i $D/r1.fq -I $D/r2.fq -o $D/pe.rfq $f; done
  $B -c -i $D/r1.fq -I $D/r2.fq -o $D/pe.rfq --trace $f 2>&1 | grep "pipeline up\|all batches\|done" | tr '\n' ' '; echo
done
rm -rf /dev/shm/e2e
