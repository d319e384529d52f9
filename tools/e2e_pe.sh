# configs[2] through the driver with phase marks: PE150 2 x 4 GB (-i/-I), files on /dev/shm.  One-shot runs (process start-up included) and the same jobs in ONE resident
# process (--serve: the HIP runtime is up, the device's context keeps its workspace).  usage: bash tools/e2e_pe.sh
set -e
cd $GRAFT_REPO_ROOT
D=/dev/shm/e2e; mkdir -p $D
./tools/fqgen --profile 1 --reads 11200000 --seed 3 -o $D/r1.fq -O $D/r2.fq
B=repaq_amd/bin/repaq_hip
for i in 1 2 3; do TIMEFORMAT="PE compress wall %R s (one-shot process)"; time $B -c -i $D/r1.fq -I $D/r2.fq -o $D/pe.rfq ${E2E_FLAGS:-}; done
$B -c -i $D/r1.fq -I $D/r2.fq -o $D/pe.rfq --trace ${E2E_FLAGS:-} 2>&1 | grep -v "batch resident\|batch encoded" | head -20
md5sum $D/pe.rfq
for i in 1 2; do TIMEFORMAT="PE decompress to /dev/null wall %R s (one-shot process)"; time $B -d -i $D/pe.rfq -o /dev/null -O /dev/null; done
for i in 1 2; do rm -f $D/o1.fq $D/o2.fq; TIMEFORMAT="PE decompress to two tmpfs files wall %R s (one-shot process)"; time $B -d -i $D/pe.rfq -o $D/o1.fq -O $D/o2.fq; done
$B -d -i $D/pe.rfq -o $D/o1.fq -O $D/o2.fq --trace 2>&1 | grep -v "batch resident\|batch decoded" | head
cmp $D/r1.fq $D/o1.fq && cmp $D/r2.fq $D/o2.fq && echo PE_ROUNDTRIP_OK
# the resident process: job 1 pays the start-up, the others are warm
echo "== --serve: compress x3, decompress to tmpfs x2, decompress to /dev/null (one process)"
rm -f $D/o1.fq $D/o2.fq
printf '%s\n' "-c -i $D/r1.fq -I $D/r2.fq -o $D/pe_a.rfq" "-c -i $D/r1.fq -I $D/r2.fq -o $D/pe_b.rfq" "-c -i $D/r1.fq -I $D/r2.fq -o $D/pe_c.rfq" \
   "-d -i $D/pe_c.rfq -o $D/o1.fq -O $D/o2.fq" "-d -i $D/pe_b.rfq -o $D/p1.fq -O $D/p2.fq" "-d -i $D/pe_a.rfq -o /dev/null -O /dev/null" | $B --serve
md5sum $D/pe_a.rfq $D/pe_c.rfq
cmp $D/r1.fq $D/p1.fq && cmp $D/r2.fq $D/p2.fq && cmp $D/r1.fq $D/o1.fq && echo SERVE_ROUNDTRIP_OK
rm -rf /dev/shm/e2e
