# configs[2] through the driver with phase marks: PE150 2 x 4 GB (-i/-I), files on /dev/shm, process start-up included.  usage: bash tools/e2e_pe.sh
set -e
cd $GRAFT_REPO_ROOT
D=/dev/shm/e2e; mkdir -p $D
./tools/fqgen --profile 1 --reads 11200000 --seed 3 -o $D/r1.fq -O $D/r2.fq
B=repaq_amd/bin/repaq_hip
for i in 1 2 3; do TIMEFORMAT="PE compress wall %R s"; time $B -c -i $D/r1.fq -I $D/r2.fq -o $D/pe.rfq ${E2E_FLAGS:-}; done
$B -c -i $D/r1.fq -I $D/r2.fq -o $D/pe.rfq --trace ${E2E_FLAGS:-} 2>&1 | grep -v "batch resident\|batch encoded" | head -20
$B -c -i $D/r1.fq -I $D/r2.fq -o $D/pe.rfq --trace ${E2E_FLAGS:-} 2>&1 | grep "batch resident\|batch encoded" | head -12
md5sum $D/pe.rfq
for i in 1 2; do TIMEFORMAT="PE decompress to /dev/null wall %R s"; time $B -d -i $D/pe.rfq -o /dev/null -O /dev/null; done
for i in 1 2; do rm -f $D/o1.fq $D/o2.fq; TIMEFORMAT="PE decompress wall %R s"; time $B -d -i $D/pe.rfq -o $D/o1.fq -O $D/o2.fq; done
$B -d -i $D/pe.rfq -o /dev/null -O /dev/null --trace 2>&1 | grep -v "batch resident\|batch decoded" | head
cmp $D/r1.fq $D/o1.fq && cmp $D/r2.fq $D/o2.fq && echo PE_ROUNDTRIP_OK
rm -rf /dev/shm/e2e
