set -e
cd $GRAFT_REPO_ROOT
D=/dev/shm/e2e; mkdir -p $D
./tools/fqgen --profile 0 --reads 11200000 --seed 5 -o $D/a.fq
ls -l $D/a.fq | awk '{print "fastq bytes", $5}'
B=repaq_amd/bin/repaq_hip
for i in 1 2; do TIMEFORMAT="compress wall %R s"; time $B -c -i $D/a.fq -o $D/a.rfq; done
TIMEFORMAT="compress --devices 0,0 (two contexts, one GPU) wall %R s"; time $B -c -i $D/a.fq -o $D/a3.rfq --devices 0,0; cmp $D/a.rfq $D/a3.rfq && echo MULTI_CONTEXT_OK; rm -f $D/a3.rfq
ls -l $D/a.rfq | awk '{print "rfq bytes", $5}'
for i in 1 2; do TIMEFORMAT="decompress wall %R s"; time $B -d -i $D/a.rfq -o $D/b.fq; done
cmp $D/a.fq $D/b.fq && echo ROUNDTRIP_OK
TIMEFORMAT="compare wall %R s"; time $B -p -i $D/a.fq -r $D/a.rfq | head -3
TIMEFORMAT="compress 64MB batches wall %R s"; time $B -c -i $D/a.fq -o $D/a2.rfq --batch_mb 64; cmp $D/a.rfq $D/a2.rfq && echo BATCH_INDEPENDENT_OK

# configs[2] through the driver: PE150 2 x 4 GB (-i/-I), encode + decode round trip (-o/-O)
D=/dev/shm/e2e; mkdir -p $D
./tools/fqgen --profile 1 --reads 11200000 --seed 3 -o $D/r1.fq -O $D/r2.fq
ls -l $D/r1.fq $D/r2.fq | awk '{print "PE fastq bytes", $5}'
TIMEFORMAT="PE compress wall %R s"; time $B -c -i $D/r1.fq -I $D/r2.fq -o $D/pe.rfq
ls -l $D/pe.rfq | awk '{print "PE rfq bytes", $5}'
TIMEFORMAT="PE decompress wall %R s"; time $B -d -i $D/pe.rfq -o $D/o1.fq -O $D/o2.fq
cmp $D/r1.fq $D/o1.fq && cmp $D/r2.fq $D/o2.fq && echo PE_ROUNDTRIP_OK

rm -rf /dev/shm/e2e
