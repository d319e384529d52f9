# End to end through the driver on the GPU box: files on /dev/shm, process start-up included.  usage: bash tools/e2e_cli.sh
set -e
cd $GRAFT_REPO_ROOT
D=/dev/shm/e2e; mkdir -p $D
./tools/fqgen --profile 0 --reads 11200000 --seed 5 -o $D/a.fq
ls -l $D/a.fq | awk '{print "fastq bytes", $5}'
B=repaq_amd/bin/repaq_hip
printf '@r1\nACGT\n+\nIIII\n' > $D/tiny.fq
for i in 1 2; do TIMEFORMAT="start-up floor (4-line FASTQ -> .rfq) wall %R s"; time $B -c -i $D/tiny.fq -o $D/tiny.rfq; done
TIMEFORMAT="cat a.fq > /dev/null wall %R s"; time cat $D/a.fq > /dev/null
TIMEFORMAT="cp a.fq b.fq (the tmpfs write floor of a 4 GB output) wall %R s"; time cp $D/a.fq $D/b.fq; rm -f $D/b.fq
for i in 1 2 3; do TIMEFORMAT="compress wall %R s"; time $B -c -i $D/a.fq -o $D/a.rfq; done
$B -c -i $D/a.fq -o $D/a.rfq --trace 2>&1 | grep -v "batch resident\|batch encoded"
TIMEFORMAT="compress --devices 0,0 (two contexts, one GPU) wall %R s"; time $B -c -i $D/a.fq -o $D/a3.rfq --devices 0,0; cmp $D/a.rfq $D/a3.rfq && echo MULTI_CONTEXT_OK; rm -f $D/a3.rfq
ls -l $D/a.rfq | awk '{print "rfq bytes", $5}'
for i in 1 2 3; do rm -f $D/b.fq; TIMEFORMAT="decompress wall %R s"; time $B -d -i $D/a.rfq -o $D/b.fq; done
cmp $D/a.fq $D/b.fq && echo ROUNDTRIP_OK
rm -f $D/b.fq; $B -d -i $D/a.rfq -o $D/b.fq --trace 2>&1 | grep -v "batch resident\|batch decoded"
for i in 1 2; do TIMEFORMAT="decompress -o /dev/null (no tmpfs page allocation) wall %R s"; time $B -d -i $D/a.rfq -o /dev/null; done
for i in 1 2; do TIMEFORMAT="compare wall %R s"; time $B -p -i $D/a.fq -r $D/a.rfq | head -3; done
TIMEFORMAT="compress 16 MB batches, one reader wall %R s"; time $B -c -i $D/a.fq -o $D/a2.rfq --batch_mb 16 --io_threads 1; cmp $D/a.rfq $D/a2.rfq && echo BATCH_INDEPENDENT_OK
TIMEFORMAT="compress -v (every batch decoded and compared on the device) wall %R s"; time $B -c -v -i $D/a.fq -o $D/a2.rfq; cmp $D/a.rfq $D/a2.rfq && echo VERIFY_RUN_OK
rm -f $D/a2.rfq $D/b.fq

# configs[2] through the driver: PE150 2 x 4 GB (-i/-I), encode + decode round trip (-o/-O)
./tools/fqgen --profile 1 --reads 11200000 --seed 3 -o $D/r1.fq -O $D/r2.fq
ls -l $D/r1.fq $D/r2.fq | awk '{print "PE fastq bytes", $5}'
for i in 1 2; do TIMEFORMAT="PE compress wall %R s"; time $B -c -i $D/r1.fq -I $D/r2.fq -o $D/pe.rfq; done
ls -l $D/pe.rfq | awk '{print "PE rfq bytes", $5}'
md5sum $D/pe.rfq
for i in 1 2; do rm -f $D/o1.fq $D/o2.fq; TIMEFORMAT="PE decompress wall %R s"; time $B -d -i $D/pe.rfq -o $D/o1.fq -O $D/o2.fq; done
cmp $D/r1.fq $D/o1.fq && cmp $D/r2.fq $D/o2.fq && echo PE_ROUNDTRIP_OK
rm -rf /dev/shm/e2e
