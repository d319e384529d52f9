# .gz in and out through the driver on the GPU box (blocked gzip on many threads vs zlib's one stream).  usage: bash tools/e2e_gz.sh
set -e
cd $GRAFT_REPO_ROOT
D=/dev/shm/e2egz; mkdir -p $D
./tools/fqgen --profile 0 --reads 2800000 --seed 5 -o $D/a.fq
ls -l $D/a.fq | awk '{print "fastq bytes", $5}'
B=repaq_amd/bin/repaq_hip
$B -c -i $D/a.fq -o $D/a.rfq
for i in 1 2; do TIMEFORMAT="decompress -> blocked .gz (driver, level 3) wall %R s"; time $B -d -i $D/a.rfq -o $D/b.fq.gz; done
ls -l $D/b.fq.gz | awk '{print "blocked gz bytes", $5}'
TIMEFORMAT="gzip -3 of the same text (one stream, one core) wall %R s"; time gzip -3 -c $D/a.fq > $D/p.fq.gz
ls -l $D/p.fq.gz | awk '{print "plain gz bytes", $5}'
for i in 1 2; do TIMEFORMAT="compress <- blocked .gz wall %R s"; time $B -c -i $D/b.fq.gz -o $D/b.rfq; done
TIMEFORMAT="compress <- plain .gz (zlib, one stream) wall %R s"; time $B -c -i $D/p.fq.gz -o $D/p.rfq
cmp $D/a.rfq $D/b.rfq && cmp $D/a.rfq $D/p.rfq && echo GZ_INPUTS_OK
gzip -d -c $D/b.fq.gz | cmp - $D/a.fq && echo GZ_OUTPUT_OK
rm -rf $D
