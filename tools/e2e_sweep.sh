# host pipeline sweep on the GPU box: 4 GB SE150 file -> file on /dev/shm, wall time per setting (+ --trace runs of the default)
cd $GRAFT_REPO_ROOT
D=/dev/shm/e2e; mkdir -p $D
./tools/fqgen --profile 0 --reads 11200000 --seed 5 -o $D/a.fq
ls -l $D/a.fq | awk '{print "fastq bytes", $5}'
B=repaq_amd/bin/repaq_hip
printf '@r1\nACGT\n+\nIIII\n' > $D/tiny.fq
TIMEFORMAT="start-up (4-line FASTQ) wall %R s"; time $B -c -i $D/tiny.fq -o $D/tiny.rfq
TIMEFORMAT="start-up again wall %R s"; time $B -c -i $D/tiny.fq -o $D/tiny.rfq
TIMEFORMAT="cat wall %R s"; time cat $D/a.fq > /dev/null
$B -c -i $D/a.fq -o $D/ref.rfq --batch_mb 16 --io_threads 1
for bt in "256 16 8" "256 16 8" "256 16 16" "256 16 4" "64 16 8" "64 8 8" "128 8 8" "128 16 8" "512 16 8"; do
  set -- $bt
  TIMEFORMAT="compress batch $1 MB block $2 MB io_threads $3: wall %R s"; time $B -c -i $D/a.fq -o $D/a.rfq --batch_mb $1 --block_mb $2 --io_threads $3
  cmp $D/a.rfq $D/ref.rfq || echo IMAGE_DIFFERS
done
$B -c -i $D/a.fq -o $D/a.rfq --trace 2>&1 | grep -v "batch resident\|batch encoded"
for bt in "256 16 1" "256 16 1" "256 16 2" "64 16 1" "128 16 1" "512 16 1" "256 32 1" "256 8 1"; do
  set -- $bt
  rm -f $D/b.fq
  TIMEFORMAT="decompress batch $1 MB block $2 MB write_threads $3: wall %R s"; time $B -d -i $D/a.rfq -o $D/b.fq --batch_mb $1 --block_mb $2 --write_threads $3
  cmp $D/a.fq $D/b.fq || echo TEXT_DIFFERS
done
rm -f $D/b.fq; $B -d -i $D/a.rfq -o $D/b.fq --trace 2>&1 | grep -v "batch resident\|batch decoded"
TIMEFORMAT="decompress to /dev/null: wall %R s"; time $B -d -i $D/a.rfq -o /dev/null
TIMEFORMAT="decompress to /dev/null: wall %R s"; time $B -d -i $D/a.rfq -o /dev/null
TIMEFORMAT="compress to /dev/null: wall %R s"; time $B -c -i $D/a.fq -o /dev/null
TIMEFORMAT="cp wall %R s"; rm -f $D/b.fq; time cp $D/a.fq $D/b.fq
TIMEFORMAT="compare wall %R s"; time $B -p -i $D/a.fq -r $D/a.rfq | head -3
rm -rf /dev/shm/e2e
