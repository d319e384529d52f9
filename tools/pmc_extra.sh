#!/bin/bash
# extra PMC passes (TLB, wait states) per kernel.  usage (on the box): bash tools/pmc_extra.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
set -u
TAG=${1:-px}; shift; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0; dbs=""
for C in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/p$i -o a -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-verify > $OUT/p$i.log 2>&1
  dbs="$dbs $(find $OUT/p$i -name '*.db' | head -1)"
done
python $ROOT/tools/pmc_sq.py $dbs > $OUT/px.txt 2>&1
find $OUT -name "*.db" -delete
head -14 $OUT/px.txt
