#!/bin/bash
# SQ instruction / activity counters per kernel (two PMC passes, --kernel-trace only).  usage (on the box): bash tools/pmc_sq.sh <tag>
set -u
TAG=${1:-sq}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $OUT/p1 -o a -- python $ROOT/bench.py ${WL:+--workload $WL} --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-verify --no-pmc > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/p2 -o b -- python $ROOT/bench.py ${WL:+--workload $WL} --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-verify --no-pmc > $OUT/p2.log 2>&1
python $ROOT/tools/pmc_sq.py $(find $OUT/p1 -name "*.db") $(find $OUT/p2 -name "*.db") > $OUT/sq.txt 2>&1
cat $OUT/sq.txt
