#!/bin/bash
# Where the waves' cycles go, per kernel: wait / issue / memory-latency SQ counters (PMC passes, --kernel-trace only).  usage (on the box): bash tools/pmc_sq2.sh <tag>
set -u
TAG=${1:-sq2}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/avail_sq.txt
run() { # name counters...
  local n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$n -o $n -- python $ROOT/bench.py ${WL:+--workload $WL} --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-verify --no-pmc > $OUT/$n.log 2>&1
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
run b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
run c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
run d SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS
run e SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU
python $ROOT/tools/pmc_sq.py $(find $OUT/a $OUT/b $OUT/c $OUT/d $OUT/e -name "*.db" 2>/dev/null) > $OUT/sq2.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
cut -c1-400 $OUT/sq2.txt | head -14
