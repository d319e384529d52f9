#!/bin/bash
# k_overlap timed alone (RFQ_TUNE bit 11 keeps it on the main stream): rocprofv3 kernel trace of two bench steps.  usage (on the box): bash tools/ovl_alone.sh <tag>
set -u
TAG=${1:-ovl}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
RFQ_TUNE=2048 timeout 300 rocprofv3 --kernel-trace -d $OUT/t -o a -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-verify > $OUT/run.log 2>&1
python $ROOT/tools/rocprof_summary.py $(find $OUT/t -name "*.db" | head -1) | grep -E "k_overlap|k_read_table|k_gather" 
find $OUT -name "*.db" -delete
