#!/bin/bash
# Lane utilisation of the VALU per kernel: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) (one PMC pass, --kernel-trace only).  usage (on the box): bash tools/pmc_lanes.sh <tag>
set -u
TAG=${1:-lanes}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for W in ${WLS:-cfg2}; do
timeout 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace -d $OUT/$W -o $W -- python $ROOT/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-verify --no-pmc > $OUT/$W.log 2>&1
python $ROOT/tools/pmc_sq.py $(find $OUT/$W -name "*.db") > $OUT/lanes_$W.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
awk 'NR==1 {print $0, "  lanes/64"} NR>1 && $3>0 {printf "%s  %.2f\n", $0, $2/($3*64)}' $OUT/lanes_$W.txt | cut -c1-140 | head -30
done
