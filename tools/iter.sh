#!/bin/bash
# One iteration's worth of GPU measurements (a few minutes): the GPU parity suite (optional), the bench line without the CPU legs, and a rocprofv3
# kernel trace of the headline workload summarised per kernel.  usage (on the box): bash tools/iter.sh <tag> [tests|notests] [workloads...]
set -u
TAG=${1:-iter}; TESTS=${2:-tests}; shift 2 2>/dev/null; WL=${@:-cfg2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
if [ "$TESTS" = tests ]; then timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; tail -5 $OUT/gputest.log; fi
timeout 600 python bench.py --no-cpu-baseline --no-pmc > $OUT/bench.json.log 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json.log
cd /tmp; export TMPDIR=/tmp
for W in $WL; do
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$W -o $W -- python $ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-pmc > $OUT/bench_${W}_under_rocprof.json.log 2>&1
f=$(find $OUT/trace_$W -name "*.db" 2>/dev/null | head -1); [ -n "$f" ] && (cd $ROOT && python tools/rocprof_summary.py $f > $OUT/kernel_stats_$W.txt 2>&1)
done
find $OUT -name "*.db" -size +20M -delete
head -45 $OUT/kernel_stats_cfg2.txt 2>/dev/null
