#!/bin/bash
# Round-3 GPU session: GPU tests, the bench line, A/B variants in one process, a kernel trace.  usage (on the box): bash tools/r03_run.sh <tag> [variants...]
set -u
TAG=${1:-r03a}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
if [ -z "${SKIP_TESTS:-}" ]; then timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log; fi
if [ -z "${SKIP_BENCH:-}" ]; then timeout 900 python bench.py ${BENCH_ARGS:-} > $OUT/bench.json.log 2> $OUT/bench.err; tail -c 600 $OUT/bench.json.log; fi
if [ $# -gt 0 ]; then timeout 900 python tools/ab_encode.py ${AB_ARGS:-} "$@" > $OUT/ab.log 2>&1; cat $OUT/ab.log; fi
if [ -z "${SKIP_TRACE:-}" ]; then
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfg2 -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json.log 2>&1
cd $ROOT
f=$(find $OUT/trace -name "*.db" 2>/dev/null | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/kernel_stats_cfg2.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
head -40 $OUT/kernel_stats_cfg2.txt
fi
