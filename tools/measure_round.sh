#!/bin/bash
# Round measurement on the GPU box: default bench line (with the reference CPU baseline), rocprofv3 kernel trace of the same command,
# the two PMC passes (FETCH_SIZE / WRITE_SIZE, separately, --kernel-trace only), the PE150 profile, the end-to-end CLI timing.
# usage (on the box): bash tools/measure_round.sh <tag>      -> everything lands in gpurun_out/<tag>/
set -u
TAG=${1:-round}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py > $OUT/bench.json.log 2> $OUT/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o se -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_pe -o pe -- python $ROOT/bench.py --pe --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_pe_under_rocprof.json.log 2>&1
cd $ROOT
timeout 500 bash tools/e2e_cli.sh > $OUT/e2e_cli.txt 2>&1
ls -R $OUT | head -40
