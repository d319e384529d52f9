#!/bin/bash
# Round measurement on the GPU box: the default bench line (configs[2] PE150 2 x 4 GB, reference CPU baseline on a sample, secondary configs[1] /
# configs[4] lines), rocprofv3 kernel trace of the same workload, the two PMC passes (FETCH_SIZE / WRITE_SIZE, separately, --kernel-trace only).
# usage (on the box): bash tools/measure_round.sh <tag> [quick]     -> everything lands in gpurun_out/<tag>/
set -u
TAG=${1:-round}; QUICK=${2:-}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py > $OUT/bench.json.log 2> $OUT/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfg2 -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-pmc > $OUT/bench_under_rocprof.json.log 2>&1
if [ -z "$QUICK" ] || [ "$QUICK" = kernels ]; then
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-verify --no-pmc > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-verify --no-pmc > $OUT/pmc_write.log 2>&1
fi
if [ -z "$QUICK" ]; then
for W in cfg1 cfg4; do
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch_$W -o f -- python $ROOT/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-pmc > $OUT/pmc_fetch_$W.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write_$W -o w -- python $ROOT/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-pmc > $OUT/pmc_write_$W.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_cfg1 -o cfg1 -- python $ROOT/bench.py --workload cfg1 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_cfg1_under_rocprof.json.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_cfg4 -o cfg4 -- python $ROOT/bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_cfg4_under_rocprof.json.log 2>&1
fi
cd $ROOT
if [ "$QUICK" = kernels ]; then   # kernels only: the bench line, the trace and the PMC passes of configs[2] (host paths unchanged since the last full set)
for d in trace; do f=$(find $OUT/$d -name "*.db" 2>/dev/null | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/${d}_kernel_stats.txt 2>&1; done
ff=$(find $OUT/pmc_fetch -name "*.db" 2>/dev/null | head -1); fw=$(find $OUT/pmc_write -name "*.db" 2>/dev/null | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_summary.py $ff $fw $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
find $OUT -name "*.db" -size +20M -delete; ls -R $OUT | head -30; exit 0
fi
for d in trace trace_cfg1 trace_cfg4; do f=$(find $OUT/$d -name "*.db" 2>/dev/null | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/${d}_kernel_stats.txt 2>&1; done
ff=$(find $OUT/pmc_fetch -name "*.db" 2>/dev/null | head -1); fw=$(find $OUT/pmc_write -name "*.db" 2>/dev/null | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_summary.py $ff $fw $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
for W in cfg1 cfg4; do ff=$(find $OUT/pmc_fetch_$W -name "*.db" 2>/dev/null | head -1); fw=$(find $OUT/pmc_write_$W -name "*.db" 2>/dev/null | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_summary.py $ff $fw $OUT/pmc_traffic_$W.json > $OUT/pmc_traffic_$W.txt 2>&1; done
# the N > 1 control flow on this one GPU (parity strings, plan_ms), the SQ counters, the driver end to end
bash tools/multi_single_device.sh $TAG/multi > $OUT/multi.log 2>&1
# the host work queue in bench form: small (1 and 2 ranks on this GPU) and the WHOLE configs[3] input through one GPU
bash tools/queue_check.sh $TAG/queue small > $OUT/queue_small.log 2>&1
bash tools/queue_check.sh $TAG/queue cfg3 > $OUT/queue_cfg3.log 2>&1
bash tools/pmc_sq.sh $TAG/sq > /dev/null 2>&1; cp $OUT/sq/sq.txt $OUT/sq_counters.txt 2>/dev/null
bash tools/e2e_pe.sh > $OUT/e2e_pe.txt 2>&1
# the raw databases are large: keep the summaries, drop the traces
find $OUT -name "*.db" -size +20M -delete
ls -R $OUT | head -60
