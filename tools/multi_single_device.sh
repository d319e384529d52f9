#!/bin/bash
# N > 1 control flow on ONE GPU (RFQ_BENCH_SINGLE_DEVICE=1 puts every rank on GPU 0): the plan (parallel scans), every rank's encode + decode of its own byte
# range, the chunk tables of all ranks against the reference's golden for the configs[3] input.  The rates say nothing (the ranks share one GPU);
# the parity strings and plan_ms do.  usage (on the box): bash tools/multi_single_device.sh <tag>
set -u
TAG=${1:-r03multi}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
export RFQ_BENCH_SINGLE_DEVICE=1
run() { local n=$1; shift; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps 2 --warmup 1 "$@"; }
run 2 > $OUT/gpus2_single_device.json.log 2> $OUT/gpus2.err; tail -c 1500 $OUT/gpus2_single_device.json.log
run 4 --segs-per-gpu 4 > $OUT/gpus4_single_device.json.log 2> $OUT/gpus4.err; tail -c 1500 $OUT/gpus4_single_device.json.log
run 2 --strong > $OUT/gpus2_strong_single_device.json.log 2> $OUT/gpus2s.err; tail -c 1500 $OUT/gpus2_strong_single_device.json.log
for f in $OUT/*.err; do tail -n 3 $f; done; true
