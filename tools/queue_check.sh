#!/bin/bash
# bench.py --queue (the host work queue in bench form) and the stream-to-device set-up of --gpus N, on ONE GPU.
#   small : a 2-segment input through the queue with one rank and with two ranks on GPU 0, and the static-share path with two ranks (RFQ_BENCH_SINGLE_DEVICE=1)
#   cfg3  : the WHOLE configs[3] logical input (64 segments, 2 x 64 GB) resident on one GPU, every one of its 53,750 chunk images against the reference's table
# usage (on the box): bash tools/queue_check.sh <tag> [small|cfg3]
set -u
TAG=${1:-queue}; WHAT=${2:-small}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
export RFQ_BENCH_SINGLE_DEVICE=1
run() { local n=$1; shift; timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700 + n)) bench.py --gpus $n "$@"; }
if [ "$WHAT" = small ]; then
  timeout 600 python bench.py --gpus 1 --queue --segs-per-gpu 2 --queue-chunks 64 --steps 2 --warmup 1 > $OUT/queue_gpus1_2seg.json.log 2> $OUT/q1.err; tail -c 1200 $OUT/queue_gpus1_2seg.json.log
  run 2 --queue --segs-per-gpu 1 --queue-chunks 64 --steps 2 --warmup 1 > $OUT/queue_gpus2_single_device.json.log 2> $OUT/q2.err; tail -c 1500 $OUT/queue_gpus2_single_device.json.log
  run 2 --segs-per-gpu 2 --steps 2 --warmup 1 > $OUT/gpus2_single_device.json.log 2> $OUT/m2.err; tail -c 1500 $OUT/gpus2_single_device.json.log
else
  timeout 1500 python bench.py --gpus 1 --queue --segs-per-gpu 64 --queue-chunks 1024 --steps 2 --warmup 1 --encode-only > $OUT/queue_cfg3_whole_one_gpu.json.log 2> $OUT/q3.err; tail -c 2000 $OUT/queue_cfg3_whole_one_gpu.json.log
fi
for f in $OUT/*.err; do echo "== $f"; grep -v 'amdgpu.ids' $f | tail -n 5; done; true
