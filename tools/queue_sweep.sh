#!/bin/bash
# bench.py --queue on ONE GPU by batch size: the rate of the host work queue against the one-shot rate of the same input (VERDICT r5 #2a).
# An 8-segment input (2 x 8 GB: one GPU's share of configs[3]) in batches of 128 ... 4096 chunks, encode only and encode + decode.
# usage (on the box): bash tools/queue_sweep.sh <tag> [segments] [sizes...]
set -u
TAG=${1:-qsweep}; SEGS=${2:-8}; shift 2 2>/dev/null; SIZES=${@:-128 256 512 1024 2048 4096}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
for q in $SIZES; do for mode in --encode-only ""; do
  timeout 900 python bench.py --gpus 1 --queue --segs-per-gpu $SEGS --queue-chunks $q --steps 3 --warmup 1 --no-verify $mode 2> $OUT/err_$q.txt | tail -1 | Q=$q MODE="${mode:-encode+decode}" python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); r = d['config']['ranks'][0]
print('queue_chunks', os.environ['Q'], os.environ['MODE'], 'value_MBps', d['value'], 'batches', d['config']['batches'], 'encode_MBps', r['encode_MBps'], 'decode_MBps', r['decode_MBps'], 'ms_per_step', d['ms_per_step'])"
done; done | tee $OUT/queue_sweep.txt
