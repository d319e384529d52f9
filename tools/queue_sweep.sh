#!/bin/bash
# bench.py --queue on ONE GPU by batch size and worker contexts: the rate of the host work queue against the one-shot rate of the same input (VERDICT r5 #2a).
# An 8-segment input (2 x 8 GB: one GPU's share of configs[3]); encode + decode (value) and the encode-only pass of the same run (encode_MBps_wall).
# usage (on the box): [WORKERS="1 2 3"] bash tools/queue_sweep.sh <tag> [segments] [sizes...]
set -u
TAG=${1:-qsweep}; SEGS=${2:-8}; shift 2 2>/dev/null; SIZES=${@:-128 256 512 1024 2048}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
for w in ${WORKERS:-1 2}; do for q in $SIZES; do
  timeout 900 python bench.py --gpus 1 --queue --segs-per-gpu $SEGS --queue-chunks $q --queue-workers $w --steps 3 --warmup 1 --no-verify 2> $OUT/err_${q}_$w.txt | tail -1 | tee $OUT/line_${q}_$w.json | Q=$q W=$w python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); c = d['config']
print('queue_chunks', os.environ['Q'], 'workers', os.environ['W'], 'batches', c['batches'], 'value_MBps', d['value'], 'encode_MBps_wall', c['encode_MBps_wall'], 'ms_per_step', d['ms_per_step'])"
done; done | tee $OUT/queue_sweep.txt
