#!/bin/bash
# Sweep of one context option over the bench (the environment variable of the same name is read at rfq_create): bash tools/opt_sweep.sh NAME "v1 v2 ..." [rounds=2] [bench args]
cd $GRAFT_REPO_ROOT; NAME=$1; VALS=$2; N=${3:-2}; shift 3 2>/dev/null
export AB_STAGES="${STAGES:-gather pos_coder assemble}"
for i in $(seq $N); do for v in $VALS; do
  env $NAME=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-pmc --steps 8 --warmup 2 "$@" 2>/dev/null < /dev/null | AB_TAG="$NAME=$v" python -c "
import sys, json, os
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c = d['config']; s = c['stage_ms']
print(os.environ['AB_TAG'], d['value'], 'enc', c.get('encode_MBps_per_gpu'), 'dec', c.get('decode_MBps_per_gpu'), ' '.join('%s=%s' % (k, s.get(k)) for k in os.environ['AB_STAGES'].split()), c['parity'][:20])"
done; done
