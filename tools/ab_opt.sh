#!/bin/bash
# A/B of one context option inside one gpurun call: bash tools/ab_opt.sh NAME "v1 v2 ..." [bench args]
cd $GRAFT_REPO_ROOT; NAME=$1; VALS=$2; shift 2
for v in $VALS; do env $NAME=$v python bench.py --no-cpu-baseline --no-secondary --no-pmc --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$NAME=$v', d['value'], d['ms_per_step'], d['config']['stage_ms'])"; done
