#!/bin/bash
# one workload's bench line, stage times only: bash tools/bench_wl.sh cfg4 [ENV=VAL ...]
cd $GRAFT_REPO_ROOT; W=$1; shift
env "$@" python bench.py --workload $W --no-cpu-baseline --no-secondary --no-pmc --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$W $*', d['value'], d['ms_per_step'], d['config']['stage_ms'], d['config']['parity'][:40])"
