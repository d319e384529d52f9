cd $GRAFT_REPO_ROOT
for pad in ${PADS:-0 4096}; do RFQ_G2_PAD=$pad python bench.py --no-cpu-baseline --no-secondary --no-pmc --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pad',$pad, d['value'], d['config']['stage_ms'])"; done
