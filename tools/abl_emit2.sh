cd $GRAFT_REPO_ROOT
for t in 0 16 32 48 64 14 62 126; do
  RFQ_TUNE=$((t*4096)) timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-verify --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abl', $t, 'emit', l['config']['stage_ms'].get('dec:emit'))"
done
