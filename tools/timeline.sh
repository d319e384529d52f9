#!/bin/bash
# kernel timeline of the last bench step (rocprofv3 kernel trace; kernels of at least 40 us, per stream).  usage (on the box): bash tools/timeline.sh <tag> [bench args]
set -u
TAG=${1:-tl}; shift; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/t -o a -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-verify "$@" > $OUT/run.log 2>&1
python - $(find $OUT/t -name "*.db" | head -1) <<'P' | tee $OUT/timeline.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name,start,end,stream_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if r[0].startswith('k_nl_bitmap')]
i0 = idx[-2]; t0 = rows[i0][1]
for r in rows[i0:]:
    if r[2] - r[1] >= 40000:
        print("%9.1f .. %9.1f  %8.1f us  s%s %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0].split('(')[0].replace('void ', '')[:40]))
P
find $OUT -name "*.db" -delete
