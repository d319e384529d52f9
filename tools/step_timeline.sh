#!/bin/bash
# Every kernel of the LAST step (encode + decode) of a bench workload with its start, duration, stream and the idle gap in front of it - where a step's fixed cost goes
# (host round trips show as gaps, chains of small kernels as rows of a few microseconds).  usage (on the box): bash tools/step_timeline.sh <tag> <workload> [bench args]
set -u
TAG=${1:-stl}; WL=${2:-cfg1}; shift 2 2>/dev/null; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/t_$WL -o a -- python $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-verify --no-pmc "$@" > $OUT/run_$WL.log 2>&1
python - $(find $OUT/t_$WL -name "*.db" | head -1) <<'P' | tee $OUT/step_timeline_$WL.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name,start,end,stream_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'k_line_index' in r[0] or 'k_nl_bitmap' in r[0] or 'k_gather1' in r[0]]
starts = [i for k, i in enumerate(idx) if k == 0 or i - idx[k - 1] > 6]
# the last timed step; the indexed-decode loop that follows the timed steps has no encode in it: cut at the second emitter launch behind the step's start
i0 = starts[-1]
em = [i for i in range(i0, len(rows)) if 'k_dec_emit' in rows[i][0]]
i1 = em[0] + 1 if em else len(rows)
while i0 > 0 and ('fillBuffer' in rows[i0 - 1][0] or 'copyBuffer' in rows[i0 - 1][0]) and rows[i0][1] - rows[i0 - 1][2] < 100000: i0 -= 1
t0 = rows[i0][1]; busy = t0; gaps = 0.0
print("%10s %9s %9s  %-3s %s" % ("start us", "dur us", "gap us", "st", "kernel"))
for r in rows[i0:i1]:
    gap = max(0.0, (r[1] - busy) / 1e3); gaps += gap
    print("%10.1f %9.1f %9.1f  s%-2s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, r[3], r[0].split('(')[0].replace('void ', '')[:48]))
    busy = max(busy, r[2])
print("# step: %.1f us from its first kernel to the end of its emitter, %.1f us of it with no kernel running (%d kernels)" % ((busy - t0) / 1e3, gaps, i1 - i0))
P
find $OUT -name "*.db" -delete
