#!/bin/bash
# Every kernel of ONE batch of the host work queue (bench.py --queue, encode only) with its start, duration and the idle gap in front of it: where a small batch's
# fixed cost goes - host round trips (gaps), chains of small kernels, tails.  usage (on the box): bash tools/queue_timeline.sh <tag> [queue-chunks=256] [segments=2]
set -u
TAG=${1:-qtl}; Q=${2:-256}; SEGS=${3:-2}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/t -o a -- python $ROOT/bench.py --gpus 1 --queue --segs-per-gpu $SEGS --queue-chunks $Q --steps 1 --warmup 1 --no-verify --encode-only > $OUT/run.log 2>&1
tail -c 1500 $OUT/run.log
python - $(find $OUT/t -name "*.db" | head -1) <<'P' | tee $OUT/queue_timeline_$Q.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name,start,end,stream_id from kernels order by start").fetchall()
first = [i for i, r in enumerate(rows) if 'k_line_index' in r[0] or 'k_gather1' in r[0]]
# batches start with the index kernel(s); two per PE batch: take the start of a batch in the middle of the last pass
starts = [i for k, i in enumerate(first) if k == 0 or i - first[k - 1] > 3]
i0, i1 = starts[-3], starts[-2]
# (memsets in front of the index kernel belong to the batch)
while i0 > 0 and ('fillBuffer' in rows[i0 - 1][0] or 'copyBuffer' in rows[i0 - 1][0]) and rows[i0][1] - rows[i0 - 1][2] < 200000: i0 -= 1
while i1 > 0 and ('fillBuffer' in rows[i1 - 1][0] or 'copyBuffer' in rows[i1 - 1][0]) and rows[i1][1] - rows[i1 - 1][2] < 200000: i1 -= 1
t0 = rows[i0][1]; busy_until = t0; gaps = 0.0
print("%10s %9s %9s  %-3s %s" % ("start us", "dur us", "gap us", "st", "kernel"))
for r in rows[i0:i1]:
    gap = max(0.0, (r[1] - busy_until) / 1e3); gaps += gap
    print("%10.1f %9.1f %9.1f  s%-2s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, r[3], r[0].split('(')[0].replace('void ', '')[:48]))
    busy_until = max(busy_until, r[2])
print("# batch: %.1f us from its first kernel to the next batch's first, %.1f us of it with no kernel running (%d kernels)" % ((rows[i1][1] - t0) / 1e3, gaps + max(0.0, (rows[i1][1] - busy_until) / 1e3), i1 - i0))
P
find $OUT -name "*.db" -delete
