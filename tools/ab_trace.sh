#!/bin/bash
# A/B of one context option under a kernel trace (one stream, so that kernel times are times alone): bash tools/ab_trace.sh NAME "v1 v2" kernel_regex
cd /tmp; export TMPDIR=/tmp; NAME=$1; VALS=$2; PAT=$3; ROOT=$GRAFT_REPO_ROOT
for v in $VALS; do
  rm -rf /tmp/abt; env $NAME=$v RFQ_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/abt -o t -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-pmc --no-verify > /tmp/abt.log 2>&1
  f=$(find /tmp/abt -name "*.db" | head -1); echo "== $NAME=$v"; (cd $ROOT && python tools/rocprof_summary.py $f | grep -E "$PAT" | cut -c1-120)
done
