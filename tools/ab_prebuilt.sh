#!/bin/bash
# Same-box A/B of libraries built beforehand (tools/build_variant.sh): the bench alternates between them, `rounds` times.
# usage (on the box): [STAGES="gather dec:emit"] [BENCH_ARGS="--workload cfg4"] bash tools/ab_prebuilt.sh <rounds> <name> [<name> ...]      (variants/<name>.so; "product" = repaq_amd/lib/librfq_hip.so; name@VAR=value sets an environment switch for that variant)
cd $GRAFT_REPO_ROOT; N=$1; shift
export AB_STAGES="${STAGES:-index lens+cut gather pos_coder dec:read_table dec:streams dec:emit}"
for i in $(seq $N); do for t in "$@"; do
  n=${t%%@*}; e=""; [ "$n" != "$t" ] && e=${t#*@}                        # name@VAR=value: an environment switch for this variant's runs
  lib=variants/$n.so; [ "$n" = product ] && lib=repaq_amd/lib/librfq_hip.so
  env $e RFQ_HIP_LIBRARY=$PWD/$lib python bench.py --no-cpu-baseline --no-secondary --no-pmc --steps 8 --warmup 2 $BENCH_ARGS 2>/dev/null | AB_TAG="$t" python -c "
import sys, json, os
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c = d['config']; s = c['stage_ms']
print(os.environ['AB_TAG'], d['value'], 'enc', c.get('encode_MBps_per_gpu'), 'dec', c.get('decode_MBps_per_gpu'), ' '.join('%s=%s' % (k, s.get(k)) for k in os.environ['AB_STAGES'].split()), c['parity'][:20])"
done; done
