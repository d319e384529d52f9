#!/bin/bash
# A/B of two source trees on ONE box: the library of each is built there and the bench run alternately with RFQ_HIP_LIBRARY pointing at one or the other.
# usage (on the box): [STAGES="gather dec:emit"] bash tools/ab_lib.sh <treeA> <treeB> [rounds] [bench args]   (a tree = a directory holding repaq_amd/csrc and include, e.g. "." and ".ab_old")
cd $GRAFT_REPO_ROOT; A=$1; B=$2; N=${3:-2}; shift 3 2>/dev/null
export AB_STAGES="${STAGES:-index lens+cut gather pos_coder dec:read_table dec:streams dec:emit}"
for t in A B; do d=${!t}; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -Wl,--version-script=repaq_amd/csrc/exports.map -Wno-unused-result $d/repaq_amd/csrc/rfq_api.hip $d/repaq_amd/csrc/rfq_encode.hip $d/repaq_amd/csrc/rfq_decode.hip -o /tmp/lib_$t.so || exit 1; done
for i in $(seq $N); do for t in A B; do
  RFQ_HIP_LIBRARY=/tmp/lib_$t.so python bench.py --no-cpu-baseline --no-secondary --no-pmc --steps 8 --warmup 2 "$@" 2>/dev/null | AB_TAG="$t (${!t})" python -c "
import sys, json, os
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c = d['config']; s = c['stage_ms']
print(os.environ['AB_TAG'], d['value'], 'enc', c.get('encode_MBps_per_gpu'), 'dec', c.get('decode_MBps_per_gpu'), ' '.join('%s=%s' % (k, s.get(k)) for k in os.environ['AB_STAGES'].split()), c['parity'][:20])"
done; done
