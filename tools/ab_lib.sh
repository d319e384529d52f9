#!/bin/bash
# A/B of two source trees on ONE box: the library of each is built there and the bench run alternately with RFQ_HIP_LIBRARY pointing at one or the other.
# usage (on the box): bash tools/ab_lib.sh <treeA> <treeB> [rounds] [bench args]      (a tree = a directory holding repaq_amd/csrc and include, e.g. "." and ".ab_old")
cd $GRAFT_REPO_ROOT; A=$1; B=$2; N=${3:-2}; shift 3 2>/dev/null
for t in A B; do d=${!t}; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result $d/repaq_amd/csrc/rfq_api.hip $d/repaq_amd/csrc/rfq_encode.hip $d/repaq_amd/csrc/rfq_decode.hip -o /tmp/lib_$t.so || exit 1; done
for i in $(seq $N); do for t in A B; do
  RFQ_HIP_LIBRARY=/tmp/lib_$t.so python bench.py --no-cpu-baseline --no-secondary --no-pmc --steps 8 --warmup 2 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['config']['stage_ms']; c=d['config']
print('$t (${!t})', d['value'], 'enc', c.get('encode_MBps_per_gpu'), 'dec', c.get('decode_MBps_per_gpu'), ' '.join('%s=%s' % (k, s.get(k)) for k in ('dec:read_table','dec:streams','dec:emit')), c['parity'][:20])"
done; done
