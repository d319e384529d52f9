#!/bin/bash
# A/B of a #define in the kernel sources on ONE box (variants are only comparable inside one gpurun call): the library is rebuilt on the box per value.
# usage (on the box): [WL=cfg4] bash tools/ab_macro.sh <file> <MACRO> "v1 v2 ..." [stage ...]
cd $GRAFT_REPO_ROOT; F=$1; M=$2; VALS=$3; shift 3; ST=${@:-gather pos_coder dec:emit}
cp $F /tmp/ab_macro_orig
for v in $VALS; do
  sed -E "s/^(#define $M )[^ ]+/\1$v/" /tmp/ab_macro_orig > $F
  python __graft_entry__.py > /tmp/build.log 2>&1 || { tail -3 /tmp/build.log; continue; }
  python bench.py --workload ${WL:-cfg2} --no-cpu-baseline --no-secondary --no-pmc --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['config']['stage_ms']; print('$M=$v', d['value'], ' '.join('%s=%s' % (k, s.get(k)) for k in '$ST'.split()), d['config']['parity'][:28])"
done
cp /tmp/ab_macro_orig $F
