#!/bin/bash
# The hostile-image decode set (tests/_hostile.py, tame: read counts bounded for the interpreter) against an AddressSanitizer build of the SIMT-interpreter
# library - the same .hip sources, g++ -fsanitize=address: every out-of-bounds access to "device" memory (heap blocks there), LDS arrays and host buffers is a report.
# usage (this container, no GPU): bash tools/hostile_asan.sh [out=profiles/r06_hostile_asan.txt]
set -u
cd "$(dirname "$0")/.."; OUT=${1:-profiles/r06_hostile_asan.txt}; B=/tmp/rfq_asan; mkdir -p $B
SRC=repaq_amd/csrc
for f in rfq_api rfq_encode rfq_decode; do
  g++ -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -Wno-unknown-pragmas -Wno-attributes -w -Itests/emu/include -x c++ -c $SRC/$f.hip -o $B/$f.o || exit 1
done
g++ -shared -fsanitize=address -o $B/librfq_emu_asan.so $B/rfq_api.o $B/rfq_encode.o $B/rfq_decode.o -lpthread || exit 1
ASAN_LIB=$(gcc -print-file-name=libasan.so)
{
echo "# hostile-image decode set under AddressSanitizer: $(date -u +%F) g++ $(g++ -dumpversion), tests/_hostile.py (tame), SIMT-interpreter build of repaq_amd/csrc/*.hip"
# (the interpreter runs HIP threads as ucontext fibers on heap stacks: ASan's stack-use-after-return bookkeeping does not follow swapcontext - switched off; leaks are not the subject)
LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0:halt_on_error=1 RFQ_ASAN_LIB=$B/librfq_emu_asan.so python - <<'P'
import os, sys, time, json
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import _hostile as H
from repaq_amd import RfqCodec
c = RfqCodec(device=0, library=os.environ["RFQ_ASAN_LIB"])
print("#", c.version())
for m in [(), (("RFQ_MATERIALISE", "1"),), (("RFQ_WALK", "exact"),), (("RFQ_MATERIALISE", "1"), ("RFQ_WALK", "exact"))]:
    t = time.time()
    s = H.run(c, modes=(m,), good_every=5, tame=True, time_bound_s=300.0)
    print("+".join("%s=%s" % kv for kv in m) or "default", json.dumps(s), "%.0f s" % (time.time() - t), flush=True)
print("# no AddressSanitizer report: every access of every mutant stayed inside its buffers")
P
echo "# exit status $?"
} 2>&1 | tee $OUT
