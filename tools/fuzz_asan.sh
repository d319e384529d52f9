#!/bin/bash
# The differential fuzz (tests/_fuzz.py: general, reader-block, overlap and read-shape seeds; encode == oracle or the same refusal, decode == oracle) against an AddressSanitizer
# build of the SIMT-interpreter library - every access of the round's new host paths (arenas sized in advance and their repeat, index totals left on the device, one
# clear kernel) and of every kernel to "device" memory (heap blocks there), LDS arrays and host buffers is checked.  The contexts are reused across seeds, so a batch meets
# arenas and unit guesses left by a file of another shape.  usage (this container, no GPU): bash tools/fuzz_asan.sh [general=200] [block=30] [out=profiles/r06_fuzz_asan.txt]
set -u
cd "$(dirname "$0")/.."; NG=${1:-200}; NB=${2:-30}; OUT=${3:-profiles/r06_fuzz_asan.txt}; B=/tmp/rfq_asan; mkdir -p $B
SRC=repaq_amd/csrc
for f in rfq_api rfq_encode rfq_decode; do
  g++ -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -Wno-unknown-pragmas -Wno-attributes -w -Itests/emu/include -x c++ -c $SRC/$f.hip -o $B/$f.o || exit 1
done
g++ -shared -fsanitize=address -o $B/librfq_emu_asan.so $B/rfq_api.o $B/rfq_encode.o $B/rfq_decode.o -lpthread || exit 1
ASAN_LIB=$(gcc -print-file-name=libasan.so)
{
echo "# differential fuzz under AddressSanitizer: $(date -u +%F) g++ $(g++ -dumpversion), tests/_fuzz.py, SIMT-interpreter build of repaq_amd/csrc/*.hip ($NG general + $NB block seeds)"
LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0:halt_on_error=1 RFQ_ASAN_LIB=$B/librfq_emu_asan.so NG=$NG NB=$NB python - <<'P'
import os, sys, time, collections
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import _engine as E, _fuzz as F
from repaq_amd import RfqCodec
c = RfqCodec(device=0, library=os.environ["RFQ_ASAN_LIB"])
print("#", c.version())
t = time.time(); res = collections.Counter()
for seed in range(9000, 9000 + int(os.environ["NG"])):
    res[F.check(c, E.encode, seed)] += 1
print("general", dict(res), "%.0f s" % (time.time() - t), flush=True)
t = time.time(); res = collections.Counter()
for seed in range(700, 700 + int(os.environ["NB"])):
    res[F.check_block(c, E.encode, seed)] += 1
print("block", dict(res), "%.0f s" % (time.time() - t), flush=True)
print("# no AddressSanitizer report, no difference")
P
echo "# exit status $?"
} 2>&1 | tee $OUT
