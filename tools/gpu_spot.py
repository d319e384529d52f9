"""A few seconds of GPU parity on the paths the round's last changes touch (quality tables with many values + exception records; decode with the
list chain started first, incl. sliced ranges): tests/_fuzz.py generators against the oracle.  usage (on the box): python tools/gpu_spot.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _engine as E
import _fuzz as F
from repaq_amd import RfqCodec
c = RfqCodec(device=0, library=E.PRODUCT_LIB)
res = {}
for seed in range(60):
    r = F.check_gen(c, E.encode, F.qual_case, seed); res[r] = res.get(r, 0) + 1
for seed in range(40):
    r = F.check(c, E.encode, seed); res[r] = res.get(r, 0) + 1
print("gpu spot check:", res)
