"""A longer run of the differential fuzz (tests/_fuzz.py) on the GPU box: python tools/fuzz_more.py  (test infrastructure)"""
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","."), "tests")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","."))
import _engine as E, _fuzz as F
from repaq_amd import RfqCodec
c = RfqCodec(device=0, library=E.PRODUCT_LIB)
bad = 0
import collections
res = collections.Counter()
for seed in range(300, 2300):
    try: res[F.check(c, E.encode, seed)] += 1
    except AssertionError as e: bad += 1; print("FAIL", seed, str(e)[:200])
    except Exception as e: bad += 1; print("EXC", seed, repr(e)[:200])
print(res, "bad", bad)
res = collections.Counter()
for seed in range(120, 720):
    try: res[F.check_block(c, E.encode, seed)] += 1
    except AssertionError as e: bad += 1; print("BLOCK FAIL", seed, str(e)[:200])
    except Exception as e: bad += 1; print("BLOCK EXC", seed, repr(e)[:200])
print("block", res, "bad", bad)
