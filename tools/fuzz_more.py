"""A longer run of the differential fuzz (tests/_fuzz.py) on the GPU box: [RFQ_GATHER=old ...] python tools/fuzz_more.py [general seeds = 2000] [block seeds = 600]  (test infrastructure;
the RFQ_* switches of the environment are read by rfq_create: the same seeds under every alternative formulation - tools/fuzz_forms.sh)"""
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","."), "tests")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","."))
import _engine as E, _fuzz as F
from repaq_amd import RfqCodec
c = RfqCodec(device=0, library=E.PRODUCT_LIB)
bad = 0
import collections
res = collections.Counter()
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 2000; NB = int(sys.argv[2]) if len(sys.argv) > 2 else 600
print("switches:", {k: v for k, v in os.environ.items() if k.startswith("RFQ_") and k != "RFQ_HIP_LIBRARY"} or "defaults")
for seed in range(300, 300 + NG):
    try: res[F.check(c, E.encode, seed)] += 1
    except AssertionError as e: bad += 1; print("FAIL", seed, str(e)[:200])
    except Exception as e: bad += 1; print("EXC", seed, repr(e)[:200])
print(res, "bad", bad)
res = collections.Counter()
for seed in range(120, 120 + NB):
    try: res[F.check_block(c, E.encode, seed)] += 1
    except AssertionError as e: bad += 1; print("BLOCK FAIL", seed, str(e)[:200])
    except Exception as e: bad += 1; print("BLOCK EXC", seed, repr(e)[:200])
print("block", res, "bad", bad)
