"""GPU (MI355X): hostile .rfq images through rfq_decode_batch (tests/_hostile.py).  >= 200 mutants of each of three images - bit flips anywhere,
random bytes in the header and in every chunk's fixed fields / length arrays / quality length table, truncation at every section boundary, chunk indexes that lie -
under the default path, RFQ_MATERIALISE=1 and RFQ_WALK=exact: every call returns RFQ_E_FORMAT / RFQ_E_DATA / ... or some text within the time bound, and the SAME
context then decodes the good image to the expected text (no fault, no sticky state).  The interpreter cannot see an out-of-bounds device access that stays inside
the process's mappings; the GPU does - a fault ends the process, so the run lives in a CHILD process and the test reports how it ended."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r"""
import json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import _engine as E, _hostile as H
from repaq_amd import RfqCodec
c = RfqCodec(device=0, library=E.PRODUCT_LIB)
assert "gfx950" in c.version()
modes = [(), (("RFQ_MATERIALISE", "1"),), (("RFQ_WALK", "exact"),), (("RFQ_MATERIALISE", "1"), ("RFQ_WALK", "exact"))]
out = {}
for m in modes:
    tot = None
    for seed in (7, 8, 9, 10):                       # four seeds: ~2,900 mutants per mode
        s = H.run(c, modes=(m,), seed=seed, good_every=1, time_bound_s=60.0)
        if tot is None:
            tot = s
        else:
            for k in ("mutants", "decoded", "good_checks"):
                tot[k] += s[k]
            for k, v in s["errors"].items():
                tot["errors"][k] = tot["errors"].get(k, 0) + v
            if s["slowest_s"] > tot["slowest_s"]:
                tot["slowest_s"], tot["slowest"] = s["slowest_s"], s["slowest"]
    out["+".join("%%s=%%s" %% kv for kv in m) or "default"] = tot
    print("MODE", json.dumps(out), flush=True)
c.close()
print("SUMMARY " + json.dumps(out))
""" % (HERE, os.path.join(HERE, "golden"), os.path.dirname(HERE))


def test_hostile_images_never_fault_and_leave_no_state():
    r = subprocess.run([sys.executable, "-c", CHILD], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    tail = (r.stdout[-1500:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, "the child process ended with status %d (negative: a signal - a device fault aborts the process):\n%s" % (r.returncode, tail)
    line = [l for l in r.stdout.splitlines() if l.startswith("SUMMARY ")]
    assert line, tail
    s = json.loads(line[-1][8:])
    assert len(s) == 4
    for mode, v in s.items():
        assert v["mutants"] >= 2400 and v["good_checks"] >= 2400 and v["errors"].get("FORMAT", 0) > 400 and v["decoded"] > 400, (mode, v)
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out):
        json.dump(s, open(os.path.join(out, "hostile_gpu_summary.json"), "w"), indent=1)
