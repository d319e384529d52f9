#!/usr/bin/env python3
"""Debug aid (GPU box): decode one oracle-encoded SE file through the C-ABI with the emitters / walks selectable by environment, report where the text differs.
usage: python tools/dbg_decode_diff.py [reads=120000] [seed=43] [chunk_bases=100000]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
from repaq_amd import RfqCodec  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 43
cb = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
fq, _ = O.gen(O.NOVA_SE150, reads, seed=seed)
rfq = O.encode_file(fq, b"", O.SE, cb)
lib = os.environ.get("DBG_LIB")
c = RfqCodec(device=0, library=lib) if lib else RfqCodec(device=0)
print("library", lib or "in-tree", c.version())
import _engine as E  # noqa: E402
step = int(sys.argv[4]) if len(sys.argv) > 4 else 0
for env in ({}, {"RFQ_EMIT": "2"}):
    for k in ("RFQ_EMIT", "RFQ_WALK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    d = E.decode_in_slices(c, rfq, False, step) if step else c.decode_bytes(rfq)
    nd = 0; first = []
    if d != fq:
        import numpy as np
        a = np.frombuffer(d, dtype=np.uint8); b = np.frombuffer(fq, dtype=np.uint8)
        if len(a) == len(b):
            w = np.nonzero(a != b)[0]; nd = len(w); first = w[:12].tolist()
        else:
            nd = -1
    print(env, sorted(dict(c.timings())), "len", len(d), len(fq), "diffs", nd, first, flush=True)
    for p in first[:3]:
        ls = fq.rfind(b"\n", 0, p) + 1
        print("   at", p, "col", p - ls, "got", d[p:p + 1], "want", fq[p:p + 1], "record", fq.count(b"\n", 0, p) // 4, flush=True)
