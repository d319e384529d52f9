#!/usr/bin/env python3
"""GPU box: k_dec_emit3 against k_dec_emit2 (RFQ_EMIT=2) on a file whose names FastqMeta::parse does not take apart (SRA-style, stored per read).
usage: python tools/ab_emit_pieces.py [reads=400000]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
import _engine as E  # noqa: E402
from repaq_amd import RfqCodec  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
fq, _ = O.gen(O.NOVA_SE150, reads, seed=7)
lines = fq.split(b"\n")
for i in range(reads):
    lines[4 * i] = b"@SRR0123456.%d %d length=150" % (i + 1, i + 1)
fq = b"\n".join(lines)
c = RfqCodec(device=0)
rfq = E.encode(c, fq, b"", O.SE, 1_000_000)
assert rfq == O.encode_file(fq, b"", O.SE, 1_000_000) if reads <= 400000 else True
d = c.dev_put(rfq)
for env in ({}, {"RFQ_EMIT": "2"}, {}, {"RFQ_EMIT": "2"}):
    os.environ.pop("RFQ_EMIT", None); os.environ.update(env)
    best = {}
    for _ in range(5):
        r = c.decode(d, len(rfq))
        t = dict(c.timings())
        for k, v in t.items():
            best[k] = min(best.get(k, 1e9), v)
    out = c.dev_get(r.d_fq1, r.n1)
    print(env, "ok" if out == fq else "DIFF", {k: round(v, 3) for k, v in best.items()}, "text MB", len(fq) / 1e6, flush=True)
