#!/usr/bin/env python3
"""f4's optional half, measured on the CPU (a design note, no product code): what a second-stage entropy coder behind .rfq would save against the
reference's external `xz` (src/main.cpp:134-159) on the headline shape, section by section.

For every section of every chunk of a configs[2]-shaped image (PE150, NovaSeq-binned qualities): bytes, order-0 entropy (what a static-model rANS / Huffman pass over
the section's bytes reaches, the cheapest thing a GPU does at memory speed), order-1 entropy conditioned on the previous byte (an adaptive-context coder's bound), and
what `xz -3` (the reference's choice) and `xz -1`, `zstd -3` make of the whole image, with their single-thread times.  Test infrastructure: uses the oracle.
usage: python tools/entropy_note.py [pairs=300000]"""
import collections
import math
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import _oracle as O
import _sections as S


def h0(b):
    if not len(b):
        return 0.0
    c = np.bincount(np.frombuffer(b, dtype=np.uint8), minlength=256).astype(np.float64); c = c[c > 0]; p = c / c.sum()
    return float(-(p * np.log2(p)).sum()) * len(b) / 8.0            # bytes


def h1(b):
    if len(b) < 2:
        return float(len(b))
    a = np.frombuffer(b, dtype=np.uint8).astype(np.int64); pair = a[:-1] * 256 + a[1:]
    cp = np.bincount(pair, minlength=65536).astype(np.float64).reshape(256, 256); row = cp.sum(axis=1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(cp > 0, cp * np.log2(cp / row), 0.0)
    return float(-t.sum()) / 8.0 + 1.0


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    fq1, fq2 = O.gen(O.NOVA_PE150, pairs, seed=3)
    t0 = time.time(); rfq = O.encode_file(fq1, fq2, O.PE_TWO_FILES, 1_000_000); t_enc = time.time() - t0
    h, chunks = S.parse(rfq)
    sec = collections.OrderedDict((k, bytearray()) for k in ("fixed+lengths", "lanes+tiles", "x", "y", "name pieces", "sequence (2-bit)", "quality streams", "overlap", "N positions"))
    for c in chunks:
        raw = rfq[c.off:c.off + c.total]
        sec["fixed+lengths"] += raw[:c.len_arrays_end]
        sec["x"] += c.x; sec["y"] += c.y; sec["name pieces"] += c.n1 + c.n2 + c.st; sec["sequence (2-bit)"] += c.seq
        sec["quality streams"] += c.qual; sec["overlap"] += c.ov; sec["N positions"] += c.npos
        sec["lanes+tiles"] += bytes(c.lanes) + b"".join(int(t).to_bytes(2, "little") for t in c.tiles)
    n = len(rfq); text = len(fq1) + len(fq2)
    print("# configs[2] shape, %d pairs: FASTQ %.1f MB -> .rfq %.1f MB (%.4f of the text), %d chunks" % (pairs, text / 1e6, n / 1e6, n / text, len(chunks)))
    print("%-20s %12s %7s %12s %7s %12s %7s" % ("section", "bytes", "share", "order-0", "ratio", "order-1", "ratio"))
    t0_, t1_ = 0.0, 0.0
    for k, b in sec.items():
        b = bytes(b); e0, e1 = h0(b), min(h1(b), float(len(b)))
        t0_ += e0; t1_ += e1
        print("%-20s %12d %6.1f%% %12.0f %7.3f %12.0f %7.3f" % (k, len(b), 100.0 * len(b) / n, e0, e0 / max(1, len(b)), e1, e1 / max(1, len(b))))
    print("%-20s %12d %6.1f%% %12.0f %7.3f %12.0f %7.3f" % ("all sections", n, 100.0, t0_, t0_ / n, t1_, t1_ / n))
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        p = os.path.join(d, "a.rfq"); open(p, "wb").write(rfq)
        for cmd in (["xz", "-3", "-T1", "-k", "-c", p], ["xz", "-1", "-T1", "-k", "-c", p], ["zstd", "-3", "-T1", "-c", p], ["gzip", "-6", "-c", p]):
            try:
                t = time.time(); out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout; dt = time.time() - t
                print("%-20s %12d %6.3f of the image, %.2f s on one core = %.1f MB/s of .rfq (%.0f MB/s of FASTQ)" % (" ".join(cmd[:2]), len(out), len(out) / n, dt, n / dt / 1e6, text / dt / 1e6))
            except (OSError, subprocess.CalledProcessError):
                print("%-20s not available" % " ".join(cmd[:2]))


if __name__ == "__main__":
    main()
