#!/usr/bin/env python3
"""Golden maker for the full-size configs (build container only: needs oracle/_ref/repaq, the reference compiled by
oracle/Makefile, and ~20 GB of /tmp).  Writes tests/golden/big.json:
  cfg2  BASELINE.json configs[2]: synthetic NovaSeq PE150 2 x 4 GB (fqgen profile 1, 11.2 M pairs, seed 3): md5 / size of the
        reference's .rfq, md5s of the inputs, and whether the reference's own decode restored them
  cfg3_share  one GPU's share of configs[3] (2 x 64 GB over 8 GPUs = 2 x 8 GB: fqgen profile 1, 22.4 M pairs, seed 4) encoded by
        the reference as a file of its own: md5 / size of the image + crc32 and size of every chunk image (the table a
        chunk-parallel encode is checked against, chunk by chunk)
  cfg4  the bench's configs[4]-shaped input (BGI-style PE100 with long names, 40 quality values, N runs: fqgen profile 3, 1.4 M pairs, seed 6,
        --nppm 0 --nquals 40): md5 / size of the reference's .rfq and whether its own decode restored the inputs
The committed JSON is data (hashes of inputs and of the reference's outputs); no reference source is stored."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O

FQGEN = os.path.join(ROOT, "tools", "fqgen")


def md5_file(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def run_case(label, pairs, seed, chunk_table, profile=1, extra=(), nppm=20, n_quals=0):
    out = {"label": label, "profile": profile, "pairs": pairs, "seed": seed, "nppm": nppm, "n_quals": n_quals, "k": 1000, "paired": O.PE_TWO_FILES}
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        r1, r2, rfq = (os.path.join(d, n) for n in ("r1.fq", "r2.fq", "o.rfq"))
        subprocess.check_call([FQGEN, "--profile", str(profile), "--reads", str(pairs), "--seed", str(seed)] + list(extra) + ["-o", r1, "-O", r2])
        out["fq_bytes"] = [os.path.getsize(r1), os.path.getsize(r2)]
        out["fq_md5"] = [md5_file(r1), md5_file(r2)]
        t0 = time.time()
        subprocess.check_call([O.REF_BIN, "-c", "-i", r1, "-I", r2, "-o", rfq])
        out["ref_encode_s"] = round(time.time() - t0, 1)
        out["rfq_len"] = os.path.getsize(rfq); out["rfq_md5"] = md5_file(rfq)
        if chunk_table:
            img = open(rfq, "rb").read()
            offs = O.chunk_table(img)
            out["n_chunks"] = len(offs) - 1
            out["header_len"] = offs[0]
            out["chunk_len"] = [offs[i + 1] - offs[i] for i in range(len(offs) - 1)]
            out["chunk_crc32"] = ["%08x" % (zlib.crc32(img[offs[i]:offs[i + 1]]) & 0xFFFFFFFF) for i in range(len(offs) - 1)]
            del img
        else:
            o1, o2 = os.path.join(d, "b1.fq"), os.path.join(d, "b2.fq")
            t0 = time.time()
            subprocess.check_call([O.REF_BIN, "-d", "-i", rfq, "-o", o1, "-O", o2])
            out["ref_decode_s"] = round(time.time() - t0, 1)
            out["ref_roundtrip"] = md5_file(o1) == out["fq_md5"][0] and md5_file(o2) == out["fq_md5"][1]
    print(label, {k: v for k, v in out.items() if k not in ("chunk_len", "chunk_crc32")}, flush=True)
    return out


def main():
    assert O.have_ref(), "build the reference first: make -C oracle"
    which = sys.argv[1:] or ["cfg2", "cfg3_share"]
    p = os.path.join(HERE, "big.json")
    res = json.load(open(p)) if os.path.exists(p) else {}
    if "cfg2" in which:
        res["cfg2"] = run_case("cfg2_pe150_2x4GB", 11_200_000, 3, False)
        json.dump(res, open(p, "w"), indent=0, sort_keys=True)
    if "cfg4" in which:
        res["cfg4"] = run_case("cfg4_bgi_pe100_q40", 1_400_000, 6, False, profile=3, extra=("--nppm", "0", "--nquals", "40"), nppm=0, n_quals=40)
        json.dump(res, open(p, "w"), indent=0, sort_keys=True)
    if "cfg3_share" in which:
        res["cfg3_share"] = run_case("cfg3_share_pe150_2x8GB", 22_400_000, 4, True)
        json.dump(res, open(p, "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
