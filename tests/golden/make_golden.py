#!/usr/bin/env python3
"""Golden-vector maker.  Runs ONLY in the build container (needs oracle/_ref/{repaq,ref_harness}, i.e. the
reference compiled by oracle/Makefile).  Writes:
  tests/golden/cases.json      expected .rfq (hex) / error text of the reference for every tests/golden/cases.py case
  tests/golden/unit.json       unit-level known answers from the reference's private codec members
  tests/golden/generated.json  md5/size of the reference's .rfq for fqgen-generated inputs (profile, reads, seed, k)
The committed JSON files are data (inputs' hashes + expected outputs); no reference source is stored."""
import hashlib, json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import _oracle as O
from cases import CASES


def harness(args, data=b""):
    return subprocess.run([O.REF_HARNESS] + args, input=data, capture_output=True, check=True).stdout


def main():
    assert O.have_ref(), "build the reference first: make -C oracle"
    out = {}
    for name, c in CASES.items():
        fq1, fq2, p, k = c["fq1"], c.get("fq2", b""), c["paired"], c.get("k", 1000)
        e = {"in_md5": hashlib.md5(fq1 + b"|" + fq2).hexdigest(), "paired": p, "k": k}
        try:
            rfq = O.ref_encode(fq1, fq2, p, k)
            e["rfq_md5"] = hashlib.md5(rfq).hexdigest(); e["rfq_len"] = len(rfq)
            if len(rfq) <= 6000:
                e["rfq_hex"] = rfq.hex()
            try:
                d = O.ref_decode(rfq, p != 0)
                e["decode_md5"] = [hashlib.md5(x).hexdigest() for x in (d if p != 0 else (d,))]
            except O.OracleError:
                e["decode_md5"] = None     # reference crashes / UB on its own output: decode parity unpinned
        except O.OracleError as ex:
            e["error"] = str(ex).strip().replace("ERROR: ", "", 1)
        out[name] = e
    json.dump(out, open(os.path.join(HERE, "cases.json"), "w"), indent=0, sort_keys=True)

    import random, struct
    rng = random.Random(777)
    unit = {"coords": [], "pos": [], "overlap": [], "parse": []}
    coord_sets = [[1000, 1001, 1065, 1130, 1130, 40000, 7, 7, 7], [1000] * 70, [5] * 33 + [6] + [6] * 32, [0, 32767, 32768, 2097151, 1, 65, 66, 130],
                  [rng.randrange(0, 1 << 21) for _ in range(200)], sorted(rng.randrange(1000, 32000) for _ in range(500))]
    for v in coord_sets:
        enc = harness(["coords"], struct.pack("<%dI" % len(v), *v))
        dec = list(struct.unpack("<%dI" % len(v), harness(["decoords", str(len(v))], enc)))
        assert dec == v
        unit["coords"].append({"values": v, "hex": enc.hex()})
    from cases import pos_buffers
    for buf, q in pos_buffers():
        enc = harness(["pos", str(q)], buf)
        unit["pos"].append({"buf_md5": hashlib.md5(buf).hexdigest(), "q": q, "hex": enc.hex()})
    ov_sets = [("ACGTACGTACGTACGTAAAA", "ACGTACGTACGTAAAATTTT"), ("A" * 50, "A" * 50), ("ACGT" * 10, "TTTT" * 10), ("ACGTTGCAACGTTGCA" * 3, "TGCAACGTTGCA" + "G" * 30),
                ("G" * 30 + "ACGTACGTACGTAC", "ACGTACGTACGTAC" + "T" * 30), ("TTTTTTTTTTTTACGTACGTACGTACGT", "CCCCCCCCACGTTTTTTTTTTTTT"), ("ACGTACGTACG", "ACGTACGTACG"), ("ACGTACGTACGT", "ACGTACGTACGT")]
    for a, b in ov_sets:
        unit["overlap"].append({"r1": a, "r2": b, "ov": int(harness(["overlap"], (a + "\n" + b + "\n").encode()))})
    names = ["@A00251:28:H3YV7DSXX:40:1101:2356:1000 1:N:0:TAAGTGGC", "@A:B:C:4:55:66:77 1:N:0:X", "@A:B:C:D:E rest of it", "@A:B:C:1:2:3:4:5:6 tail", "@A:B:C:1:2:3:4", "@A:B:C:1:2:3",
             "@noColonsAtAll", "@ leading space", "@A:B:C:007:0012:+33:-4 x", "@A:B:C:300:70000:99999999999:4294967299 big", "@A:B:C:1:2: 3:\t4 ws", "@A:B:C:::: empty",
             "@A:B:C:1x:2y:3z:4w q", "@a b:c:d:e:f:g:h", "@A:B:C:1:2:3 onlysix:colons", "@:::::::", "@A:B:C:1:2:3:9223372036854775808 of", "@A:B:C:1:2:3:-9223372036854775809 uf"]
    res = harness(["parse"], ("\n".join(names) + "\n").encode()).decode("latin-1").split("\n")
    for n, r in zip(names, res):
        unit["parse"].append({"name": n, "ref": r})
    json.dump(unit, open(os.path.join(HERE, "unit.json"), "w"), indent=0, sort_keys=True)

    # legacy run-length quality coding (src/rfqcodec.cpp:767-824, 919-955): v0.5.1 never WRITES it (App. C Q13), so the images
    # come from the reference's own encodeChunk with BIT_ENCODE_QUAL_BY_COL cleared on the header it made (oracle/ref_harness.cpp rle_image);
    # what the reference binary decodes from them is the golden
    import tempfile
    rle = {}
    RLE_IN = {"d5_tiny_se": (CASES["d5_tiny_se"]["fq1"], b""), "d6_tiny_pe": (CASES["d6_tiny_pe"]["fq1"], CASES["d6_tiny_pe"]["fq2"]),
              "se150_4q": (O.gen(O.NOVA_SE150, 120, seed=61)[0], b""), "se_var_manyN": (O.gen(O.SE_VAR, 150, seed=62, nppm=20000)[0], b""),
              "pe150_manyN": O.gen(O.NOVA_PE150, 80, seed=63, nppm=20000), "bgi_13q": O.gen(O.BGI_PE100, 60, seed=64, n_quals=13), "bgi_40q": O.gen(O.BGI_PE100, 40, seed=65, n_quals=40)}
    for name, (f1, f2) in RLE_IN.items():
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            p1, p2, o = os.path.join(d, "a_1.fq"), os.path.join(d, "a_2.fq"), os.path.join(d, "o.rfq")
            open(p1, "wb").write(f1)
            if f2:
                open(p2, "wb").write(f2)
            subprocess.run([O.REF_HARNESS, "rle_image", p1, p2 if f2 else "-", o], check=True, capture_output=True)
            img = open(o, "rb").read()
        dec = O.ref_decode(img, bool(f2))
        rle[name] = {"rfq_hex": img.hex(), "paired": 1 if f2 else 0, "decode_md5": [hashlib.md5(x).hexdigest() for x in (dec if f2 else (dec,))],
                     "roundtrip": (dec == (f1, f2)) if f2 else (dec == f1)}
        print("rle", name, len(img), rle[name]["roundtrip"])
    json.dump(rle, open(os.path.join(HERE, "rle.json"), "w"), indent=0, sort_keys=True)

    gen = []
    GEN = [  # (label, profile, reads, seed, nppm, nonl, interleaved, n_quals, paired, k)
        ("cfg0_se_var_50k", O.SE_VAR, 50000, 1, 20, 0, False, 13, O.SE, 1000),
        ("cfg1s_se150_100k", O.NOVA_SE150, 100000, 2, 20, 0, False, 13, O.SE, 1000),
        ("cfg1s_se150_100k_highN", O.NOVA_SE150, 100000, 2, 500, 0, False, 13, O.SE, 1000),
        ("cfg1s_se150_100k_k100", O.NOVA_SE150, 100000, 2, 20, 0, False, 13, O.SE, 100),
        ("cfg1s_se150_20k_nonl", O.NOVA_SE150, 20000, 9, 20, 1, False, 13, O.SE, 100),
        ("cfg2s_pe150_60k", O.NOVA_PE150, 60000, 3, 20, 0, False, 13, O.PE_TWO_FILES, 1000),
        ("cfg2s_pe150_60k_interleaved", O.NOVA_PE150, 60000, 3, 20, 0, True, 13, O.PE_INTERLEAVED, 1000),
        ("cfg2s_pe150_20k_k100_nonl2", O.NOVA_PE150, 20000, 4, 500, 2, False, 13, O.PE_TWO_FILES, 100),
        ("cfg4s_bgi_30k", O.BGI_PE100, 30000, 5, 0, 0, False, 13, O.PE_TWO_FILES, 1000),
        ("cfg4s_bgi_30k_q40_nonl", O.BGI_PE100, 30000, 6, 0, 2, False, 40, O.PE_TWO_FILES, 1000),
    ]
    if "--big" in sys.argv:
        GEN += [("cfg1_se150_1GB", O.NOVA_SE150, 2_800_000, 2, 20, 0, False, 13, O.SE, 1000),
                ("cfg1_se150_1GB_highN", O.NOVA_SE150, 2_800_000, 2, 500, 0, False, 13, O.SE, 1000)]
    for (label, prof, reads, seed, nppm, nonl, il, nq, paired, k) in GEN:
        fq1, fq2 = O.gen(prof, reads, seed=seed, nppm=nppm, nonl=nonl, interleaved=il, n_quals=nq)
        rfq = O.ref_encode(fq1, fq2, paired, k)
        d = O.ref_decode(rfq, paired != O.SE)
        ok = (d == (fq1, fq2)) if paired == O.PE_TWO_FILES else ((d[0] + d[1] == b"" or True) if paired == O.PE_INTERLEAVED else d == fq1)
        gen.append({"label": label, "profile": prof, "reads": reads, "seed": seed, "nppm": nppm, "nonl": nonl, "interleaved": il, "n_quals": nq,
                    "paired": paired, "k": k, "fq_md5": hashlib.md5(fq1 + b"|" + fq2).hexdigest(), "fq_bytes": len(fq1) + len(fq2),
                    "rfq_md5": hashlib.md5(rfq).hexdigest(), "rfq_len": len(rfq), "ref_roundtrip": bool(ok)})
        print(label, len(fq1) + len(fq2), len(rfq), ok)
    old = []
    p = os.path.join(HERE, "generated.json")
    if os.path.exists(p) and "--big" not in sys.argv:
        old = [e for e in json.load(open(p)) if e["label"] not in {g["label"] for g in gen}]
    json.dump(gen + old, open(p, "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
