"""CPU: the oracle (oracle/rfq_oracle.c) against the committed golden vectors that the REFERENCE produced
(tests/golden/make_golden.py, run in the build container against oracle/_ref).  This is what pins the oracle."""
import hashlib
import json
import os
import struct

import pytest

import _oracle as O
from cases import CASES, pos_buffers

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES_J = json.load(open(os.path.join(G, "cases.json")))
UNIT_J = json.load(open(os.path.join(G, "unit.json")))
GEN_J = json.load(open(os.path.join(G, "generated.json")))


def test_fastqmeta_reference_known_answer():
    # the reference's own (only) unit test: src/fastqmeta.cpp:82-109
    ok, n1, lane, tile, x, y, n2 = O.parse_name(b"@A00251:28:H3YV7DSXX:40:1101:2356:1000 1:N:0:TAAGTGGC")
    assert (ok, n1, lane, tile, x, y, n2) == (1, b"@A00251:28:H3YV7DSXX", 40, 1101, 2356, 1000, b" 1:N:0:TAAGTGGC")


def test_survey_appendix_d_vectors():
    assert O.encode_coords([1000, 1001, 1065, 1130, 1130, 40000, 7, 7, 7]).hex() == "c080bf046ac0e09c400007c1"
    buf, q = pos_buffers()[0]
    assert O.pos_encode(buf, q).hex() == "0000c180c7c05fdfc6e0004df8"
    assert O.overlap(b"ACGTACGTACGTACGTAAAA", b"ACGTACGTACGTAAAATTTT") == 16
    rfq = O.encode_file(CASES["d5_tiny_se"]["fq1"])
    assert len(rfq) == 130 and hashlib.md5(rfq).hexdigest() == "a77cc57c8d01c2da31bfce69b8a98f66"
    pe = O.encode_file(CASES["d6_tiny_pe"]["fq1"], CASES["d6_tiny_pe"]["fq2"], O.PE_TWO_FILES)
    assert len(pe) == 141 and hashlib.md5(pe).hexdigest() == "174bda4b113bea3c16f86068b318a62e"


@pytest.mark.parametrize("name", sorted(CASES))
def test_case_matches_reference_golden(name):
    c, g = CASES[name], CASES_J[name]
    fq1, fq2 = c["fq1"], c.get("fq2", b"")
    assert hashlib.md5(fq1 + b"|" + fq2).hexdigest() == g["in_md5"], "case generator drifted; rerun make_golden.py"
    if "error" in g:
        with pytest.raises(O.OracleError) as e:
            O.encode_file(fq1, fq2, c["paired"], c.get("k", 1000) * 1000)
        assert str(e.value).strip() == g["error"]
        return
    rfq = O.encode_file(fq1, fq2, c["paired"], c.get("k", 1000) * 1000)
    assert len(rfq) == g["rfq_len"] and hashlib.md5(rfq).hexdigest() == g["rfq_md5"]
    if "rfq_hex" in g:
        assert rfq.hex() == g["rfq_hex"]
    if g.get("decode_md5"):
        d = O.decode_file(rfq, c["paired"] != 0)
        d = d if c["paired"] != 0 else (d,)
        assert [hashlib.md5(x).hexdigest() for x in d] == g["decode_md5"]


def test_unit_vectors():
    for e in UNIT_J["coords"]:
        enc = O.encode_coords(e["values"])
        assert enc.hex() == e["hex"]
        assert O.decode_coords(enc, len(e["values"])) == e["values"]
    bufs = pos_buffers()
    assert len(bufs) == len(UNIT_J["pos"])
    for (buf, q), e in zip(bufs, UNIT_J["pos"]):
        assert hashlib.md5(buf).hexdigest() == e["buf_md5"]
        enc = O.pos_encode(buf, q)
        assert enc.hex() == e["hex"]
        base = bytes(b"F" if q != ord("F") else b"G") * len(buf)
        dec = O.pos_decode(enc, q, base)
        assert all((dec[i] == q) == (buf[i] == q) for i in range(len(buf)))
    for e in UNIT_J["overlap"]:
        assert O.overlap(e["r1"].encode(), e["r2"].encode()) == e["ov"]
    for e in UNIT_J["parse"]:
        ok, n1, lane, tile, x, y, n2 = O.parse_name(e["name"].encode("latin-1"))
        got = "%d|%s|%d|%d|%d|%d|%s" % (ok, n1.decode("latin-1"), lane, tile, x, y, n2.decode("latin-1"))
        assert got == e["ref"]


@pytest.mark.parametrize("e", [g for g in GEN_J if g["fq_bytes"] < 64_000_000], ids=lambda g: g["label"])
def test_generated_config_md5(e):
    fq1, fq2 = O.gen(e["profile"], e["reads"], seed=e["seed"], nppm=e["nppm"], nonl=e["nonl"], interleaved=e["interleaved"], n_quals=e["n_quals"])
    assert hashlib.md5(fq1 + b"|" + fq2).hexdigest() == e["fq_md5"]
    rfq = O.encode_file(fq1, fq2, e["paired"], max(100, e["k"]) * 1000)
    assert len(rfq) == e["rfq_len"] and hashlib.md5(rfq).hexdigest() == e["rfq_md5"]
    d = O.decode_file(rfq, e["paired"] != O.SE)
    if e["paired"] == O.PE_TWO_FILES:
        assert d == (fq1, fq2)            # the oracle keeps every read (see rfq_oracle.c note on decompressPE)
    elif e["paired"] == O.SE:
        assert d == fq1
    else:
        assert O.decode_file(rfq, False) == fq1


@pytest.mark.skipif(not O.have_ref(), reason="reference binary only exists in the build container")
def test_oracle_equals_reference_binary_randomised():
    import random
    rng = random.Random(99)
    for it in range(6):
        prof = rng.choice([O.NOVA_SE150, O.NOVA_PE150, O.SE_VAR, O.BGI_PE100])
        reads = rng.randrange(500, 9000)
        nonl = rng.randrange(4); il = prof in (O.NOVA_PE150, O.BGI_PE100) and rng.random() < 0.3
        fq1, fq2 = O.gen(prof, reads, seed=rng.randrange(1 << 30), nppm=rng.choice([0, 20, 500, 5000]), nonl=nonl, interleaved=il, n_quals=rng.randrange(13, 41))
        paired = O.SE if prof in (O.NOVA_SE150, O.SE_VAR) else (O.PE_INTERLEAVED if il else O.PE_TWO_FILES)
        k = rng.choice([100, 137, 1000])
        assert O.encode_file(fq1, fq2, paired, k * 1000) == O.ref_encode(fq1, fq2, paired, k)


def test_oracle_decodes_legacy_run_length_images_like_the_reference():
    """decodeQualByRunLenCoding (src/rfqcodec.cpp:919-955) against what the reference binary decoded."""
    import hashlib
    import _engine as E
    for name, g in E.rle_goldens().items():
        img = bytes.fromhex(g["rfq_hex"]); split = bool(g["paired"])
        got = O.decode_file(img, split)
        assert [hashlib.md5(x).hexdigest() for x in (got if split else (got,))] == g["decode_md5"], name
        assert g["roundtrip"]


COMPAT_J = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "compat.json")))


@pytest.mark.parametrize("name", sorted(COMPAT_J))
def test_bug_compat_decode_equals_the_reference_binarys_output(name):
    """rfqo_decode_file_compat - the reference's decompress loops as they stand, data loss included - against what the reference BINARY wrote for the
    same image (tests/golden/compat.json, made by make_golden_compat.py); and against the binary itself where it exists."""
    g = COMPAT_J[name]
    fq1, fq2 = O.gen(g["profile"], g["reads"], seed=g["seed"], **g["kw"])
    rfq = O.encode_file(fq1, fq2, g["paired"], 100_000)
    assert hashlib.md5(rfq).hexdigest() == g["rfq_md5"]
    split = g.get("split", g["paired"] != O.SE)                                           # (a paired image decoded to ONE output: Repaq::decompress, which loses nothing)
    got = O.decode_file(rfq, split_pe=split, bug_compat=True)
    texts = got if split else (got,)
    assert [len(t) for t in texts] == g["ref_decode_len"] and [hashlib.md5(t).hexdigest() for t in texts] == g["ref_decode_md5"]
    keep = O.decode_file(rfq, split_pe=split)
    if split or g["paired"] == O.SE:
        assert (list(keep) if split else [keep]) == ([fq1, fq2] if split else [fq1])      # the default keeps every read
        assert g["ref_roundtrip"] == (list(texts) == ([fq1, fq2] if split else [fq1]))
    else:
        assert keep == got and len(got) == sum(len(f) + (0 if f.endswith(b"\n") else 1) for f in (fq1, fq2)) - 1   # both mates of every pair, only the last '\n' dropped
    if name == "se_nonl_small":
        assert g["flagged_not_last"] >= 1 and got == fq1                                  # (flagged chunks that are not the last, and nothing lost)
    if O.have_ref():
        assert O.ref_decode(rfq, split_pe=split) == got


def test_oracle_equals_reference_binary_on_almost_uniform_read_lengths():
    """tests/_shapes.py (reads of one length, or of one length but for one read, or of one length per file only) - what the GPU suite checks the encoder's
    closed forms with: the oracle's image of each against the reference binary's, where the binary exists (this container)"""
    if not O.have_ref():
        pytest.skip("the reference binary does not travel with -m 'not gpu' on a box without /root/reference")
    import _shapes as SH
    for label, fq1, fq2, paired, cb in SH.cases(2400, 100000):
        assert O.ref_encode(fq1, fq2, paired, k=cb // 1000) == O.encode_file(fq1, fq2, paired, cb), label
