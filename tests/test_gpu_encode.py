"""GPU (MI355X): the hand-written HIP encode path, through the C-ABI, against the reference's golden vectors and the
oracle — bit-exact.  Nothing here reads /root/reference."""
import hashlib
import json
import os

import pytest

import _engine as E
import _oracle as O
from cases import CASES

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES_J = json.load(open(os.path.join(G, "cases.json")))
GEN_J = json.load(open(os.path.join(G, "generated.json")))


@pytest.fixture(scope="module")
def codec():
    import torch
    assert torch.cuda.is_available()
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.PRODUCT_LIB)      # the product library: fails loudly if it is missing
    assert "gfx950" in c.version()
    yield c
    c.close()


@pytest.mark.parametrize("name", sorted(CASES))
def test_case_matches_reference_golden(codec, name):
    E.check_case(codec, name, CASES[name], CASES_J[name])


@pytest.mark.parametrize("e", [g for g in GEN_J if g["fq_bytes"] < 64_000_000], ids=lambda g: g["label"])
def test_generated_config_md5_equals_reference(codec, e):
    fq1, fq2 = O.gen(e["profile"], e["reads"], seed=e["seed"], nppm=e["nppm"], nonl=e["nonl"], interleaved=e["interleaved"], n_quals=e["n_quals"])
    assert hashlib.md5(fq1 + b"|" + fq2).hexdigest() == e["fq_md5"]
    got = E.encode(codec, fq1, fq2, e["paired"], max(100, e["k"]) * 1000)
    assert len(got) == e["rfq_len"] and hashlib.md5(got).hexdigest() == e["rfq_md5"]


SMALL_CHUNK = [
    ("se150_cb20000", O.NOVA_SE150, 20000, 2, 20000, O.SE, {}),
    ("se_var_cb15000", O.SE_VAR, 20000, 3, 15000, O.SE, {}),
    ("pe150_cb33333", O.NOVA_PE150, 10000, 4, 33333, O.PE_TWO_FILES, dict(nppm=2000)),
    ("pe150_interleaved_cb50000", O.NOVA_PE150, 10000, 4, 50000, O.PE_INTERLEAVED, dict(interleaved=True)),
    ("bgi_q40_cb10000", O.BGI_PE100, 8000, 5, 10000, O.PE_TWO_FILES, dict(n_quals=40)),
    ("bgi_q13_nonl_cb77777", O.BGI_PE100, 8000, 6, 77777, O.PE_TWO_FILES, dict(n_quals=13, nonl=3)),
]


@pytest.mark.parametrize("label,prof,reads,seed,cb,paired,kw", SMALL_CHUNK, ids=[m[0] for m in SMALL_CHUNK])
def test_many_small_chunks_match_oracle(codec, label, prof, reads, seed, cb, paired, kw):
    fq1, fq2 = O.gen(prof, reads, seed=seed, **kw)
    assert E.encode(codec, fq1, fq2, paired, cb) == O.encode_file(fq1, fq2, paired, cb)


def test_full_size_se150_1gb_md5_equals_reference(codec):
    """BASELINE.json configs[1]: 1 GB NovaSeq SE150, -k 1000; golden md5 produced by the reference binary (make_golden.py --big)."""
    gold = [g for g in GEN_J if g["label"] == "cfg1_se150_1GB"]
    if not gold:
        pytest.skip("no full-size golden committed")
    e = gold[0]
    fq1, _ = O.gen(e["profile"], e["reads"], seed=e["seed"], nppm=e["nppm"])
    assert hashlib.md5(fq1 + b"|").hexdigest() == e["fq_md5"]
    got = E.encode(codec, fq1, b"", O.SE, 1_000_000)
    assert len(got) == e["rfq_len"] and hashlib.md5(got).hexdigest() == e["rfq_md5"]
    # size-independent property: the chunk table parses and every chunk but the last holds >= 1,000,000 bases
    offs = O.chunk_table(got)
    assert offs[-1] == len(got) and len(offs) - 1 == (e["reads"] * 150 + 1_000_049) // 1_000_050


def test_batched_encode_equals_one_shot(codec):
    fq1, _ = O.gen(O.NOVA_SE150, 60000, seed=21)
    cb = 300_000
    want = O.encode_file(fq1, b"", O.SE, cb)
    codec.clearHeader()
    out = b""; pos = 0; step = 5_000_000; first = True
    while pos < len(fq1):
        end = min(len(fq1), pos + step); final = end == len(fq1)
        buf = fq1[pos:end]
        if not final:
            cut = buf.rfind(b"\n@A00250")      # record boundary of the NovaSeq profile
            buf = buf[: cut + 1]
        d = codec.dev_put(buf)
        r = codec.encode(d, len(buf), None, 0, O.SE, cb, final=final, emit_header=first, file_off1=pos)
        out += codec.dev_get(r.d_rfq, r.rfq_len) if r.rfq_len else b""
        codec.dev_free(d)
        pos += r.consumed1 if not final else len(buf)
        first = False
    assert out == want


def test_caller_buffer_and_nospace(codec):
    from repaq_amd import RfqError
    fq1, _ = O.gen(O.NOVA_SE150, 5000, seed=5)
    want = O.encode_file(fq1, b"", O.SE, 1_000_000)
    d = codec.dev_put(fq1); o = codec.dev_put(b"\0" * (len(want) + 1000))
    codec.clearHeader()
    r = codec.encode(d, len(fq1), None, 0, O.SE, 1_000_000, d_out=o, out_cap=len(want) + 1000)
    assert r.d_rfq == o.value and codec.dev_get(o, r.rfq_len) == want
    codec.clearHeader()
    with pytest.raises(RfqError) as e:
        codec.encode(d, len(fq1), None, 0, O.SE, 1_000_000, d_out=o, out_cap=len(want) // 2)
    assert e.value.code == -8
    codec.dev_free(d); codec.dev_free(o)


@pytest.mark.parametrize("label", ["se_nonl", "pe_nonl2", "se_crlf", "interleaved"])
def test_scan_then_chunk_parallel_encode_equals_one_shot(label):
    """rfq_scan_batch plans the chunk ends; separate contexts encode chunk ranges with flush_all; the concatenation is the one-shot image."""
    from repaq_amd import RfqCodec, PE_TWO_FILES, PE_INTERLEAVED, SE
    mk = lambda: RfqCodec(device=0, library=None)
    if label == "se_nonl":
        fq1, fq2 = O.gen(O.NOVA_SE150, 3000, seed=5, nonl=1); paired = SE
    elif label == "pe_nonl2":
        fq1, fq2 = O.gen(O.NOVA_PE150, 2000, seed=6, nonl=2); paired = PE_TWO_FILES
    elif label == "se_crlf":
        fq1, fq2 = O.gen(O.SE_VAR, 2500, seed=7); fq1 = fq1.replace(b"\n", b"\r\n"); paired = SE
    else:
        fq1, fq2 = O.gen(O.NOVA_PE150, 1500, seed=8, interleaved=True); paired = PE_INTERLEAVED
    for parts in (2, 5):
        got, nc = E.scan_and_encode_in_ranges(mk, fq1, fq2, paired, 100_000, parts)
        assert nc >= 3 and got == O.encode_file(fq1, fq2, paired, 100_000)


def test_overlap_search_paths(codec):
    """k_overlap: packed 2-bit rows, the > 256-base path and the odd-character path (see tests/_engine.py)."""
    E.overlap_search_paths(codec)


def test_overlap_search_paths_short_rows(codec):
    """the same adversarial pairs with no read above 160 bases: the launch takes the 160-base row geometry (k_overlap<true, 160>)"""
    E.overlap_search_paths(codec, lengths=(150, 150, 151, 100, 145, 40, 13, 12, 11), seed=77)


import _shapes as SH

_SHAPES = SH.cases(9000, 100000)


@pytest.mark.parametrize("label,fq1,fq2,paired,cb", _SHAPES, ids=[c[0] for c in _SHAPES])
def test_uniform_and_almost_uniform_read_lengths(codec, label, fq1, fq2, paired, cb):
    """closed-form prefixes / cuts where every read has one length, the scans everywhere else - and nothing in between (tests/_shapes.py);
    -k 100 is a chunk size the reference binary takes: where it travelled, the oracle's image is checked against its own first"""
    want = O.encode_file(fq1, fq2, paired, cb)
    if O.have_ref():
        assert O.ref_encode(fq1, fq2, paired, k=cb // 1000) == want
    assert E.encode(codec, fq1, fq2, paired, cb) == want
