"""GPU (MI355X): the hand-written HIP decode path, through the C-ABI, against the oracle and the original FASTQ — byte-exact."""
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
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.PRODUCT_LIB)
    assert "gfx950" in c.version()
    yield c
    c.close()


def _oracle_rfq(case):
    try:
        return O.encode_file(case["fq1"], case.get("fq2", b""), case["paired"], case.get("k", 1000) * 1000)
    except O.OracleError:
        return None


DECODABLE = sorted(n for n in CASES if n != "se_name_over_255" and _oracle_rfq(CASES[n]) is not None)


@pytest.mark.parametrize("name", DECODABLE)
def test_case_decodes_into_the_callers_buffers(codec, name):
    """the same through rfq_decode_args.d_out1 / d_out2 - the emitter launched ahead of the host's look at the status -, and into buffers one byte too small"""
    from repaq_amd import RfqError
    rfq = _oracle_rfq(CASES[name]); split = CASES[name]["paired"] != 0
    want = O.decode_file(rfq, split); w1, w2 = (want if split else (want, b""))
    assert codec.decode_bytes(rfq, split_pe=split, out_caps=(len(w1) + 64, len(w2) + 64)) == want
    if len(w1) > 1:
        with pytest.raises(RfqError):
            codec.decode_bytes(rfq, split_pe=split, out_caps=(len(w1) - 1, len(w2) + 64))
        assert codec.decode_bytes(rfq, split_pe=split, out_caps=(len(w1) + 64, len(w2) + 64)) == want      # (the context is as good as new behind the refusal)


@pytest.mark.parametrize("name", DECODABLE)
def test_case_decodes_like_reference(codec, name):
    rfq = _oracle_rfq(CASES[name]); split = CASES[name]["paired"] != 0
    got = codec.decode_bytes(rfq, split_pe=split)
    assert got == O.decode_file(rfq, split)
    g = CASES_J[name].get("decode_md5")
    if g:   # what the reference binary itself decoded (make_golden.py)
        assert [hashlib.md5(x).hexdigest() for x in (got if split else (got,))] == g
    if split:
        assert codec.decode_bytes(rfq, split_pe=False) == O.decode_file(rfq, False)


@pytest.mark.parametrize("e", [g for g in GEN_J if g["fq_bytes"] < 64_000_000], ids=lambda g: g["label"])
def test_generated_config_round_trip(codec, e):
    fq1, fq2 = O.gen(e["profile"], e["reads"], seed=e["seed"], nppm=e["nppm"], nonl=e["nonl"], interleaved=e["interleaved"], n_quals=e["n_quals"])
    rfq = E.encode(codec, fq1, fq2, e["paired"], max(100, e["k"]) * 1000)
    assert hashlib.md5(rfq).hexdigest() == e["rfq_md5"]
    if e["paired"] == O.PE_TWO_FILES:
        assert codec.decode_bytes(rfq, split_pe=True) == (fq1, fq2)
    else:
        assert codec.decode_bytes(rfq, split_pe=False) == fq1


SMALL_CHUNK = [
    ("se_var_cb15000", O.SE_VAR, 20000, 3, 15000, O.SE, {}),
    ("pe150_cb33333", O.NOVA_PE150, 10000, 4, 33333, O.PE_TWO_FILES, dict(nppm=2000)),
    ("bgi_q40_cb10000", O.BGI_PE100, 8000, 5, 10000, O.PE_TWO_FILES, dict(n_quals=40)),
]


@pytest.mark.parametrize("label,prof,reads,seed,cb,paired,kw", SMALL_CHUNK, ids=[m[0] for m in SMALL_CHUNK])
def test_many_small_chunks_round_trip(codec, label, prof, reads, seed, cb, paired, kw):
    fq1, fq2 = O.gen(prof, reads, seed=seed, **kw)
    rfq = O.encode_file(fq1, fq2, paired, cb)
    d = codec.decode_bytes(rfq, split_pe=(paired != O.SE))
    assert d == ((fq1, fq2) if paired != O.SE else fq1)


def test_full_size_se150_1gb_round_trip_on_device(codec):
    """BASELINE.json configs[1] at full size: encode -> decode entirely in HBM; md5 of the FASTQ that comes back."""
    import torch
    gold = [g for g in GEN_J if g["label"] == "cfg1_se150_1GB"]
    if not gold:
        pytest.skip("no full-size golden committed")
    e = gold[0]
    fq1, _ = O.gen(e["profile"], e["reads"], seed=e["seed"], nppm=e["nppm"])
    t = torch.frombuffer(bytearray(fq1), dtype=torch.uint8).cuda()
    codec.clearHeader()
    r = codec.encode(t.data_ptr(), len(fq1), None, 0, O.SE, 1_000_000)
    assert r.rfq_len == e["rfq_len"]
    d = codec.decode(r.d_rfq, r.rfq_len)
    assert d.n1 == len(fq1) and d.n_reads == e["reads"]
    back = codec.dev_get(d.d_fq1, d.n1)
    assert hashlib.md5(back).hexdigest() == hashlib.md5(fq1).hexdigest()


def test_pe_2x_round_trip_mid_size(codec):
    """configs[2]-shaped (PE150, -i/-I) at 2 x 107 MB: encode + decode round trip on one GPU."""
    fq1, fq2 = O.gen(O.NOVA_PE150, 300000, seed=3)
    rfq = E.encode(codec, fq1, fq2, O.PE_TWO_FILES, 1_000_000)
    assert rfq == O.encode_file(fq1, fq2, O.PE_TWO_FILES, 1_000_000)
    assert codec.decode_bytes(rfq, split_pe=True) == (fq1, fq2)


@pytest.mark.parametrize("step", [700, 5000, 60000])
def test_decode_in_slices_equals_one_shot(codec, step):
    """rfq_decode_batch on successive byte ranges (a chunk cut by the range end is carried into the next call)."""
    fq1, fq2 = O.gen(O.NOVA_PE150, 400, seed=71, nonl=2)
    rfq = O.encode_file(fq1, fq2, O.PE_TWO_FILES, 20000)
    assert E.decode_in_slices(codec, rfq, True, step) == (fq1, fq2)
    se, _ = O.gen(O.SE_VAR, 500, seed=72, nonl=1)
    rfq = O.encode_file(se, b"", O.SE, 15000)
    assert E.decode_in_slices(codec, rfq, False, step) == se


def _first_diff_cases():
    import random
    rng = random.Random(9)
    base = bytes(rng.randrange(256) for _ in range(70_001))
    yield base, base, 0                                   # identical (odd length: 16-byte body + byte tail)
    for at in (0, 1, 15, 16, 17, 4095, 4096, 65_535, 69_999, 70_000):
        b = bytearray(base); b[at] ^= 0x40
        if at < 60_000:
            b[at + 3000] ^= 1                               # a later difference must not win
        yield base, bytes(b), 0
    b = bytearray(base); b[333] ^= 2
    yield base, bytes(b), 5                               # mis-aligned operands take the byte path
    yield b"", b"", 0


def test_first_diff_of_two_device_texts(codec):
    """rfq_compare_bytes (--compare on the device): first differing offset, n when identical."""
    for a, b, skew in _first_diff_cases():
        n = len(a) - skew
        want = next((i for i in range(n) if a[skew + i] != b[i]), n)
        da, db = codec.dev_put(a), codec.dev_put(b)
        try:
            assert codec.first_diff(da.value + skew, db, n) == want
        finally:
            codec.dev_free(da); codec.dev_free(db)


def test_decode_with_chunk_index(codec):
    """rfq_decode_args.h_chunk_off: verified on the device, ignored when it does not verify (see tests/_engine.py)."""
    E.decode_with_chunk_index(codec)


@pytest.mark.parametrize("name", sorted(E.rle_goldens()))
def test_legacy_run_length_quality_images_decode_like_the_reference(codec, name):
    E.check_rle_decode(codec, name, E.rle_goldens()[name])
