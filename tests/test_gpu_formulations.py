"""GPU (MI355X): EVERY alternative formulation of the HIP path, forced by its rfq_set_option switch, against the reference's goldens.

The product picks a formulation by the data (reads too long for a tile -> byte-wise gather, many quality values -> list coder, no chunk index -> guess and
verify, a streaming caller's non-final slice -> materialising decode, ...), so on the hardware a fallback otherwise runs only when an input happens to need
it - and the SIMT interpreter, where tests/test_emu_*.py force the same switches, cannot see a missing stream dependency, a stale entry state or a DPP
problem (DESIGN.md §6 lesson vii).  Here every switch of tests/_engine.ENC_FORMS / DEC_FORMS runs on six inputs - SE variable lengths, PE150 interleaved
input, PE150 two files with N and an unterminated R2, the configs[4] shape with forty quality values, '\\r\\n' text, reads of 12 - 40 kB - plus the legacy
run-length images: encode == the reference's image (md5 from the reference binary: tests/golden/generated.json, formulations.json), decode == what the
reference decodes.  Nothing here reads /root/reference."""
import functools
import hashlib
import json
import os

import pytest

import _engine as E
import _oracle as O
from formulation_inputs import INPUTS as EXTRA

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GEN_J = {g["label"]: g for g in json.load(open(os.path.join(G, "generated.json")))}
FORM_J = json.load(open(os.path.join(G, "formulations.json")))
GENERATED = ["cfg0_se_var_50k", "cfg2s_pe150_60k_interleaved", "cfg2s_pe150_20k_k100_nonl2", "cfg4s_bgi_30k_q40_nonl"]
INPUT_NAMES = GENERATED + sorted(EXTRA)


@pytest.fixture(scope="module")
def codec():
    import torch
    assert torch.cuda.is_available()
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.PRODUCT_LIB)
    assert "gfx950" in c.version()
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _options_back_to_default(codec):
    yield
    E.reset_options(codec)


@functools.lru_cache(maxsize=None)
def _input(name):
    """(fq1, fq2, paired, chunk_bases, golden md5, golden size, what the reference decodes from its image [md5s or None = the input itself])"""
    if name in EXTRA:
        build, paired, k = EXTRA[name]; fq = build(); g = FORM_J[name]
        assert hashlib.md5(fq).hexdigest() == g["in_md5"]
        return fq, b"", paired, max(100, k) * 1000, g["rfq_md5"], g["rfq_len"], [g["decode_md5"]]
    e = GEN_J[name]
    fq1, fq2 = O.gen(e["profile"], e["reads"], seed=e["seed"], nppm=e["nppm"], nonl=e["nonl"], interleaved=e["interleaved"], n_quals=e["n_quals"])
    assert hashlib.md5(fq1 + b"|" + fq2).hexdigest() == e["fq_md5"]
    return fq1, fq2, e["paired"], max(100, e["k"]) * 1000, e["rfq_md5"], e["rfq_len"], None


@functools.lru_cache(maxsize=None)
def _image(name):
    fq1, fq2, paired, cb, md5, n, _ = _input(name)
    rfq = O.encode_file(fq1, fq2, paired, cb)                                # (the oracle's image; it must be the reference's)
    assert len(rfq) == n and hashlib.md5(rfq).hexdigest() == md5
    return rfq


@pytest.mark.parametrize("form", sorted(E.ENC_FORMS))
@pytest.mark.parametrize("name", INPUT_NAMES)
def test_encode_formulation_equals_reference_golden(codec, name, form):
    fq1, fq2, paired, cb, md5, n, _ = _input(name)
    if name == "long_reads" and E.ENC_FORMS[form][1] == "!quality_masks":
        pytest.skip("reads that do not fit a tile take the byte-wise gather: no k_gather2 to switch")
    if name == "long_reads":                                                 # (the byte-wise gather whatever the switch says: that IS the point of the input)
        opts, _ = E.ENC_FORMS[form]
        with E._Options(codec, {k: (str(max(4096, int(3 * 2.6 * cb))) if v is None else v) for k, v in opts.items()}):
            got = E.encode(codec, fq1, fq2, paired, cb)
            assert "gather_bytes" in dict(codec.timings())
        assert len(got) == n and hashlib.md5(got).hexdigest() == md5
        return
    E.check_encode_formulation(codec, form, fq1, fq2, paired, cb, want_md5=md5, want_len=n)


@pytest.mark.parametrize("form", sorted(E.DEC_FORMS))
@pytest.mark.parametrize("name", INPUT_NAMES)
def test_decode_formulation_equals_reference(codec, name, form):
    fq1, fq2, paired, cb, _, _, dec_md5 = _input(name)
    rfq = _image(name); split = paired == O.PE_TWO_FILES
    want = O.decode_file(rfq, split)
    if dec_md5 is None:                                                      # generated configs: the reference round-trips them (ref_roundtrip in generated.json)
        assert want == ((fq1, fq2) if split else (fq1 if paired == O.SE else want))
    else:
        assert [hashlib.md5(want).hexdigest()] == dec_md5                    # what the reference binary decoded from its own image
    if name == "long_reads":                                                 # reads of more than 2000 bases: the expanded path whatever the switch says
        opts, _ = E.DEC_FORMS[form]
        with E._Options(codec, {k: (str(int(2.5 * cb)) if v is None else v) for k, v in opts.items()}):
            assert codec.decode_bytes(rfq, split_pe=split) == want
            assert "emit_expanded" in dict(codec.timings())
        return
    E.check_decode_formulation(codec, form, rfq, split, want, cb)
    if split and form in ("materialise", "walk_exact"):                      # Repaq::decompress on a PE file: one interleaved stream
        E.check_decode_formulation(codec, form, rfq, False, O.decode_file(rfq, False), cb)


@pytest.mark.parametrize("form", sorted(E.DEC_FORMS))
def test_legacy_run_length_images_under_every_decode_formulation(codec, form):
    opts, _ = E.DEC_FORMS[form]
    with E._Options(codec, {k: ("150000" if v is None else v) for k, v in opts.items()}):
        for name, g in sorted(E.rle_goldens().items()):
            E.check_rle_decode(codec, name, g)
            assert "emit_expanded" in dict(codec.timings())


def test_interleaved_chunk_whose_mate_test_fails_midway_under_every_encode_formulation(codec):
    """phase 2 of the gather (k_gather_redo_reset + the second k_gather2 launch): a chunk whose interleave test fails behind its first pairs is gathered
    once more with the mates as they stand - under every encode formulation"""
    fq1, fq2 = O.gen(O.NOVA_PE150, 3000, seed=23)
    lines = fq2.split(b"\n")
    for k in (700, 2100):                                                   # an R2 name that is not its R1's: canBePeInterleaved flips mid-chunk (src/rfqcodec.cpp:233-270)
        lines[4 * k] = lines[4 * k].replace(b":", b";", 1)
    fq2 = b"\n".join(lines)
    want = O.encode_file(fq1, fq2, O.PE_TWO_FILES, 100_000)
    for form in sorted(E.ENC_FORMS):
        E.check_encode_formulation(codec, form, fq1, fq2, O.PE_TWO_FILES, 100_000, want=want)
    for form in sorted(E.DEC_FORMS):
        E.check_decode_formulation(codec, form, want, True, O.decode_file(want, True), 100_000)


def test_option_values_are_validated_and_restored(codec):
    from repaq_amd import RfqError
    for name, bad in (("RFQ_STREAMS", "foo"), ("RFQ_STREAMS", "3"), ("RFQ_SLICE_BYTES", "-5"), ("RFQ_SLICE_BYTES", "12x"), ("RFQ_MATERIALISE", "abc"),
                      ("RFQ_G2_PAD", "-1"), ("RFQ_GW_SHIFT", "99"), ("RFQ_NO_SUCH_SWITCH", "1"), ("RFQ_IDX_TILES", "5")):
        with pytest.raises(RfqError) as e:
            codec.set_option(name, bad)
        assert e.value.code == -3
    codec.set_option("RFQ_WALK", "exact")
    with codec.option("RFQ_WALK", "guess"):
        assert codec.get_option("RFQ_WALK") == ""
    assert codec.get_option("RFQ_WALK") == "exact"                          # (what was there before, not the built-in default)
    assert set(codec.option_names()) >= {"RFQ_GATHER", "RFQ_G2_PAD", "RFQ_SP_PAD"}
