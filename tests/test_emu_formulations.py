"""CPU: the forced-formulation checks of tests/test_gpu_formulations.py (tests/_engine.ENC_FORMS / DEC_FORMS) at interpreter sizes, against the oracle - so that the
switches, their markers and the helper logic are exercised where there is no GPU.  Parity proper is the GPU file."""
import pytest

import _engine as E
import _oracle as O
from formulation_inputs import crlf_mid, long_reads


@pytest.fixture(scope="module")
def codec():
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.build_emu())
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _options_back_to_default(codec):
    yield
    E.reset_options(codec)


def _inputs():
    yield "se_var", O.gen(O.SE_VAR, 500, seed=3) + (O.SE, 15000)
    yield "pe150_interleaved_in", O.gen(O.NOVA_PE150, 260, seed=4, interleaved=True) + (O.PE_INTERLEAVED, 20000)
    yield "pe150_nonl2_manyN", O.gen(O.NOVA_PE150, 260, seed=7, nonl=2, nppm=3000) + (O.PE_TWO_FILES, 9000)
    yield "bgi_q40", O.gen(O.BGI_PE100, 200, seed=5, n_quals=40) + (O.PE_TWO_FILES, 10000)
    yield "crlf", (crlf_mid(n=180), b"", O.SE, 9000)


INPUTS = dict(_inputs())


@pytest.mark.parametrize("form", sorted(E.ENC_FORMS))
@pytest.mark.parametrize("name", sorted(INPUTS))
def test_encode_formulation_matches_oracle(codec, name, form):
    fq1, fq2, paired, cb = INPUTS[name]
    E.check_encode_formulation(codec, form, fq1, fq2, paired, cb, want=O.encode_file(fq1, fq2, paired, cb))


@pytest.mark.parametrize("form", sorted(E.DEC_FORMS))
@pytest.mark.parametrize("name", sorted(INPUTS))
def test_decode_formulation_matches_oracle(codec, name, form):
    fq1, fq2, paired, cb = INPUTS[name]
    rfq = O.encode_file(fq1, fq2, paired, cb); split = paired == O.PE_TWO_FILES
    E.check_decode_formulation(codec, form, rfq, split, O.decode_file(rfq, split), cb)


def test_reads_too_long_for_a_tile_take_the_bytewise_gather_and_the_expanded_decode(codec):
    fq = long_reads(n=10)
    want = O.encode_file(fq, b"", O.SE, 100_000)
    assert E.encode(codec, fq, b"", O.SE, 100_000) == want
    assert "gather_bytes" in dict(codec.timings())
    assert codec.decode_bytes(want) == O.decode_file(want)
    assert "emit_expanded" in dict(codec.timings())


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
    assert codec.get_option("RFQ_WALK") == "exact"
    assert set(codec.option_names()) >= {"RFQ_GATHER", "RFQ_G2_PAD", "RFQ_SP_PAD"}
