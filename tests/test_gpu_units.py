"""GPU (MI355X): the reference's private-member known answers - encodeCoords, encodeSingleQualByCol, overlap, FastqMeta::parse (tests/golden/unit.json, made by
oracle/ref_harness.cpp from the reference's own members) - against the HIP kernels at the boundary: each vector is a small FASTQ, the C-ABI encodes it, and the section of
the image that holds that member's output verbatim is compared (tests/_units.py, tests/_sections.py).  Whole-image goldens cannot tell two cancelling errors inside one
section from none; these can.  Default formulation and the switches that swap the kernel under test."""
import pytest

import _engine as E
import _units as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.PRODUCT_LIB)
    assert "gfx950" in c.version()
    yield c
    c.close()


FORMS = [{}, {"RFQ_GATHER": "old"}, {"RFQ_QUAL": "bytes"}, {"RFQ_QUAL": "bytes", "RFQ_CODER": "list"}, {"RFQ_STREAMS": "1"}]
IDS = [("+".join("%s=%s" % kv for kv in o.items()) or "default") for o in FORMS]


@pytest.mark.parametrize("opts", FORMS, ids=IDS)
def test_k_coords_writes_encodeCoords_streams(codec, opts):
    assert U.check_coords(codec, opts) == 6


@pytest.mark.parametrize("opts", FORMS, ids=IDS)
def test_position_coder_writes_encodeSingleQualByCol_streams(codec, opts):
    assert U.check_pos(codec, opts) == 13                # (the fourteenth vector asks for a value its buffer does not hold: no stream in a file)


@pytest.mark.parametrize("opts", FORMS, ids=IDS)
def test_k_overlap_finds_what_overlap_finds(codec, opts):
    assert U.check_overlap(codec, opts) == 8


@pytest.mark.parametrize("opts", FORMS, ids=IDS)
def test_name_parse_equals_FastqMeta_parse(codec, opts):
    assert U.check_parse(codec, opts) == 18
