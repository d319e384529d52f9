"""CPU: tests/_units.py (the reference's private-member vectors checked section by section at the boundary) under the SIMT interpreter."""
import pytest

import _engine as E
import _units as U


@pytest.fixture(scope="module")
def codec():
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.build_emu())
    yield c
    c.close()


FORMS = [{}, {"RFQ_GATHER": "old"}, {"RFQ_QUAL": "bytes"}, {"RFQ_QUAL": "bytes", "RFQ_CODER": "list"}]


@pytest.mark.parametrize("opts", FORMS, ids=lambda o: "+".join("%s=%s" % kv for kv in o.items()) or "default")
def test_unit_vectors_at_the_boundary(codec, opts):
    assert U.check_coords(codec, opts) == 6
    assert U.check_overlap(codec, opts) == 8
    assert U.check_parse(codec, opts) == 18
    assert U.check_pos(codec, opts) == 13                # (the fourteenth vector asks for a value its buffer does not hold: no stream in a file)
