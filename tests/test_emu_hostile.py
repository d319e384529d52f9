"""CPU: hostile .rfq images through the decoder under the SIMT interpreter (tests/_hostile.py) - a bounded subset in the suite; tools/hostile_asan.sh runs the
full tame set against an AddressSanitizer build of the same sources (log: profiles/r06_hostile_asan.txt).  The GPU twin is tests/test_gpu_hostile.py."""
import pytest

import _engine as E
import _hostile as H

COUNTS = dict(flip=12, header=6, fixed=10, lengths=6, quality=6, index=12)


@pytest.fixture(scope="module")
def codec():
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.build_emu())
    assert "simt-emulation" in c.version()
    yield c
    c.close()


@pytest.mark.parametrize("mode", [(), (("RFQ_MATERIALISE", "1"), ("RFQ_WALK", "exact"))], ids=["default", "materialise+exact_walk"])
def test_hostile_images_are_refused_or_decoded_and_leave_no_state(codec, mode):
    s = H.run(codec, modes=(mode,), counts=COUNTS, good_every=8, tame=True, time_bound_s=30.0)
    assert s["mutants"] >= 250 and s["good_checks"] >= 30 and s["errors"].get("FORMAT", 0) > 50 and s["decoded"] > 50, s
