"""Differential fuzz of the HIP path against the oracle on seeded random inputs (tests/_fuzz.py): a few seeds under the SIMT
interpreter on CPU, a few hundred through the product library on the GPU.  The oracle itself is checked against the reference
binary on the same seeds where oracle/_ref/repaq exists."""
import pytest

import _engine as E
import _fuzz as F
import _oracle as O


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_on_simt_emulation(seed):
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.build_emu())
    try:
        F.check(c, E.encode, seed)
    finally:
        c.close()


@pytest.mark.skipif(not O.have_ref(), reason="needs the compiled reference (oracle/_ref/repaq)")
@pytest.mark.parametrize("seed", range(0, 40))
def test_oracle_equals_reference_binary_on_fuzz_inputs(seed, tmp_path):
    fq1, fq2, paired, cb = F.case(seed)
    assert O.encode_file(fq1, fq2, paired, cb) == O.ref_encode(fq1, fq2, paired, cb // 1000, tmpdir=str(tmp_path))


@pytest.mark.parametrize("seed", range(6))
def test_block_fuzz_on_simt_emulation(seed):
    """inputs of one to three reader blocks with varied ends and block edges (tests/_fuzz.py::block_case)"""
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.build_emu())
    try:
        F.check_block(c, E.encode, seed)
    finally:
        c.close()


@pytest.mark.skipif(not O.have_ref(), reason="needs the compiled reference (oracle/_ref/repaq)")
@pytest.mark.parametrize("seed", range(0, 60))
def test_oracle_equals_reference_binary_on_block_fuzz_inputs(seed, tmp_path):
    fq1, fq2, paired, cb = F.block_case(seed)
    try:
        want = O.encode_file(fq1, fq2, paired, cb)
    except O.OracleError:
        pytest.skip("an input both sides refuse (reference UB zone)")
    assert want == O.ref_encode(fq1, fq2, paired, cb // 1000, tmpdir=str(tmp_path))


@pytest.mark.parametrize("gen,seed", [(g, s) for g in ("long_case", "qual_case") for s in range(4)])
def test_shape_fuzz_on_simt_emulation(gen, seed):
    """long reads (byte-wise tile paths, materialising decoder) / quality tables of 1 .. 93 values with late exceptions (tests/_fuzz.py)"""
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.build_emu())
    try:
        F.check_gen(c, E.encode, getattr(F, gen), seed)
    finally:
        c.close()


@pytest.mark.skipif(not O.have_ref(), reason="needs the compiled reference (oracle/_ref/repaq)")
@pytest.mark.parametrize("gen,seed", [(g, s) for g in ("long_case", "qual_case") for s in range(12)])
def test_oracle_equals_reference_binary_on_shape_fuzz_inputs(gen, seed, tmp_path):
    fq1, fq2, paired, cb = getattr(F, gen)(seed)
    try:
        want = O.encode_file(fq1, fq2, paired, cb)
    except O.OracleError:
        pytest.skip("an input both sides refuse (reference UB zone)")
    assert want == O.ref_encode(fq1, fq2, paired, cb // 1000, tmpdir=str(tmp_path))


@pytest.mark.parametrize("seed", range(10))
def test_overlap_fuzz_on_simt_emulation(seed):
    """low-complexity pairs, planted overlaps with point changes, N and odd characters: many candidates pass the search's filter (tests/_fuzz.py)"""
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.build_emu())
    try:
        assert F.check_gen(c, E.encode, F.overlap_case, seed) in ("ok", "error")
    finally:
        c.close()


@pytest.mark.skipif(not O.have_ref(), reason="needs the compiled reference (oracle/_ref/repaq)")
@pytest.mark.parametrize("seed", range(24))
def test_oracle_equals_reference_binary_on_overlap_fuzz_inputs(seed, tmp_path):
    fq1, fq2, paired, cb = F.overlap_case(seed)
    try:
        want = O.encode_file(fq1, fq2, paired, cb)
    except O.OracleError:
        pytest.skip("a file the reference refuses (a base outside A/C/G/T/N)")
    assert want == O.ref_encode(fq1, fq2, paired, cb // 1000, tmpdir=str(tmp_path))


@pytest.mark.gpu
def test_overlap_fuzz_on_gpu():
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.PRODUCT_LIB)
    try:
        outcomes = [F.check_gen(c, E.encode, F.overlap_case, seed) for seed in range(300)]
    finally:
        c.close()
    assert outcomes.count("ok") > 240 and outcomes.count("ok") + outcomes.count("error") == 300


@pytest.mark.gpu
def test_shape_fuzz_on_gpu():
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.PRODUCT_LIB)
    try:
        outcomes = [F.check_gen(c, E.encode, g, seed) for g in (F.long_case, F.qual_case) for seed in range(80)]
    finally:
        c.close()
    assert outcomes.count("ok") > 120


@pytest.mark.gpu
def test_block_fuzz_on_gpu():
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.PRODUCT_LIB)
    try:
        outcomes = [F.check_block(c, E.encode, seed) for seed in range(120)]
    finally:
        c.close()
    assert outcomes.count("ok") > 60


@pytest.mark.gpu
def test_fuzz_on_gpu():
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.PRODUCT_LIB)
    assert "gfx950" in c.version()
    try:
        outcomes = [F.check(c, E.encode, seed) for seed in range(300)]
    finally:
        c.close()
    assert outcomes.count("ok") > 250
