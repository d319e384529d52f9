"""CPU: kernel LOGIC of the decode path under the SIMT interpreter (see test_emu_encode.py for what that is)."""
import pytest

import _engine as E
import _oracle as O
from cases import CASES


@pytest.fixture(scope="module")
def codec():
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.build_emu())
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _options_back_to_default(codec):
    """rfq_set_option switches a test sets on the shared codec do not outlive it"""
    yield
    E.reset_options(codec)


def _oracle_rfq(case):
    try:
        return O.encode_file(case["fq1"], case.get("fq2", b""), case["paired"], case.get("k", 1000) * 1000)
    except O.OracleError:
        return None


DECODABLE = sorted(n for n in CASES if n != "se_name_over_255" and _oracle_rfq(CASES[n]) is not None)   # >255-byte names: reference UB


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
def test_case_decodes_like_oracle(codec, name):
    rfq = _oracle_rfq(CASES[name]); split = CASES[name]["paired"] != 0
    assert codec.decode_bytes(rfq, split_pe=split) == O.decode_file(rfq, split)
    if split:   # Repaq::decompress on a PE file: one interleaved stream (Q14)
        assert codec.decode_bytes(rfq, split_pe=False) == O.decode_file(rfq, False)


MULTI = [
    ("se150", O.NOVA_SE150, 600, 2, 20000, O.SE, {}),
    ("se150_manyN", O.NOVA_SE150, 600, 2, 20000, O.SE, dict(nppm=5000)),
    ("se_var", O.SE_VAR, 600, 3, 15000, O.SE, {}),
    ("pe150", O.NOVA_PE150, 300, 4, 20000, O.PE_TWO_FILES, {}),
    ("bgi_q40", O.BGI_PE100, 300, 5, 10000, O.PE_TWO_FILES, dict(n_quals=40)),
    ("se150_no_final_newline", O.NOVA_SE150, 500, 6, 7777, O.SE, dict(nonl=1)),
    ("pe150_r2_no_final_newline", O.NOVA_PE150, 300, 7, 9000, O.PE_TWO_FILES, dict(nonl=2, nppm=3000)),
]


@pytest.mark.parametrize("label,prof,reads,seed,cb,paired,kw", MULTI, ids=[m[0] for m in MULTI])
def test_multichunk_round_trip(codec, label, prof, reads, seed, cb, paired, kw):
    fq1, fq2 = O.gen(prof, reads, seed=seed, **kw)
    rfq = O.encode_file(fq1, fq2, paired, cb)
    d = codec.decode_bytes(rfq, split_pe=(paired != O.SE))
    assert d == ((fq1, fq2) if paired != O.SE else fq1)


@pytest.mark.parametrize("label,prof,reads,seed,cb,paired,kw", MULTI, ids=[m[0] for m in MULTI])
def test_multichunk_round_trip_tile_fitting_emitter(codec, label, prof, reads, seed, cb, paired, kw):
    """RFQ_MATERIALISE=1: the expanded path (qualities and bases expanded in HBM, k_dec_emit's tiles fitted read by read) - the path of a streaming
    caller's non-final slices and of files whose name pieces are stored per read."""
    codec.set_option("RFQ_MATERIALISE", "1")
    fq1, fq2 = O.gen(prof, reads, seed=seed, **kw)
    rfq = O.encode_file(fq1, fq2, paired, cb)
    d = codec.decode_bytes(rfq, split_pe=(paired != O.SE))
    assert d == ((fq1, fq2) if paired != O.SE else fq1)
    assert "emit_expanded" in dict(codec.timings())


def test_emitters_are_the_ones_expected(codec):
    """k_dec_emit3 (fixed tiles, no output tile) decodes files whose chunks share their name pieces - NovaSeq-style names do; a file with
    per-read name pieces goes to the expanded path (k_dec_emit)."""
    for prof, paired in ((O.NOVA_PE150, O.PE_TWO_FILES), (O.NOVA_SE150, O.SE)):
        fq1, fq2 = O.gen(prof, 300, seed=41)
        rfq = O.encode_file(fq1, fq2, paired, 20000)
        d = codec.decode_bytes(rfq, split_pe=(paired != O.SE))
        assert d == ((fq1, fq2) if paired != O.SE else fq1)
        assert "emit" in dict(codec.timings()), dict(codec.timings())
    fq1, _ = O.gen(O.SE_VAR, 300, seed=42)
    lines = fq1.split(b"\n"); lines[4 * 7] = lines[4 * 7] + b"x"; fq = b"\n".join(lines)      # one name that differs from the others in the part before the coordinates
    rfq = O.encode_file(fq, b"", O.SE, 20000)
    assert codec.decode_bytes(rfq) == fq


def _handmade(n, name_of, len_of, strand_of, seed, qual_of=None):
    import random
    rnd = random.Random(seed); out = []
    for i in range(n):
        ln = len_of(i); seq = "".join(rnd.choice("ACGT") if rnd.random() > 0.01 else "N" for _ in range(ln))
        qual = "".join("#" if c == "N" else (qual_of(i, j) if qual_of else rnd.choice("FFFFFF:,")) for j, c in enumerate(seq))
        out.append("@%s\n%s\n%s\n%s\n" % (name_of(i), seq, strand_of(i), qual))
    return "".join(out).encode()


@pytest.mark.parametrize("label,name_of,strand_of", [
    ("novaseq_names", lambda i: "A00123:45:HXXYYDSXX:1:1101:%d:%d 1:N:0:ACGTACGT" % (1000 + 7 * i, 2000 + i), lambda i: "+"),
    ("short_names", lambda i: "a:1:b:1:2:%d:%d x" % (1 + i % 120, 3 + i % 95), lambda i: "+"),         # name lines of 17 .. 21 bytes
    ("tiny_names", lambda i: "a:1:b:1:2:%d:%d" % (i % 9, i % 7), lambda i: "+"),                       # name lines of < 16 bytes
    ("strand_text", lambda i: "A00123:45:HXXYYDSXX:1:1101:%d:%d" % (1000 + i, 2000 + i), lambda i: "+strand"),
], ids=lambda v: v if isinstance(v, str) else None)
def test_fixed_tile_emitter_line_shapes(codec, label, name_of, strand_of):
    """k_dec_emit3 writes the name line as whole 16-byte groups merged from its three pieces, and lets "\\n+\\n" and the last '\\n' ride on the last
    stores of the base / quality lines: every read length 1 .. 70 (all residues mod 16, reads shorter than one group) under name lines shorter
    than, around and well above 16 bytes, and strand lines that carry text (written piece by piece)."""
    fq = _handmade(420, name_of, lambda i: 1 + (i * 11) % 70, strand_of, seed=len(label))
    for cb in (900, 100000):
        rfq = O.encode_file(fq, b"", O.SE, cb)
        assert O.decode_file(rfq) == fq
        assert codec.decode_bytes(rfq) == fq
        assert label in ("tiny_names", "strand_text") or "emit" in dict(codec.timings()), dict(codec.timings())   # (those two hold per-read pieces: the expanded path)
    fq2 = _handmade(420, name_of, lambda i: 1 + (i * 11) % 70, strand_of, seed=99)
    rfq = O.encode_file(fq, fq2, O.PE_TWO_FILES, 2000)
    assert codec.decode_bytes(rfq, split_pe=True) == (fq, fq2)
    assert codec.decode_bytes(rfq, split_pe=False) == O.decode_file(rfq, split_pe=False)          # (interleaved text: the mates in stored orientation)


def test_fixed_tile_emitter_with_per_read_name_pieces():
    """Names FastqMeta::parse does not take apart are stored whole, per read: k_dec_emit3 stages a tile's name pieces when the chunks' AVERAGE piece
    leaves room (the host sizes the tile by it); a tile whose reads carry much longer names than that raises DE_E3_RETRY and the range is emitted
    again by the expanded path (the retry is told to go there: it terminates whatever made the tile test fail) - as are the following ranges of the SAME image on
    that context; the next header (another file) starts afresh; long names throughout go to the expanded path at once."""
    from repaq_amd import RfqCodec
    codec = RfqCodec(device=0, library=E.build_emu())                            # (a context that has not given up on such files yet)
    short = _handmade(900, lambda i: "SRR0123456.%d %d length=150" % (i + 1, i + 1), lambda i: 150 - (i % 3), lambda i: "+", seed=3)
    rfq = O.encode_file(short, b"", O.SE, 50_000)
    assert codec.decode_bytes(rfq) == short
    assert "emit" in dict(codec.timings()), dict(codec.timings())
    mixed = _handmade(900, lambda i: ("SRR0123456.%d" % i) if not 400 <= i < 480 else ("L" * 110 + "%d" % i), lambda i: 100, lambda i: "+", seed=4)
    rfq = O.encode_file(mixed, b"", O.SE, 1_000_000)
    assert codec.decode_bytes(rfq) == mixed
    assert "emit_expanded" in dict(codec.timings()), dict(codec.timings())               # (the retry)
    assert codec.decode_bytes(O.encode_file(short, b"", O.SE, 50_000)) == short
    assert "emit" in dict(codec.timings()) and "emit_expanded" not in dict(codec.timings())   # (another file, another header: not held against it)
    codec.close()
    codec = RfqCodec(device=0, library=E.build_emu())
    longn = _handmade(300, lambda i: "N" * 240 + "%d" % i, lambda i: 100, lambda i: "+", seed=5)          # (64 of them do not fit the large tile either)
    assert codec.decode_bytes(O.encode_file(longn, b"", O.SE, 1_000_000)) == longn
    assert "emit_expanded" in dict(codec.timings()) and "emit" not in dict(codec.timings())
    mid = _handmade(300, lambda i: "N" * 120 + "%d" % i, lambda i: 100, lambda i: "+", seed=6)           # (the instantiation with the 13 KB name tile)
    assert codec.decode_bytes(O.encode_file(mid, b"", O.SE, 1_000_000)) == mid
    assert "emit" in dict(codec.timings()), dict(codec.timings())
    codec.close()


def test_position_lists_of_long_runs(codec):
    """Long runs of a minority quality value: a 256-byte step of its position stream codes thousands of list entries, more than the 1024 a wave
    of k_dec_pos_list fills in at a time (run tokens whose entries straddle the windows), next to streams of single positions."""
    nm = lambda i: "A00123:45:HXXYYDSXX:1:1101:%d:%d 1:N:0:ACGT" % (1000 + 3 * i, 2000 + i)
    fq = _handmade(900, nm, lambda i: 150, lambda i: "+", seed=5, qual_of=lambda i, j: "F" if j < 78 + (i % 5) else ("," if (i + j) % 41 else ":"))
    for cb in (30000, 1_000_000):
        rfq = O.encode_file(fq, b"", O.SE, cb)
        assert O.decode_file(rfq) == fq
        assert codec.decode_bytes(rfq) == fq
        assert "emit" in dict(codec.timings())


COMPAT = [("se_nonl", O.NOVA_SE150, 1500, 1, O.SE, dict(nonl=1)), ("pe_nonl_r2", O.NOVA_PE150, 1200, 4, O.PE_TWO_FILES, dict(nonl=2)),
          ("pe_nonl_r1", O.NOVA_PE150, 1200, 5, O.PE_TWO_FILES, dict(nonl=1)), ("pe_nonl_both", O.NOVA_PE150, 1200, 6, O.PE_TWO_FILES, dict(nonl=3)),
          ("pe_one_chunk", O.NOVA_PE150, 100, 7, O.PE_TWO_FILES, dict(nonl=3)), ("pe_with_newlines", O.NOVA_PE150, 700, 8, O.PE_TWO_FILES, {})]


@pytest.mark.parametrize("label,prof,reads,seed,paired,kw", COMPAT, ids=[c[0] for c in COMPAT])
def test_bug_compat_decode_loses_what_the_reference_loses(codec, label, prof, reads, seed, paired, kw):
    """rfq_decode_args.bug_compat: Repaq::decompressPE as it stands - the chunk behind a non-last NO_LINE_BREAK chunk is lost, and the
    flagged chunk's R2 text when it is the R1 bit (src/repaq.cpp:376-403; oracle: rfqo_decode_file_compat, pinned against the reference
    binary in test_oracle_golden.py); Repaq::decompress (one output) loses nothing (the SE case: three chunks, two of them flagged and not the last).
    One call, and a streaming caller's slices (a flagged chunk at the end of a non-final slice waits for what follows)."""
    fq1, fq2 = O.gen(prof, reads, seed=seed, **kw)
    rfq = O.encode_file(fq1, fq2, paired, 100_000)
    split = paired != O.SE
    want = O.decode_file(rfq, split_pe=split, bug_compat=True)
    keep = O.decode_file(rfq, split_pe=split)
    assert keep == ((fq1, fq2) if split else fq1)
    if label in ("pe_nonl_r2", "pe_nonl_r1", "pe_nonl_both"):
        assert want != keep                                                   # (these inputs do lose text in the reference)
    if not split:
        assert want == keep and len(O.chunk_table(rfq)) - 1 >= 3             # (one output: nothing is lost, flagged chunks or not)
    assert codec.decode_bytes(rfq, split_pe=split, bug_compat=True) == want
    assert codec.decode_bytes(rfq, split_pe=split) == keep
    for step in (9000, 70000):
        assert E.decode_in_slices(codec, rfq, split, step, bug_compat=True) == want, step
    if split:
        assert codec.decode_bytes(rfq, split_pe=False, bug_compat=True) == O.decode_file(rfq, split_pe=False, bug_compat=True)


def test_chunk_starts_without_an_index(codec):
    """A .rfq has no chunk index: the decoder guesses segment starts, walks the segments in parallel and verifies every extent (k_dec_gw_*);
    RFQ_WALK=exact is the one-wave serial walk it falls back to.  Small segments (RFQ_GW_SHIFT) so that a small image has several."""
    fq1, fq2 = O.gen(O.NOVA_PE150, 3000, seed=51)
    rfq = O.encode_file(fq1, fq2, O.PE_TWO_FILES, 9000)
    assert len(O.chunk_table(rfq)) - 1 > 80
    codec.set_option("RFQ_GW_SHIFT", "12")
    assert codec.decode_bytes(rfq, split_pe=True) == (fq1, fq2)
    codec.set_option("RFQ_WALK", "exact")
    assert codec.decode_bytes(rfq, split_pe=True) == (fq1, fq2)
    codec.set_option("RFQ_WALK", None)
    # chunks of very different sizes (a guessed start that is not one, segments without any start): whatever happens, the text is right
    big, _ = O.gen(O.SE_VAR, 4000, seed=52)
    for cb in (3000, 150000):
        assert codec.decode_bytes(O.encode_file(big, b"", O.SE, cb)) == big
    trunc = rfq[: len(rfq) - 1000]
    with pytest.raises(Exception):
        codec.decode_bytes(trunc, split_pe=True)


def test_encode_then_decode_on_device_round_trip(codec):
    fq1, fq2 = O.gen(O.NOVA_PE150, 250, seed=33)
    rfq = E.encode(codec, fq1, fq2, O.PE_TWO_FILES, 15000)
    assert codec.decode_bytes(rfq, split_pe=True) == (fq1, fq2)


def test_decode_rejects_other_algorithm_version(codec):
    from repaq_amd import RfqError
    rfq = bytearray(O.encode_file(CASES["d5_tiny_se"]["fq1"])); rfq[8] = 1
    with pytest.raises(RfqError) as e:
        codec.decode_bytes(bytes(rfq))
    assert e.value.code == -6 and "different version of repaq" in e.value.message
    with pytest.raises(RfqError):
        codec.decode_bytes(b"XYZ0.5.1" + bytes(rfq[8:]).replace(b"\x01", b"\x02", 1))


def test_decode_truncated_image_fails_cleanly(codec):
    from repaq_amd import RfqError
    rfq = O.encode_file(CASES["pe_overlap_sweep"]["fq1"], CASES["pe_overlap_sweep"]["fq2"], O.PE_TWO_FILES)
    with pytest.raises(RfqError):
        codec.decode_bytes(rfq[: len(rfq) - 7], split_pe=True)


@pytest.mark.parametrize("step", [700, 5000, 60000])
def test_decode_in_slices_equals_one_shot(codec, step):
    """rfq_decode_batch on successive byte ranges (a chunk cut by the range end is carried into the next call)."""
    fq1, fq2 = O.gen(O.NOVA_PE150, 400, seed=71, nonl=2)
    rfq = O.encode_file(fq1, fq2, O.PE_TWO_FILES, 20000)
    assert E.decode_in_slices(codec, rfq, True, step) == (fq1, fq2)
    se, _ = O.gen(O.SE_VAR, 500, seed=72, nonl=1)
    rfq = O.encode_file(se, b"", O.SE, 15000)
    assert E.decode_in_slices(codec, rfq, False, step) == se


@pytest.mark.parametrize("prof,reads,kw", [(O.NOVA_SE150, 7500, {}), (O.BGI_PE100, 6000, dict(n_quals=40)), (O.NOVA_SE150, 7500, dict(nppm=5000))],
                         ids=["se150", "bgi_q40", "se150_manyN"])
def test_full_size_chunk_position_streams_span_many_segments(codec, prof, reads, kw):
    """-k 1000 chunks: position streams of 10-100 KB are decoded in 1 KB / 2 KB segments by independent waves, which enter them in any of the token
    automaton's states - by the list passes (k_dec_pos_sum2 / link2 / list) and, RFQ_MATERIALISE=1, by the materialising path a streaming caller's
    non-final slices take (k_dec_pos_sum / link / emit)."""
    fq1, fq2 = O.gen(prof, reads, seed=9, **kw)
    rfq = O.encode_file(fq1, fq2, O.PE_TWO_FILES if fq2 else O.SE, 1_000_000)
    assert codec.decode_bytes(rfq, split_pe=bool(fq2)) == ((fq1, fq2) if fq2 else fq1)
    assert "emit" in dict(codec.timings()) or "emit_expanded" in dict(codec.timings())
    codec.set_option("RFQ_MATERIALISE", "1")
    assert codec.decode_bytes(rfq, split_pe=bool(fq2)) == ((fq1, fq2) if fq2 else fq1)
    assert "textlen" in dict(codec.timings()), dict(codec.timings())           # (a stage of its own only on that path)


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
    E.decode_with_chunk_index(codec)


@pytest.mark.parametrize("name", sorted(E.rle_goldens()))
def test_legacy_run_length_quality_images_decode_like_the_reference(codec, name):
    E.check_rle_decode(codec, name, E.rle_goldens()[name])
