"""CPU: kernel LOGIC of the encode path under the SIMT interpreter (tests/emu) — the same .hip sources compiled by g++,
every HIP thread a fiber, wave64 collectives as rendezvous.  This is test infrastructure for a GPU-less box; parity
proper is tests/test_gpu_*.py on the MI355X.  Sizes are kept small: the interpreter is ~1000x slower than the GPU."""
import json
import os

import pytest

import _engine as E
import _oracle as O
from cases import CASES

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES_J = json.load(open(os.path.join(G, "cases.json")))


@pytest.fixture(scope="module")
def codec():
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.build_emu())
    assert "simt-emulation" in c.version()
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _options_back_to_default(codec):
    """rfq_set_option switches a test sets on the shared codec do not outlive it"""
    yield
    E.reset_options(codec)


@pytest.mark.parametrize("name", sorted(CASES))
def test_case_matches_reference_golden(codec, name):
    E.check_case(codec, name, CASES[name], CASES_J[name])


MULTI = [
    ("se150", O.NOVA_SE150, 600, 2, 20000, O.SE, {}),
    ("se150_manyN", O.NOVA_SE150, 600, 2, 20000, O.SE, dict(nppm=5000)),
    ("se_var", O.SE_VAR, 600, 3, 15000, O.SE, {}),
    ("pe150", O.NOVA_PE150, 300, 4, 20000, O.PE_TWO_FILES, {}),
    ("pe150_interleaved_in", O.NOVA_PE150, 300, 4, 20000, O.PE_INTERLEAVED, dict(interleaved=True)),
    ("bgi_q40", O.BGI_PE100, 300, 5, 10000, O.PE_TWO_FILES, dict(n_quals=40)),
    ("se150_no_final_newline", O.NOVA_SE150, 500, 6, 7777, O.SE, dict(nonl=1)),
    ("pe150_r2_no_final_newline", O.NOVA_PE150, 300, 7, 9000, O.PE_TWO_FILES, dict(nonl=2, nppm=3000)),
    ("se150_single_chunk", O.NOVA_SE150, 700, 8, 1_000_000, O.SE, {}),
]


@pytest.mark.parametrize("label,prof,reads,seed,cb,paired,kw", MULTI, ids=[m[0] for m in MULTI])
def test_multichunk_matches_oracle(codec, label, prof, reads, seed, cb, paired, kw):
    fq1, fq2 = O.gen(prof, reads, seed=seed, **kw)
    assert E.encode(codec, fq1, fq2, paired, cb) == O.encode_file(fq1, fq2, paired, cb)


@pytest.mark.parametrize("label,prof,reads,seed,cb,paired,kw", MULTI[:6], ids=[m[0] for m in MULTI[:6]])
def test_multichunk_bytewise_gather_matches_oracle(codec, label, prof, reads, seed, cb, paired, kw):
    """RFQ_GATHER=old: the byte-wise k_gather + k_packbytes (the path of reads too long for a tile and of mates with odd bases)."""
    codec.set_option("RFQ_GATHER", "old")
    fq1, fq2 = O.gen(prof, reads, seed=seed, **kw)
    assert E.encode(codec, fq1, fq2, paired, cb) == O.encode_file(fq1, fq2, paired, cb)
    assert "gather_bytes" in dict(codec.timings())


@pytest.mark.parametrize("label,prof,reads,seed,cb,paired,kw", MULTI[3:8], ids=[m[0] for m in MULTI[3:8]])
def test_multichunk_two_pass_index_matches_oracle(codec, label, prof, reads, seed, cb, paired, kw):
    """RFQ_INDEX=2pass: newline bitmap -> scan -> line offsets (what the one-pass k_line_index falls back to when the text holds more lines than its table)."""
    codec.set_option("RFQ_INDEX", "2pass")
    fq1, fq2 = O.gen(prof, reads, seed=seed, **kw)
    assert E.encode(codec, fq1, fq2, paired, cb) == O.encode_file(fq1, fq2, paired, cb)
    assert "index_2pass" in dict(codec.timings())


@pytest.mark.parametrize("label,prof,reads,seed,cb,paired,kw", MULTI[:4], ids=[m[0] for m in MULTI[:4]])
def test_multichunk_index_over_several_workgroups_matches_oracle(codec, label, prof, reads, seed, cb, paired, kw):
    """RFQ_IDX_TILES=4: 64 KiB of text per workgroup of k_line_index, so that these small inputs span several and the look-back runs."""
    codec.set_option("RFQ_IDX_TILES", "4")
    fq1, fq2 = O.gen(prof, reads, seed=seed, **kw)
    assert len(fq1) > 3 * 65536 or paired != O.SE
    assert E.encode(codec, fq1, fq2, paired, cb) == O.encode_file(fq1, fq2, paired, cb)
    assert "index_2pass" not in dict(codec.timings())


def test_line_index_falls_back_when_lines_are_short():
    """The one-pass index sizes its table for one line per 16 bytes (+ 4096; a table an earlier call left behind is used whole: a fresh codec here);
    a text of two-byte lines overflows it and is indexed in two passes."""
    from repaq_amd import RfqCodec
    codec = RfqCodec(device=0, library=E.build_emu())
    fq1, _ = O.gen(O.NOVA_SE150, 50, seed=12)
    assert E.encode(codec, fq1, b"", O.SE, 20000) == O.encode_file(fq1, b"", O.SE, 20000)
    assert "index_2pass" not in dict(codec.timings())
    tiny = b"".join(b"@%d\n%s\n+\n%s\n" % (i % 10, b"ACGT"[i % 4:i % 4 + 1], b"F") for i in range(12000))
    assert E.encode(codec, tiny, b"", O.SE, 5000) == O.encode_file(tiny, b"", O.SE, 5000)
    assert "index_2pass" in dict(codec.timings())
    codec.close()


@pytest.mark.parametrize("gather", ["tile", "bytes"])
def test_strand_lines_of_equal_length_and_different_bytes(codec, gather):
    """Strand lines that differ in a byte but not in length - anywhere in the chunk, first or last read of a gather tile, one character or behind the
    sixteenth - clear STRAND_SAME exactly like the reference's pass 1 (src/rfqcodec.cpp:220-250); k_read_table compares every read's with its
    predecessor's.  (Leaving the bytes to k_gather2, which has the line staged anyway, was built and measured: 2.9 GB less traffic in k_read_table but
    only 0.08 ms, against 0.2 ms more in the VALU-bound k_gather2 - not kept.)"""
    if gather == "bytes":
        codec.set_option("RFQ_GATHER", "old")
    fq1, fq2 = O.gen(O.NOVA_PE150, 400, seed=77)
    def with_strands(fq, edits):
        lines = fq.split(b"\n")
        for k, st in edits.items():
            lines[4 * k + 2] = st
        return b"\n".join(lines)
    cases = [({}, {}),                                                            # all "+": the flag stays
             ({70: b"-"}, {}), ({}, {131: b"-"}), ({63: b"x"}, {64: b"y"}), ({0: b"-"}, {}),          # one character, different places (read 0 itself: every other read differs)
             ({k: b"+strand_line_with_text_%02d" % (k % 7) for k in range(400)}, {k: b"+strand_line_with_text_%02d" % (k % 7) for k in range(400)}),   # same length, bytes differ behind the 16th
             ({k: b"+same_text_everywhere" for k in range(400)}, {k: b"+same_text_everywhere" for k in range(400)})]
    for e1, e2 in cases:
        a, b = with_strands(fq1, e1), with_strands(fq2, e2)
        for cb in (20000, 1_000_000):
            assert E.encode(codec, a, b, O.PE_TWO_FILES, cb) == O.encode_file(a, b, O.PE_TWO_FILES, cb), (sorted(e1)[:3], sorted(e2)[:3], cb)
        assert E.encode(codec, a, b"", O.SE, 30000) == O.encode_file(a, b"", O.SE, 30000)
    assert ("gather_bytes" if gather == "bytes" else "gather") in dict(codec.timings())


@pytest.mark.parametrize("label,prof,reads,seed,cb,paired,kw", MULTI[:6], ids=[m[0] for m in MULTI[:6]])
def test_multichunk_quality_bytes_instead_of_masks_matches_oracle(codec, label, prof, reads, seed, cb, paired, kw):
    """RFQ_QUAL=bytes: k_gather2 writes the quality bytes and counts them (the path of files with more than four coded values) where the default is match masks."""
    codec.set_option("RFQ_QUAL", "bytes")
    fq1, fq2 = O.gen(prof, reads, seed=seed, **kw)
    assert E.encode(codec, fq1, fq2, paired, cb) == O.encode_file(fq1, fq2, paired, cb)
    assert "quality_masks" not in dict(codec.timings())


def test_match_masks_with_values_chunk_0_does_not_have(codec):
    """Match-mask mode (<= 4 coded quality values in the header, which comes from chunk 0): values that first appear in later chunks are exception records -
    their bytes go to qcat at their positions, their bits to the exception plane; runs of them, at word / tile / segment borders, in reversed mates."""
    import random
    rng = random.Random(5)
    for paired, prof in ((O.SE, O.NOVA_SE150), (O.PE_TWO_FILES, O.NOVA_PE150)):
        fq1, fq2 = O.gen(prof, 900, seed=41)
        def spoil(fq, first_read):
            lines = fq.split(b"\n")
            for k in range(first_read, len(lines) // 4):
                q = bytearray(lines[4 * k + 3])
                r = rng.random()
                if r < 0.3:
                    for _ in range(rng.randrange(1, 4)): q[rng.randrange(len(q))] = rng.choice(b"!5?A")
                elif r < 0.4:
                    a = rng.randrange(len(q)); b = min(len(q), a + rng.randrange(1, 80)); q[a:b] = bytes([rng.choice(b"5?")]) * (b - a)
                elif r < 0.45:
                    q[:] = bytes([rng.choice(b"!A")]) * len(q)
                lines[4 * k + 3] = bytes(q)
            return b"\n".join(lines)
        a = spoil(fq1, 300); b = spoil(fq2, 300) if fq2 else b""       # (behind chunk 0 for both chunk sizes: the header keeps its three values)
        for cb in (20000, 33000):
            want = O.encode_file(a, b, paired, cb)
            assert E.encode(codec, a, b, paired, cb) == want
            assert "quality_masks" in dict(codec.timings())
            with codec.option("RFQ_QUAL", "bytes"):
                assert E.encode(codec, a, b, paired, cb) == want


@pytest.mark.parametrize("coder", ["list", "mask"])
@pytest.mark.parametrize("label,prof,reads,seed,cb,paired,kw", MULTI[:7], ids=[m[0] for m in MULTI[:7]])
def test_multichunk_both_position_coders_match_oracle(codec, coder, label, prof, reads, seed, cb, paired, kw):
    """RFQ_CODER=list / mask on the quality bytes (RFQ_QUAL=bytes: also for files that would get match masks): k_pos_coder_list - all value streams of a segment
    by one wave, from a per-step list of the coded positions - and k_pos_coder - a wave per four streams - write the same bytes."""
    codec.set_option("RFQ_QUAL", "bytes"); codec.set_option("RFQ_CODER", coder)
    fq1, fq2 = O.gen(prof, reads, seed=seed, **kw)
    assert E.encode(codec, fq1, fq2, paired, cb) == O.encode_file(fq1, fq2, paired, cb)


def test_list_coder_on_runs_that_cross_lanes_steps_and_segments(codec):
    """The list coder takes streak starts, distances and run lengths from the byte sequence: runs of one value that cross 64-position lanes, 4096-position steps
    and 32768-position segments, start at position 0 / 1 of the chunk, are 32 / 33 / 34 / 65 long, in files with many quality values."""
    import random
    rng = random.Random(11)
    vals = bytes(range(40, 40 + 24))
    def quals(n):
        out = bytearray(vals * 8)                                            # (chunk 0 sees every value: the header's table knows them all - no exception records)
        while len(out) < n:
            r = rng.random()
            run = rng.choice((1, 1, 1, 2, 3, 31, 32, 33, 34, 35, 63, 64, 65, 66, 130, 700) + ((4096, 5000, 33000) if len(out) > 110000 else ())) if r < 0.25 else 1
            out += bytes([rng.choice(vals)]) * run
        return bytes(out[:n])
    for L, nreads, cb in ((150, 1600, 40000), (100, 2500, 70000), (251, 1000, 100000)):
        q = quals(L * nreads)
        recs = []
        for i in range(nreads):
            recs.append(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGT") for _ in range(L)), q[i * L:(i + 1) * L]))
        fq = b"".join(recs)
        want = O.encode_file(fq, b"", O.SE, cb)
        for coder in ("list", "mask"):
            with codec.option("RFQ_CODER", coder):
                assert E.encode(codec, fq, b"", O.SE, cb) == want, (L, coder)
    # a streak from position 0 of the chunk (the `cur > 1` rule) in a many-valued file
    head = bytes([vals[3]]) * 70 + quals(150 * 300 - 70)[:150 * 300 - 70]
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"A" * 150, head[i * 150:(i + 1) * 150]) for i in range(300))
    for coder in ("list", "mask"):
        with codec.option("RFQ_CODER", coder):
            assert E.encode(codec, fq, b"", O.SE, 30000) == O.encode_file(fq, b"", O.SE, 30000), coder


def test_gather_paths_are_the_ones_expected(codec):
    """The tile gather (k_gather2 + k_seqpack) is what runs by default - also when a mate of an interleaved chunk holds bytes outside A/C/G/T/N
    (Read::changeToReverseComplement turns them into N, lower case into the upper-case complement) - and reads too long for a tile of two
    take the byte-wise gather."""
    fq1, fq2 = O.gen(O.NOVA_PE150, 200, seed=31)
    assert E.encode(codec, fq1, fq2, O.PE_TWO_FILES, 20000) == O.encode_file(fq1, fq2, O.PE_TWO_FILES, 20000)
    assert "gather" in dict(codec.timings()) and "quality_masks" in dict(codec.timings())     # (a NovaSeq-binned file: three coded quality values)
    lines = fq2.split(b"\n")
    for k, ch in ((150, b"r"), (151, b"a"), (152, b"n"), (170, b"."), (171, b"g")):     # mates of the second chunk or later (chunk 0 must be clean: the header is made from it)
        lines[4 * k + 1] = lines[4 * k + 1][:30] + ch + lines[4 * k + 1][31:]
    odd = b"\n".join(lines)
    l1 = fq1.split(b"\n"); l1[4 * 160 + 1] = l1[4 * 160 + 1][:7] + b"x" + l1[4 * 160 + 1][8:]; odd1 = b"\n".join(l1)
    assert E.encode(codec, odd1, odd, O.PE_TWO_FILES, 20000) == O.encode_file(odd1, odd, O.PE_TWO_FILES, 20000)
    assert "gather" in dict(codec.timings())
    long1, _ = O.gen(O.SE_VAR, 40, seed=5)
    big = b"@r\n" + b"ACGT" * 4000 + b"\n+\n" + b"F" * 16000 + b"\n"
    fq = long1 + big + long1
    assert E.encode(codec, fq, b"", O.SE, 20000) == O.encode_file(fq, b"", O.SE, 20000)
    assert "gather_bytes" in dict(codec.timings())


def test_batched_encode_with_carry_over_equals_one_shot(codec):
    """Repaq::compress reads a stream; the host driver feeds it in batches: non-final batches stop at the last full chunk and
    report consumed bytes, the remainder is carried into the next batch.  Concatenation must equal the one-shot image."""
    fq1, _ = O.gen(O.NOVA_SE150, 900, seed=21)
    cb = 12000
    want = O.encode_file(fq1, b"", O.SE, cb)
    codec.clearHeader()
    out = b""; pos = 0; step = 100_000; first = True
    while pos < len(fq1):
        end = min(len(fq1), pos + step); final = end == len(fq1)
        # cut the batch at a record boundary: keep whole 4-line records only (host driver's job)
        buf = fq1[pos:end]
        if not final:
            nl = [i for i, b in enumerate(buf) if b == 10]
            keep = nl[(len(nl) // 4) * 4 - 1] + 1
            buf = buf[:keep]
        d = codec.dev_put(buf)
        r = codec.encode(d, len(buf), None, 0, O.SE, cb, final=final, emit_header=first, file_off1=pos)
        out += codec.dev_get(r.d_rfq, r.rfq_len) if r.rfq_len else b""
        codec.dev_free(d)
        assert r.consumed1 > 0 or final
        pos += r.consumed1 if not final else len(buf)
        first = False
    assert out == want


def _segment_boundary_fastq(n_reads=1800, rl=150):
    """One ~270 kbase chunk (3 position-coder segments of 131072): streaks of a normal quality value that straddle the segment
    boundaries, a value that only occurs far before a boundary (long backward carry scan) and one that never occurs after read 3."""
    import random
    rng = random.Random(4242)
    recs = []
    for i in range(n_reads):
        q = ["F"] * rl
        pos0 = i * rl
        for k in range(rl):
            p = pos0 + k
            if 131072 - 200 <= p < 131072 + 300 or 262144 - 40 <= p < 262144 + 33:
                q[k] = ","                      # streaks across both segment boundaries
            elif rng.random() < 0.03:
                q[k] = ":"
        if i == 2:
            q[5] = "#"                          # '#' exists only here: every later segment scans back to the chunk start
        if i in (700, 701):
            q[10] = "5"
        seq = [rng.choice("ACGT") for _ in range(rl)]
        if i == 2:
            seq[5] = "N"
        recs.append("@A00250:26:H3YTWDSXX:1:1101:%d:%d 1:N:0:ACGT\n%s\n+\n%s\n" % (1000 + i, 2000 + i // 7, "".join(seq), "".join(q)))
    return "".join(recs).encode()


def test_position_coder_segments(codec):
    fq = _segment_boundary_fastq()
    assert E.encode(codec, fq, b"", O.SE, 1_000_000) == O.encode_file(fq, b"", O.SE, 1_000_000)
    assert E.encode(codec, fq, b"", O.SE, 140_000) == O.encode_file(fq, b"", O.SE, 140_000)


@pytest.mark.parametrize("label", ["se_nonl", "pe_nonl2", "se_crlf", "interleaved"])
def test_scan_then_chunk_parallel_encode_equals_one_shot(label):
    """rfq_scan_batch plans the chunk ends; separate contexts encode chunk ranges with flush_all; the concatenation is the one-shot image."""
    from repaq_amd import RfqCodec, PE_TWO_FILES, PE_INTERLEAVED, SE
    mk = lambda: RfqCodec(device=0, library=E.EMU_LIB)
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
    E.overlap_search_paths(codec)


def test_overlap_search_paths_short_rows(codec):
    """the same adversarial pairs with no read above 160 bases: the launch takes the 160-base row geometry (k_overlap<true, 160>; lengths up to the row's last base)"""
    E.overlap_search_paths(codec, lengths=(150, 150, 151, 100, 145, 40, 13, 12, 11), seed=77)


def test_sliced_calls_equal_one_shot():
    """Texts of >= 4 GiB per stream are encoded slice by slice, images of > ~1.6 G bases decoded range by range (32-bit offsets inside one
    pass).  RFQ_SLICE_BYTES / RFQ_SLICE_BASES shrink the slices so that the same code runs on small inputs (subprocess: read once)."""
    import subprocess
    import sys
    env = dict(os.environ, RFQ_SLICE_BYTES="150000", RFQ_SLICE_BASES="60000", PYTHONPATH=os.pathsep.join([E.ROOT, os.path.join(E.ROOT, "tests"), os.path.join(E.ROOT, "tests", "golden")]))
    r = subprocess.run([sys.executable, os.path.join(E.ROOT, "tests", "_slice_probe.py"), E.build_emu()], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "SLICES_OK 7" in r.stdout, r.stdout + r.stderr


import _shapes as SH

_SHAPES = SH.cases(260, 9000)


@pytest.mark.parametrize("label,fq1,fq2,paired,cb", _SHAPES, ids=[c[0] for c in _SHAPES])
def test_uniform_and_almost_uniform_read_lengths(codec, label, fq1, fq2, paired, cb):
    """closed-form prefixes / cuts where every read has one length, the scans everywhere else - and nothing in between (tests/_shapes.py)"""
    assert E.encode(codec, fq1, fq2, paired, cb) == O.encode_file(fq1, fq2, paired, cb)


def test_arenas_sized_in_advance_grow_and_the_batch_repeats():
    """The tile path sizes the stream arenas and its own image buffer before their sizes exist (no read-back between gather and coders) - from what the context holds,
    or from the header's shape (few coded quality values: a few percent of the positions are coded).  A fresh context meets a file whose four quality values are equally
    frequent - three positions in four are coded, far beyond bases / 8: DE_SCRATCH_SMALL -> room is made, the batch repeated (marker `retry_room`), the image is the
    oracle's; the next batch of the same context has room and does not repeat.  The forty-value shape is sized right at once (its header says so)."""
    import random
    from repaq_amd import RfqCodec
    rng = random.Random(5)
    recs = []
    for i in range(2400):
        seq = bytes(rng.choice(b"ACGT") for _ in range(150)); q = bytes(rng.choice(b"F:,5") for _ in range(150))
        recs.append(b"@M:1:FC:1:%d:%d:%d 1:N:0:AC\n" % (1101 + i // 500, rng.randrange(1000, 30000), rng.randrange(1000, 60000)) + seq + b"\n+\n" + q + b"\n")
    fq = b"".join(recs)
    c = RfqCodec(device=0, library=E.build_emu())
    try:
        want = O.encode_file(fq, b"", O.SE, 100000)
        assert E.encode(c, fq, b"", O.SE, 100000) == want
        assert "retry_room" in dict(c.timings()) and "quality_masks" in dict(c.timings())
        assert E.encode(c, fq, b"", O.SE, 100000) == want
        assert "retry_room" not in dict(c.timings())
        se, _ = O.gen(O.SE_VAR, 600, seed=3)                        # reads of several lengths on a context that expected one: the scans run after all
        assert E.encode(c, se, b"", O.SE, 15000) == O.encode_file(se, b"", O.SE, 15000)
    finally:
        c.close()
    c = RfqCodec(device=0, library=E.build_emu())
    try:
        fq1, fq2 = O.gen(O.BGI_PE100, 300, seed=5, n_quals=40)
        assert E.encode(c, fq1, fq2, O.PE_TWO_FILES, 10000) == O.encode_file(fq1, fq2, O.PE_TWO_FILES, 10000) and "retry_room" not in dict(c.timings())
    finally:
        c.close()


def test_no_read_back_behind_the_index_from_the_second_batch_on():
    """A context that knows the records per byte of an earlier batch sizes its per-read tables from that and leaves the index's totals on the device (marker
    `lazy_index`: one round trip less); a batch that holds more units than guessed - much shorter records - is encoded again with the read-back, and the
    guess follows the new shape.  Every image is the oracle's."""
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.build_emu())
    try:
        fq1, fq2 = O.gen(O.NOVA_PE150, 400, seed=4, nonl=2)
        want = O.encode_file(fq1, fq2, O.PE_TWO_FILES, 20000)
        assert E.encode(c, fq1, fq2, O.PE_TWO_FILES, 20000) == want and "lazy_index" not in dict(c.timings())      # the first batch of a context reads back
        assert E.encode(c, fq1, fq2, O.PE_TWO_FILES, 20000) == want and "lazy_index" in dict(c.timings())
        import _shapes as SH
        short, _ = SH.fastq([20] * 3000, None, 31)                                    # records of ~100 bytes instead of ~360: three times the units per byte
        assert E.encode(c, short, b"", O.SE, 9000) == O.encode_file(short, b"", O.SE, 9000)
        assert "lazy_index" not in dict(c.timings())                                  # (the repeat took the read-back)
        assert E.encode(c, short, b"", O.SE, 9000) == O.encode_file(short, b"", O.SE, 9000) and "lazy_index" in dict(c.timings())
        se, _ = O.gen(O.NOVA_SE150, 500, seed=9)                                     # fewer units than guessed: fine
        assert E.encode(c, se, b"", O.SE, 30000) == O.encode_file(se, b"", O.SE, 30000) and "lazy_index" in dict(c.timings())
        il, _ = O.gen(O.NOVA_PE150, 300, seed=4, interleaved=True)
        assert E.encode(c, il, b"", O.PE_INTERLEAVED, 20000) == O.encode_file(il, b"", O.PE_INTERLEAVED, 20000)
        crlf = se.replace(b"\n", b"\r\n")                                             # '\r' is only seen behind the partition now: the normalising path all the same
        assert E.encode(c, crlf, b"", O.SE, 30000) == O.encode_file(crlf, b"", O.SE, 30000)
    finally:
        c.close()
