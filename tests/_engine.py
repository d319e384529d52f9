"""Shared helpers for parity tests: run a case through a RfqCodec (product library on the GPU box, or the SIMT-emulation
TEST build on CPU) and compare with the oracle / the reference's golden vectors."""
import hashlib
import os
import subprocess

import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "librfq_emu.so")
PRODUCT_LIB = os.path.join(ROOT, "repaq_amd", "lib", "librfq_hip.so")


def reset_options(codec):
    """every switch the library knows (rfq_option_name) back to its default: what a test that sets switches on a shared codec does afterwards"""
    for name in codec.option_names():
        codec.set_option(name, None)


def build_emu():
    # (pytest-xdist workers call this at the same time: one make at a time, the others find everything up to date)
    import fcntl
    with open(os.path.join(EMU_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.check_call(["make", "-s", "-j4", "-C", EMU_DIR, "all"])       # (all: the library AND the driver linked against it)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return EMU_LIB


def nolb_args(fq1, fq2, paired):
    from repaq_amd import nolb_threshold
    t1 = nolb_threshold(len(fq1), fq1.endswith(b"\n"))
    t2 = nolb_threshold(len(fq2), fq2.endswith(b"\n")) if paired == O.PE_TWO_FILES else t1
    return dict(nolb_from1=t1, nolb_from2=t2)


def encode(codec, fq1, fq2=b"", paired=O.SE, chunk_bases=1_000_000):
    codec.clearHeader()
    return codec.encode_bytes(fq1, fq2, paired, chunk_bases, **nolb_args(fq1, fq2, paired))


# text quirks of src/fastqreader.cpp ('\r', blank lines) — handled by the normalising path since SURVEY.md §8(f) #1 was built
TEXT_QUIRK_CASES = {"se_crlf", "se_cr_only", "se_crlf_no_final", "se_blank_line_after_record", "se_two_blank_lines_truncate"}


def check_case(codec, name, case, golden):
    """Encode one tests/golden/cases.py case and compare with the reference's golden (.rfq bytes or error text)."""
    from repaq_amd import RfqError
    fq1, fq2, paired = case["fq1"], case.get("fq2", b""), case["paired"]
    cb = case.get("k", 1000) * 1000
    if "error" in golden:
        try:
            encode(codec, fq1, fq2, paired, cb)
        except RfqError as e:
            assert e.message.strip() == golden["error"], (e.message, golden["error"])
            return "error-parity"
        raise AssertionError("expected the reference's error for %s" % name)
    got = encode(codec, fq1, fq2, paired, cb)
    assert len(got) == golden["rfq_len"], (len(got), golden["rfq_len"])
    assert hashlib.md5(got).hexdigest() == golden["rfq_md5"]
    if "rfq_hex" in golden:
        assert got.hex() == golden["rfq_hex"]
    return got


def decode_in_slices(codec, rfq: bytes, split_pe: bool, step: int, **kw):
    """Feed an image `step` bytes at a time (has_header on the first call, final on the last, unconsumed tail carried over): the
    streaming contract of rfq_decode_batch that the C++ driver relies on."""
    out1, out2 = bytearray(), bytearray()
    pos, end, first = 0, min(step, len(rfq)), True
    while True:
        final = end == len(rfq)
        buf = rfq[pos:end]
        d = codec.dev_put(buf)
        try:
            r = codec.decode(d, len(buf), has_header=first, split_pe=split_pe, final=final, **kw)
            if r.n1:
                out1 += codec.dev_get(r.d_fq1, r.n1)
            if split_pe and r.n2:
                out2 += codec.dev_get(r.d_fq2, r.n2)
            consumed = r.consumed
        finally:
            codec.dev_free(d)
        first = False
        if final:
            break
        pos += consumed
        end = min(len(rfq), max(end, pos) + step)
    return (bytes(out1), bytes(out2)) if split_pe else bytes(out1)


def scan_and_encode_in_ranges(make_codec, fq1, fq2, paired, chunk_bases, parts):
    """The multi-GPU host queue in miniature: one context plans the chunk ends (rfq_scan_batch), `parts` other contexts each encode a
    contiguous range of chunks (flush_all, header of the first range set on the others); returns the concatenated image."""
    from repaq_amd import PE_TWO_FILES, nolb_threshold
    two = paired == PE_TWO_FILES
    scanner = make_codec()
    d1 = scanner.dev_put(fq1); d2 = scanner.dev_put(fq2) if two else None
    r, e1, e2 = scanner.scan(d1, len(fq1), d2, len(fq2) if two else 0, paired, chunk_bases, final=True)
    scanner.dev_free(d1)
    if d2:
        scanner.dev_free(d2)
    scanner.close()
    nc = r.n_chunks
    th1 = nolb_threshold(len(fq1), fq1.endswith(b"\n")); th2 = nolb_threshold(len(fq2), fq2.endswith(b"\n")) if two else th1
    out, hdr, base = b"", None, max(1, nc // parts)
    for w in range(parts):
        c0, c1 = w * base, (nc if w == parts - 1 else min(nc, (w + 1) * base))
        if c1 <= c0:
            continue
        last = c1 == nc
        s1, t1 = (e1[c0 - 1] if c0 else 0), (len(fq1) if last else e1[c1 - 1])
        s2, t2 = ((e2[c0 - 1] if c0 else 0), (len(fq2) if last else e2[c1 - 1])) if two else (0, 0)
        wk = make_codec()
        if hdr is not None:
            wk.setHeader(hdr)
        p1, p2 = fq1[s1:t1], (fq2[s2:t2] if two else b"")
        a1 = wk.dev_put(p1); a2 = wk.dev_put(p2) if two else None
        rr = wk.encode(a1, len(p1), a2, len(p2) if two else 0, paired, chunk_bases, final=last, emit_header=(hdr is None), file_off1=s1, file_off2=s2,
                       nolb_from1=th1, nolb_from2=th2, flush_all=not last)
        out += wk.dev_get(rr.d_rfq, rr.rfq_len) if rr.rfq_len else b""
        if hdr is None:
            hdr = wk.header()
        wk.dev_free(a1)
        if a2:
            wk.dev_free(a2)
        wk.close()
    return out, nc


def overlap_search_paths(codec, lengths=(150, 150, 151, 100, 256, 257, 300, 40, 13, 12, 11), seed=99):
    """k_overlap's three paths against the oracle: packed 2-bit rows (reads <= 256 bases; N inside and beside the overlap, lower-case
    bases in R2, homopolymers that pass the 12-base filter at many candidates), the byte-wise search for reads > 256 bases, and the
    byte-wise search for an R1 that holds a character outside A/C/G/T/N in a later chunk (the header only vets chunk 0).
    `lengths`: the read lengths drawn from - with none above 151 (+ 9 for the mate) the launch takes the 160-base row geometry (k_overlap<true, 160>)."""
    import random
    rng = random.Random(seed)
    comp = {65: 84, 84: 65, 67: 71, 71: 67, 78: 78}

    def rc(s):
        return bytes(comp.get(b, 78) for b in reversed(s))

    r1, r2 = [], []
    for i in range(900):
        ln = rng.choice(lengths)
        kind = rng.random()
        seq = bytearray(rng.choice(b"ACGT") for _ in range(ln))
        if kind < 0.15:
            seq = bytearray(rng.choice(b"AC") * 1 for _ in range(ln)) if rng.random() < 0.5 else bytearray(b"A" * ln)   # low complexity: many filter hits
        if rng.random() < 0.3:
            for _ in range(rng.randint(1, 6)):
                seq[rng.randrange(ln)] = 78
        ln2 = rng.choice([ln, ln, max(1, ln - 7), ln + 9])
        mode = rng.random()
        if mode < 0.45 and ln >= 12:
            ov = rng.randint(12, ln); frag = bytes(seq[ln - ov:]) + bytes(rng.choice(b"ACGT") for _ in range(max(0, ln2 - ov))); s2 = rc(frag[:ln2])
        elif mode < 0.7 and ln >= 12:
            ov = rng.randint(12, ln); frag = bytes(rng.choice(b"ACGT") for _ in range(max(0, ln2 - ov))) + bytes(seq[:ov]); s2 = rc(frag[-ln2:])
        else:
            s2 = bytes(rng.choice(b"ACGT") for _ in range(ln2))
        s2 = bytearray(s2)
        if i > 500 and rng.random() < 0.1 and len(s2):
            k = rng.randrange(len(s2)); s2[k] = ord(chr(s2[k]).lower())          # complement rule takes either case
        if i > 500 and rng.random() < 0.1 and len(s2):
            s2[rng.randrange(len(s2))] = ord("X")                                # -> N in RC(R2)
        if i > 500 and rng.random() < 0.1:
            seq[rng.randrange(ln)] = rng.choice(b"acgtnX.")                       # R1 as it stands: equals nothing in RC(R2)
        name = b"@A00250:26:H3YTWDSXX:1:1101:%d:%d" % (1000 + i, 2000 + i // 7)
        r1.append(name + b" 1:N:0:ACGT\n" + bytes(seq) + b"\n+\n" + b"F" * ln + b"\n")
        r2.append(name + b" 2:N:0:ACGT\n" + bytes(s2) + b"\n+\n" + b"F" * len(s2) + b"\n")
    fq1, fq2 = b"".join(r1), b"".join(r2)
    want = O.encode_file(fq1, fq2, O.PE_TWO_FILES, 30000)
    assert encode(codec, fq1, fq2, O.PE_TWO_FILES, 30000) == want


def decode_with_chunk_index(codec):
    """rfq_decode_args.h_chunk_off: a caller that has the chunk offsets skips the chunk walk; every extent is still verified on the
    device, and an index that does not verify (shifted, truncated, out of range) is ignored - the chain is walked as without one."""
    fq1, fq2 = O.gen(O.NOVA_PE150, 400, seed=35)
    rfq = O.encode_file(fq1, fq2, O.PE_TWO_FILES, 15000)
    offs = O.chunk_table(rfq)
    assert len(offs) > 5

    def run(table):
        d = codec.dev_put(rfq)
        try:
            r = codec.decode(d, len(rfq), split_pe=True, chunk_off=table)
            return r.n_chunks, codec.dev_get(r.d_fq1, r.n1), codec.dev_get(r.d_fq2, r.n2)
        finally:
            codec.dev_free(d)

    assert run(offs) == (len(offs) - 1, fq1, fq2)
    assert run([o + (1 if 0 < i < len(offs) - 1 else 0) for i, o in enumerate(offs)]) == (len(offs) - 1, fq1, fq2)    # shifted entries
    assert run(offs[:-2]) == (len(offs) - 1, fq1, fq2)                                                             # covers only a prefix
    assert run(offs[:-1] + [len(rfq) + 100]) == (len(offs) - 1, fq1, fq2)                                           # past the image
    assert run([offs[0], offs[0] + 5]) == (len(offs) - 1, fq1, fq2)                                                 # too short to be a chunk


def rle_goldens():
    """tests/golden/rle.json: legacy run-length-coded images made by the reference's own encodeChunk (oracle/ref_harness.cpp rle_image) and the
    md5 of what the reference binary decoded from them."""
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "rle.json")))


def check_rle_decode(codec, name, g):
    """decodeQualByRunLenCoding on the device: byte-equal to the oracle and md5-equal to the reference's own decode."""
    img = bytes.fromhex(g["rfq_hex"]); split = bool(g["paired"])
    got = codec.decode_bytes(img, split_pe=split)
    assert got == O.decode_file(img, split), name
    assert [hashlib.md5(x).hexdigest() for x in (got if split else (got,))] == g["decode_md5"], name
    if split:
        assert codec.decode_bytes(img, split_pe=False) == O.decode_file(img, False), name


# ---------------------------------------------------------------- every alternative formulation, forced
# rfq_set_option switches select another formulation of the same bit-identical result.  On the GPU the fallbacks otherwise run only when the data happens to
# need them, and the SIMT interpreter cannot see a missing stream dependency, a wrong entry state or a DPP problem: tests/test_gpu_formulations.py forces
# every one of them on the hardware against the reference's goldens.  marker: a stage name rfq_last_timings must (with "!": must not) show when the switch took effect.
ENC_FORMS = {
    "default": ({}, None),
    "gather_old": ({"RFQ_GATHER": "old"}, "gather_bytes"),                                   # k_gather + k_chunk_flags_a + k_packbytes, k_read_table on every read, k_overlap<false>
    "qual_bytes": ({"RFQ_QUAL": "bytes"}, "!quality_masks"),                                 # k_gather2<false>: quality bytes + per-byte counters
    "coder_list": ({"RFQ_QUAL": "bytes", "RFQ_CODER": "list"}, "!quality_masks"),            # k_pos_coder_list on any file
    "coder_mask": ({"RFQ_QUAL": "bytes", "RFQ_CODER": "mask"}, "!quality_masks"),            # k_pos_coder on byte streams, also with forty values
    "index_2pass": ({"RFQ_INDEX": "2pass"}, "index_2pass"),                                  # k_nl_bitmap -> scan -> k_line_offsets
    "idx_tiles4": ({"RFQ_IDX_TILES": "4"}, "!index_2pass"),                                  # k_line_index<4>: four times the workgroups, longer look-backs
    "one_stream": ({"RFQ_STREAMS": "1"}, None),                                              # both chains on one stream
    "slices": ({"RFQ_SLICE_BYTES": None}, None),                                             # (value per input: about three chunks per slice)
    "old_2pass_one_stream": ({"RFQ_GATHER": "old", "RFQ_INDEX": "2pass", "RFQ_STREAMS": "1"}, "gather_bytes"),
}
DEC_FORMS = {
    "default": ({}, None),
    "walk_exact": ({"RFQ_WALK": "exact"}, "emit_expanded"),                                  # k_dec_walk; it does not look into the payloads: the expanded path behind it
    "materialise": ({"RFQ_MATERIALISE": "1"}, "emit_expanded"),                              # k_dec_fill / unpack / pos_sum / pos_link / pos_emit / except + k_dec_emit
    "one_stream": ({"RFQ_STREAMS": "1"}, None),
    "pos_seg_2048": ({"RFQ_POS_SEG": "2048"}, None),                                         # the list chain's larger segments (default only for streams of 32 KB and more)
    "pos_seg_1024": ({"RFQ_POS_SEG": "1024"}, None),
    "no_spec": ({"RFQ_SPEC": "0"}, None),                                                    # the emitter behind the host's look at the status, not ahead of it
    "slice_bases": ({"RFQ_SLICE_BASES": None}, None),                                        # ranges of two or three chunks
    "gw_small": ({"RFQ_GW_SHIFT": "12"}, None),                                              # many guess-and-verify segments
    "exact_materialise_slices": ({"RFQ_WALK": "exact", "RFQ_MATERIALISE": "1", "RFQ_SLICE_BASES": None}, None),
}


class _Options:
    """the switches of a formulation set on a codec; what was there before is put back afterwards"""
    def __init__(self, codec, opts):
        self.codec, self.opts, self.before = codec, opts, {}

    def __enter__(self):
        for k, v in self.opts.items():
            self.before[k] = self.codec.get_option(k)
            self.codec.set_option(k, v)
        return self.codec

    def __exit__(self, *exc):
        for k, v in self.before.items():
            self.codec.set_option(k, v or None)


def _marker_ok(codec, marker):
    if marker:
        names = dict(codec.timings())
        assert (marker[1:] not in names) if marker.startswith("!") else (marker in names), (marker, sorted(names))


def check_encode_formulation(codec, form, fq1, fq2, paired, chunk_bases, want_md5=None, want_len=None, want=None):
    """one input through rfq_encode_batch with the switches of ENC_FORMS[form]: the image equals the golden (md5 + size) or the expected bytes"""
    opts, marker = ENC_FORMS[form]
    opts = dict(opts)
    if "RFQ_SLICE_BYTES" in opts:
        opts["RFQ_SLICE_BYTES"] = str(max(4096, int(3 * 2.6 * chunk_bases)))      # about three chunks of text per slice (2.6 bytes of a record per base)
    with _Options(codec, opts):
        got = encode(codec, fq1, fq2, paired, chunk_bases)
        if "RFQ_SLICE_BYTES" not in opts:                                       # (a sliced call reports the stages of its slices summed: the names are there, but not worth a rule)
            _marker_ok(codec, marker)
    if want is not None:
        assert got == want, (form, len(got), len(want))
    else:
        assert len(got) == want_len and hashlib.md5(got).hexdigest() == want_md5, (form, len(got), want_len)
    return got


def check_decode_formulation(codec, form, rfq, split, want, chunk_bases=1_000_000):
    """one image through rfq_decode_batch with the switches of DEC_FORMS[form]: the text equals `want` (bytes; a pair with split)"""
    opts, marker = DEC_FORMS[form]
    opts = dict(opts)
    if "RFQ_SLICE_BASES" in opts:
        opts["RFQ_SLICE_BASES"] = str(int(2.5 * chunk_bases))
    with _Options(codec, opts):
        got = codec.decode_bytes(rfq, split_pe=split)
        _marker_ok(codec, marker)
    assert got == want, (form, [len(x) for x in (got if split else (got,))])
