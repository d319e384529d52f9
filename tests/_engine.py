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


def build_emu():
    subprocess.check_call(["make", "-s", "-j4", "-C", EMU_DIR])
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


def decode_in_slices(codec, rfq: bytes, split_pe: bool, step: int):
    """Feed an image `step` bytes at a time (has_header on the first call, final on the last, unconsumed tail carried over): the
    streaming contract of rfq_decode_batch that the C++ driver relies on."""
    out1, out2 = bytearray(), bytearray()
    pos, end, first = 0, min(step, len(rfq)), True
    while True:
        final = end == len(rfq)
        buf = rfq[pos:end]
        d = codec.dev_put(buf)
        try:
            r = codec.decode(d, len(buf), has_header=first, split_pe=split_pe, final=final)
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
