"""Host-side mirror of the reference's RfqCodec seam (src/rfqcodec.h:17-43) over the C-ABI.

    reference                                   here
    RfqCodec codec;                             codec = RfqCodec(device=0)
    codec.setHeader(h)                          codec.setHeader(header_bytes)
    h = codec.makeHeader(reads)  (chunk 0)      implicit in the first encode (header from the first chunk), codec.header()
    chunk = codec.encodeChunk(reads); write     codec.encode(d_fq1, n1, ...) -> EncodeResult (all chunks of the batch)
    reads = codec.decodeChunk(chunk)            codec.decode(d_rfq, n, ...)  -> DecodeResult (FASTQ text of all chunks)

Pointers are device pointers (e.g. torch.uint8 CUDA tensors' data_ptr()).  The *_bytes helpers move host bytes through
rfq_dev_malloc / rfq_copy_* for tests and small tools."""
import ctypes as C

from . import _capi as A
from ._capi import RfqError, SE, PE_TWO_FILES, PE_INTERLEAVED, U64_MAX  # noqa: F401


class RfqCodec:
    def __init__(self, device=0, library=None):
        self._L = A.load(library)
        h = C.c_void_p()
        rc = self._L.rfq_create(C.byref(h), device)
        if rc != A.RFQ_OK:
            raise RfqError(rc, "rfq_create(device=%d) failed: no usable MI355X / HIP device (there is no CPU fallback)" % device)
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.rfq_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != A.RFQ_OK:
            raise RfqError(rc, self._L.rfq_last_error(self._h).decode("latin-1"))

    def version(self):
        return self._L.rfq_version().decode()

    def set_stream(self, stream_ptr):
        self._check(self._L.rfq_set_stream(self._h, stream_ptr))

    def set_option(self, name, value=None):
        """rfq_set_option: a test / diagnostic switch of this context (names = the RFQ_* environment variables; None = default)."""
        self._check(self._L.rfq_set_option(self._h, name.encode(), None if value is None else str(value).encode()))

    def get_option(self, name) -> str:
        """rfq_get_option: the switch's current value ("" = default)"""
        buf = C.create_string_buffer(64)
        self._check(self._L.rfq_get_option(self._h, name.encode(), buf, 64))
        return buf.value.decode()

    def option_names(self):
        """rfq_option_name: every switch the library knows"""
        out, i = [], 0
        while True:
            n = self._L.rfq_option_name(i)
            if not n:
                return out
            out.append(n.decode()); i += 1

    def option(self, name, value):
        """with codec.option("RFQ_GATHER", "old"): ... - the switch set for the block, back to what it WAS afterwards (an environment-provided
        or earlier value, not necessarily the built-in default)"""
        import contextlib

        @contextlib.contextmanager
        def scope():
            before = self.get_option(name)
            self.set_option(name, value)
            try:
                yield self
            finally:
                self.set_option(name, before or None)
        return scope()

    def selftest_wave(self, lanes):
        """rfq_selftest_wave: the wave scans / reductions of rfq_common.h on `lanes` (64 * k u64 values) -> 12 u64 per lane."""
        n = len(lanes); assert n and n % 64 == 0
        a = (C.c_uint64 * n)(*lanes); o = (C.c_uint64 * (12 * n))()
        self._check(self._L.rfq_selftest_wave(self._h, a, n // 64, o))
        return [list(o[12 * i: 12 * i + 12]) for i in range(n)]

    # --- RfqCodec::setHeader / header accessors
    def setHeader(self, header_bytes: bytes):
        self._check(self._L.rfq_set_header(self._h, header_bytes, len(header_bytes)))

    def clearHeader(self):
        self._L.rfq_clear_header(self._h)

    def header(self) -> bytes:
        buf = C.create_string_buffer(A.HEADER_MAX); n = C.c_size_t()
        self._check(self._L.rfq_get_header(self._h, buf, C.byref(n)))
        return buf.raw[: n.value]

    # --- RfqCodec::encodeChunk for every chunk of a batch
    def encode(self, d_fq1, n1, d_fq2=None, n2=0, paired=SE, chunk_bases=1_000_000, final=True, emit_header=True,
               file_off1=0, file_off2=0, nolb_from1=U64_MAX, nolb_from2=U64_MAX, d_out=None, out_cap=0, flush_all=False):
        a = A.EncodeArgs(d_fq1, n1, d_fq2, n2, paired, chunk_bases, 1 if final else 0, 1 if emit_header else 0,
                         file_off1, file_off2, nolb_from1, nolb_from2, d_out, out_cap, 1 if flush_all else 0, 0)
        r = A.EncodeResult()
        self._check(self._L.rfq_encode_batch(self._h, C.byref(a), C.byref(r)))
        return r

    # --- the plan pass of a chunk-parallel encode: where every chunk ends in the input stream(s)
    def scan(self, d_fq1, n1, d_fq2=None, n2=0, paired=SE, chunk_bases=1_000_000, final=True, file_off1=0, file_off2=0, carry_bases=0):
        """rfq_scan_batch: the plan pass.  carry_bases: bases the chunk that is open at the start of this text took from the text in front of it."""
        a = A.EncodeArgs(d_fq1, n1, d_fq2, n2, paired, chunk_bases, 1 if final else 0, 0, file_off1, file_off2, U64_MAX, U64_MAX, None, 0, 0, carry_bases)
        r = A.ScanResult()
        self._check(self._L.rfq_scan_batch(self._h, C.byref(a), C.byref(r)))
        ends1 = [r.h_end1[i] for i in range(r.n_chunks)]
        ends2 = [r.h_end2[i] for i in range(r.n_chunks)] if (paired == PE_TWO_FILES and r.n_chunks) else []
        return r, ends1, ends2

    # --- RfqCodec::decodeChunk for every chunk of an image
    def decode(self, d_rfq, n, has_header=True, split_pe=False, final=True, d_out1=None, cap1=0, d_out2=None, cap2=0, chunk_off=None, n_chunks=0, bug_compat=False):
        """chunk_off / n_chunks: optional chunk index (EncodeResult.h_chunk_off + n_chunks, or a sequence of n_chunks + 1 offsets).
        bug_compat: with split_pe, lose what Repaq::decompressPE loses behind a non-last NO_LINE_BREAK chunk (src/repaq.cpp:376-403); Repaq::decompress (one output) loses nothing."""
        if chunk_off is not None and not isinstance(chunk_off, C.POINTER(C.c_uint64)):
            n_chunks = len(chunk_off) - 1
            chunk_off = C.cast((C.c_uint64 * len(chunk_off))(*chunk_off), C.POINTER(C.c_uint64))
        a = A.DecodeArgs(d_rfq, n, 1 if has_header else 0, 1 if split_pe else 0, 1 if final else 0, 1 if bug_compat else 0, d_out1, cap1, d_out2, cap2,
                         chunk_off if (chunk_off is not None and n_chunks) else None, n_chunks if chunk_off is not None else 0, 0)
        r = A.DecodeResult()
        self._check(self._L.rfq_decode_batch(self._h, C.byref(a), C.byref(r)))
        return r

    # --- --compare on the device: first offset at which two device texts differ (n when identical)
    def first_diff(self, d_a, d_b, n) -> int:
        out = C.c_uint64(0)
        self._check(self._L.rfq_compare_bytes(self._h, d_a, d_b, n, C.byref(out)))
        return out.value

    def timings(self):
        names = (C.c_char_p * 32)(); ms = (C.c_float * 32)()
        n = self._L.rfq_last_timings(self._h, names, ms, 32)
        return [(names[i].decode(), ms[i]) for i in range(n)]

    # --- host-bytes conveniences (tests, small tools)
    def dev_put(self, data: bytes):
        p = C.c_void_p()
        self._check(self._L.rfq_dev_malloc(self._h, C.byref(p), max(len(data), 1) + 64))
        self._check(self._L.rfq_copy_h2d(self._h, p, data, len(data)))
        return p

    def dev_get(self, d_ptr, n) -> bytes:
        buf = C.create_string_buffer(max(n, 1))
        self._check(self._L.rfq_copy_d2h(self._h, buf, d_ptr, n))
        return buf.raw[:n]

    def dev_free(self, p):
        self._L.rfq_dev_free(self._h, p)

    def encode_bytes(self, fq1: bytes, fq2: bytes = b"", paired=SE, chunk_bases=1_000_000, **kw) -> bytes:
        d1 = self.dev_put(fq1); d2 = self.dev_put(fq2) if paired == PE_TWO_FILES else None
        try:
            r = self.encode(d1, len(fq1), d2, len(fq2) if d2 else 0, paired, chunk_bases, **kw)
            return self.dev_get(r.d_rfq, r.rfq_len) if r.rfq_len else b""
        finally:
            self.dev_free(d1)
            if d2:
                self.dev_free(d2)

    def decode_bytes(self, rfq: bytes, split_pe=False, out_caps=None, **kw):
        """out_caps = (cap1, cap2): decode into buffers of the caller (allocated here, of those sizes) instead of the context's own result buffers - the path a
        pipelined host takes (rfq_decode_args.d_out1 / d_out2), on which the emitter is launched ahead of the host's look at the status."""
        d = self.dev_put(rfq); o1 = o2 = None
        try:
            if out_caps is not None:
                o1 = self.dev_put(b"\0" * max(int(out_caps[0]), 1)); o2 = self.dev_put(b"\0" * max(int(out_caps[1]), 1)) if split_pe else None
                kw = dict(kw, d_out1=o1, cap1=int(out_caps[0]), d_out2=o2, cap2=int(out_caps[1]) if split_pe else 0)
            r = self.decode(d, len(rfq), split_pe=split_pe, **kw)
            a = self.dev_get(r.d_fq1, r.n1) if r.n1 else b""
            b = self.dev_get(r.d_fq2, r.n2) if (split_pe and r.n2) else b""
            return (a, b) if split_pe else a
        finally:
            self.dev_free(d)
            if o1 is not None: self.dev_free(o1)
            if o2 is not None: self.dev_free(o2)


def nolb_threshold(file_size: int, ends_with_newline: bool) -> int:
    """Offset from which the reference's reader has its "no line break at the end" flag up (src/fastqreader.cpp:31-46):
    the start of its final, short 1 MiB block when the file lacks a trailing newline.  A file of exactly k MiB has no short block: the
    flag goes up at the empty read behind the last full block - whatever the last byte is (the test reads the byte in front of the
    buffer there: never a line break in practice) - so the threshold is the file size itself: it is met by an unterminated last record
    and by the readers' last, failed attempt, i.e. by the input's tail chunk."""
    if file_size == 0:
        return U64_MAX
    if file_size % (1 << 20) == 0:
        return file_size
    if ends_with_newline:
        return U64_MAX
    return ((file_size - 1) >> 20) << 20
