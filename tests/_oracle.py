"""ctypes bindings for the test-only CPU oracle (oracle/liboracle.so) and the FASTQ generator
(tools/libfqgen.so).  Imported by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
FQGEN_SO = os.path.join(ROOT, "tools", "libfqgen.so")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "repaq")
REF_HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")

SE, PE_TWO_FILES, PE_INTERLEAVED = 0, 1, 2
NOVA_SE150, NOVA_PE150, SE_VAR, BGI_PE100 = 0, 1, 2, 3


def build():
    """(Re)build liboracle.so / libfqgen.so when sources are newer (gcc only)."""
    src = os.path.join(ROOT, "oracle", "rfq_oracle.c")
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), ORACLE_SO])
    gsrc = os.path.join(ROOT, "tools", "fqgen.c")
    if not os.path.exists(FQGEN_SO) or os.path.getmtime(FQGEN_SO) < os.path.getmtime(gsrc):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", gsrc, "-o", FQGEN_SO])


class _Meta(C.Structure):
    _fields_ = [("ok", C.c_int), ("name1_len", C.c_uint32), ("name2_off", C.c_uint32), ("name2_len", C.c_uint32),
                ("lane", C.c_uint8), ("tile", C.c_uint16), ("x", C.c_uint32), ("y", C.c_uint32)]


class _GenParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_reads", C.c_uint64), ("profile", C.c_int32), ("n_rate_ppm", C.c_uint32),
                ("no_trailing_newline", C.c_int32), ("interleaved", C.c_int32), ("n_quals", C.c_int32), ("reserved", C.c_int32)]


_lib = None
_gen = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_SO)
        L.rfqo_encode_file.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_uint32,
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_char_p]
        L.rfqo_encode_file.restype = C.c_int
        L.rfqo_decode_file.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_char_p]
        L.rfqo_decode_file.restype = C.c_int
        L.rfqo_decode_file_compat.argtypes = L.rfqo_decode_file.argtypes
        L.rfqo_decode_file_compat.restype = C.c_int
        L.rfqo_free.argtypes = [C.c_void_p]
        L.rfqo_parse_name.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(_Meta)]
        L.rfqo_overlap.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.rfqo_overlap.restype = C.c_int
        L.rfqo_encode_coords.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_char_p]
        L.rfqo_encode_coords.restype = C.c_int64
        L.rfqo_decode_coords.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]
        L.rfqo_pos_encode.argtypes = [C.c_char_p, C.c_uint32, C.c_uint8, C.c_char_p, C.c_char_p]
        L.rfqo_pos_encode.restype = C.c_uint32
        L.rfqo_pos_decode.argtypes = [C.c_char_p, C.c_uint32, C.c_uint8, C.c_char_p, C.c_uint32]
        L.rfqo_chunk_table.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, C.c_char_p]
        L.rfqo_chunk_table.restype = C.c_int64
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def encode_file(fq1: bytes, fq2: bytes = b"", paired: int = SE, chunk_bases: int = 1_000_000) -> bytes:
    L = lib()
    out = C.c_void_p(); n = C.c_size_t(); err = C.create_string_buffer(256)
    rc = L.rfqo_encode_file(fq1, len(fq1), fq2 if fq2 else None, len(fq2), paired, chunk_bases, C.byref(out), C.byref(n), err)
    if rc:
        raise OracleError(err.value.decode(errors="replace"))
    try:
        return C.string_at(out, n.value)
    finally:
        L.rfqo_free(out)


def decode_file(rfq: bytes, split_pe: bool = False, bug_compat: bool = False):
    """bug_compat: the reference's decompress loops as they stand - a chunk behind a non-last NO_LINE_BREAK chunk is lost (src/repaq.cpp:303-325,376-403)."""
    L = lib()
    o1 = C.c_void_p(); n1 = C.c_size_t(); o2 = C.c_void_p(); n2 = C.c_size_t(); err = C.create_string_buffer(256)
    rc = (L.rfqo_decode_file_compat if bug_compat else L.rfqo_decode_file)(rfq, len(rfq), 1 if split_pe else 0, C.byref(o1), C.byref(n1), C.byref(o2), C.byref(n2), err)
    if rc:
        raise OracleError(err.value.decode(errors="replace"))
    try:
        a = C.string_at(o1, n1.value) if n1.value else b""
        b = C.string_at(o2, n2.value) if n2.value else b""
    finally:
        L.rfqo_free(o1); L.rfqo_free(o2)
    return (a, b) if split_pe else a


def chunk_table(rfq: bytes):
    L = lib()
    cap = max(16, len(rfq) // 64)
    offs = (C.c_uint64 * cap)(); err = C.create_string_buffer(256)
    n = L.rfqo_chunk_table(rfq, len(rfq), offs, cap, err)
    if n < 0:
        raise OracleError(err.value.decode(errors="replace"))
    return [offs[i] for i in range(n + 1)]


def parse_name(name: bytes):
    m = _Meta()
    lib().rfqo_parse_name(name, len(name), C.byref(m))
    n1 = name[: m.name1_len]; n2 = name[m.name2_off: m.name2_off + m.name2_len]
    return (m.ok, n1, m.lane, m.tile, m.x, m.y, n2)


def overlap(a: bytes, b: bytes) -> int:
    return lib().rfqo_overlap(a, len(a), b, len(b))


def encode_coords(vals):
    arr = (C.c_uint32 * len(vals))(*vals)
    out = C.create_string_buffer(len(vals) * 3 + 8)
    n = lib().rfqo_encode_coords(arr, len(vals), out)
    if n < 0:
        raise OracleError("The X/Y coordinate cannot be larger than 2M")
    return out.raw[:n]


def decode_coords(stream: bytes, num: int):
    out = (C.c_uint32 * max(1, num))()
    lib().rfqo_decode_coords(stream, len(stream), out, num)
    return list(out[:num])


def pos_encode(buf: bytes, q: int) -> bytes:
    out = C.create_string_buffer(len(buf) * 4 + 16)
    n = lib().rfqo_pos_encode(buf, len(buf), q, out, None)
    return out.raw[:n]


def pos_decode(stream: bytes, q: int, base: bytes) -> bytes:
    out = C.create_string_buffer(base, len(base))
    lib().rfqo_pos_decode(stream, len(stream), q, out, len(base))
    return out.raw[: len(base)]


def gen(profile: int, n_reads: int, seed: int = 1, nppm: int = 20, nonl: int = 0, interleaved: bool = False, n_quals: int = 13):
    """Returns (fq1, fq2) bytes (fq2 == b'' for SE / interleaved)."""
    global _gen
    if _gen is None:
        build()
        _gen = C.CDLL(FQGEN_SO)
        _gen.fqgen_generate.argtypes = [C.POINTER(_GenParams), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _gen.fqgen_generate.restype = C.c_int
    p = _GenParams(seed, n_reads, profile, nppm, nonl, 1 if interleaved else 0, n_quals, 0)
    n1 = C.c_size_t(); n2 = C.c_size_t()
    _gen.fqgen_generate(C.byref(p), None, 0, None, 0, C.byref(n1), C.byref(n2))
    b1 = C.create_string_buffer(max(1, n1.value)); b2 = C.create_string_buffer(max(1, n2.value))
    rc = _gen.fqgen_generate(C.byref(p), b1, n1.value, b2, n2.value, C.byref(n1), C.byref(n2))
    assert rc == 0
    return b1.raw[: n1.value], b2.raw[: n2.value]


def gen_np(profile: int, n_reads: int, seed: int = 1, nppm: int = 20, nonl: int = 0, interleaved: bool = False, n_quals: int = 13):
    """gen() for multi-GB inputs: the generator writes straight into numpy uint8 arrays (no bytes copies)."""
    import numpy as np
    gen(profile, 1, seed)                      # loads the library / prototypes
    p = _GenParams(seed, n_reads, profile, nppm, nonl, 1 if interleaved else 0, n_quals, 0)
    n1 = C.c_size_t(); n2 = C.c_size_t()
    if profile in (NOVA_SE150, NOVA_PE150):
        # one pass: a NovaSeq-profile record is at most 54 (name) + 150 + 1 + 150 + 4 line ends = 359 bytes (untouched pages cost nothing)
        per = 360 * (2 if interleaved else 1)
        n1.value = n_reads * per + 64; n2.value = (n_reads * per + 64) if (profile == NOVA_PE150 and not interleaved) else 0
    else:
        _gen.fqgen_generate(C.byref(p), None, 0, None, 0, C.byref(n1), C.byref(n2))
    a1 = np.empty(max(1, n1.value), dtype=np.uint8); a2 = np.empty(max(1, n2.value), dtype=np.uint8)
    rc = _gen.fqgen_generate(C.byref(p), a1.ctypes.data_as(C.c_void_p), n1.value, a2.ctypes.data_as(C.c_void_p), n2.value, C.byref(n1), C.byref(n2))
    assert rc == 0
    return a1[: n1.value], a2[: n2.value]


def have_ref() -> bool:
    return os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)


def ref_encode(fq1: bytes, fq2: bytes = b"", paired: int = SE, k: int = 1000, tmpdir: str = "/tmp") -> bytes:
    """Run the compiled reference binary (this container only) on files; returns the .rfq bytes."""
    import tempfile
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        p1 = os.path.join(d, "a_1.fq"); p2 = os.path.join(d, "a_2.fq"); o = os.path.join(d, "o.rfq")
        open(p1, "wb").write(fq1)
        cmd = [REF_BIN, "-c", "-i", p1, "-o", o, "-k", str(k)]
        if paired == PE_TWO_FILES:
            open(p2, "wb").write(fq2); cmd += ["-I", p2]
        elif paired == PE_INTERLEAVED:
            cmd += ["--interleaved_in"]
        r = subprocess.run(cmd, capture_output=True)
        if r.returncode != 0:
            raise OracleError(r.stderr.decode(errors="replace"))
        return open(o, "rb").read()


def ref_decode(rfq: bytes, split_pe: bool = False, tmpdir: str = "/tmp"):
    import tempfile
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        i = os.path.join(d, "i.rfq"); o1 = os.path.join(d, "o_1.fq"); o2 = os.path.join(d, "o_2.fq")
        open(i, "wb").write(rfq)
        cmd = [REF_BIN, "-d", "-i", i, "-o", o1] + (["-O", o2] if split_pe else [])
        r = subprocess.run(cmd, capture_output=True)
        if r.returncode != 0:
            raise OracleError(r.stderr.decode(errors="replace"))
        a = open(o1, "rb").read()
        return (a, open(o2, "rb").read()) if split_pe else a
