"""ctypes binding of the C-ABI in include/rfq_hip.h (librfq_hip.so, built by hipcc for gfx950).

There is no CPU implementation behind this module: if the shared library is missing or no GPU is usable, loading /
Engine() raises.  (tests/ may point RFQ_HIP_LIBRARY at tests/emu/librfq_emu.so, the SIMT-interpreter TEST build of the
same sources, to debug kernel logic on a GPU-less box; nothing in repaq_amd does that on its own.)"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "librfq_hip.so")

RFQ_OK = 0
SE, PE_TWO_FILES, PE_INTERLEAVED = 0, 1, 2
HEADER_MAX = 17 + 255
U64_MAX = (1 << 64) - 1
ERRORS = {-1: "RFQ_E_NO_DEVICE", -2: "RFQ_E_HIP", -3: "RFQ_E_ARG", -4: "RFQ_E_TEXT", -5: "RFQ_E_DATA", -6: "RFQ_E_FORMAT",
          -7: "RFQ_E_UNPINNED", -8: "RFQ_E_NOSPACE", -9: "RFQ_E_STATE"}


class RfqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERRORS.get(code, code), msg))
        self.code = code
        self.message = msg


class EncodeArgs(C.Structure):
    _fields_ = [("d_fq1", C.c_void_p), ("n1", C.c_size_t), ("d_fq2", C.c_void_p), ("n2", C.c_size_t), ("paired", C.c_int32),
                ("chunk_bases", C.c_uint32), ("final", C.c_int32), ("emit_header", C.c_int32), ("file_off1", C.c_uint64),
                ("file_off2", C.c_uint64), ("nolb_from1", C.c_uint64), ("nolb_from2", C.c_uint64), ("d_out", C.c_void_p), ("out_cap", C.c_size_t),
                ("flush_all", C.c_int32), ("carry_bases", C.c_uint32)]


class EncodeResult(C.Structure):
    _fields_ = [("d_rfq", C.c_void_p), ("rfq_len", C.c_size_t), ("n_chunks", C.c_uint32), ("n_reads", C.c_uint64), ("n_bases", C.c_uint64),
                ("consumed1", C.c_size_t), ("consumed2", C.c_size_t), ("h_chunk_off", C.POINTER(C.c_uint64)),
                ("input_ended", C.c_int32), ("reserved", C.c_int32)]


class ScanResult(C.Structure):
    _fields_ = [("n_chunks", C.c_uint32), ("n_reads", C.c_uint64), ("consumed1", C.c_size_t), ("consumed2", C.c_size_t),
                ("h_end1", C.POINTER(C.c_uint64)), ("h_end2", C.POINTER(C.c_uint64)), ("input_ended", C.c_int32), ("unit_bases", C.c_uint32)]


class DecodeArgs(C.Structure):
    _fields_ = [("d_rfq", C.c_void_p), ("n", C.c_size_t), ("has_header", C.c_int32), ("split_pe", C.c_int32), ("final", C.c_int32), ("bug_compat", C.c_int32),
                ("d_out1", C.c_void_p), ("cap1", C.c_size_t), ("d_out2", C.c_void_p), ("cap2", C.c_size_t),
                ("h_chunk_off", C.POINTER(C.c_uint64)), ("n_chunk_off", C.c_uint32), ("reserved3", C.c_uint32)]


class DecodeResult(C.Structure):
    _fields_ = [("d_fq1", C.c_void_p), ("n1", C.c_size_t), ("d_fq2", C.c_void_p), ("n2", C.c_size_t), ("n_chunks", C.c_uint32),
                ("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("consumed", C.c_size_t)]


_libs = {}


def load(path=None):
    path = path or os.environ.get("RFQ_HIP_LIBRARY") or DEFAULT_LIB
    if path in _libs:
        return _libs[path]
    try:
        # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HSA runtimes in one process cannot both
        # open the GPU, so when torch is around let it load its runtime first; librfq_hip.so then binds to the same one
        # (same SONAME).  Hosts without torch (the C++ driver) simply use /opt/rocm's.
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(path):
        raise ImportError("librfq_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); repaq_amd has no CPU fallback" % path)
    L = C.CDLL(path)
    L.rfq_version.restype = C.c_char_p
    L.rfq_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.rfq_destroy.argtypes = [C.c_void_p]
    L.rfq_last_error.argtypes = [C.c_void_p]; L.rfq_last_error.restype = C.c_char_p
    L.rfq_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.rfq_set_header.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.rfq_get_header.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t)]
    L.rfq_clear_header.argtypes = [C.c_void_p]; L.rfq_clear_header.restype = None
    L.rfq_encode_batch.argtypes = [C.c_void_p, C.POINTER(EncodeArgs), C.POINTER(EncodeResult)]
    L.rfq_decode_batch.argtypes = [C.c_void_p, C.POINTER(DecodeArgs), C.POINTER(DecodeResult)]
    L.rfq_scan_batch.argtypes = [C.c_void_p, C.POINTER(EncodeArgs), C.POINTER(ScanResult)]
    L.rfq_last_timings.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    L.rfq_dev_malloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]
    L.rfq_dev_free.argtypes = [C.c_void_p, C.c_void_p]
    L.rfq_copy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    L.rfq_copy_h2d_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.rfq_copy_done.argtypes = [C.c_void_p, C.c_uint64]
    L.rfq_copy_sync.argtypes = [C.c_void_p]
    L.rfq_copy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.rfq_copy_d2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.rfq_copy_peer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.rfq_host_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]
    L.rfq_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.rfq_compare_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.rfq_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.rfq_get_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.rfq_option_name.argtypes = [C.c_int]; L.rfq_option_name.restype = C.c_char_p
    L.rfq_selftest_wave.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint64)]
    _libs[path] = L
    return L


EXPORTS = ["rfq_version", "rfq_create", "rfq_destroy", "rfq_last_error", "rfq_set_stream", "rfq_set_header", "rfq_get_header", "rfq_clear_header",
           "rfq_encode_batch", "rfq_scan_batch", "rfq_decode_batch", "rfq_last_timings", "rfq_dev_malloc", "rfq_dev_free", "rfq_copy_h2d", "rfq_copy_d2h",
           "rfq_copy_h2d_async", "rfq_copy_done", "rfq_copy_sync",
           "rfq_copy_d2d", "rfq_copy_peer", "rfq_host_alloc", "rfq_host_free", "rfq_compare_bytes", "rfq_selftest_wave", "rfq_set_option", "rfq_get_option", "rfq_option_name", "rfq_host_register", "rfq_host_unregister"]
