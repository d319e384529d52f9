"""CPU: the product C-ABI library loads, exports every symbol include/rfq_hip.h declares, and fails loudly without a GPU.
No compute calls are made here."""
import ctypes as C
import os
import re

import pytest

import _engine as E

ROOT = E.ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "rfq_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rfq_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def product_lib():
    if not os.path.exists(E.PRODUCT_LIB):
        import __graft_entry__ as g
        g.build_hip()
    return C.CDLL(E.PRODUCT_LIB)


def test_header_declares_expected_entry_points():
    from repaq_amd import _capi
    assert _declared_symbols() == sorted(_capi.EXPORTS)


def test_library_exports_every_declared_symbol(product_lib):
    for sym in _declared_symbols():
        assert hasattr(product_lib, sym), "librfq_hip.so does not export %s" % sym


def test_library_exports_nothing_else(product_lib):
    """-fvisibility=hidden + RFQ_API: the dynamic symbol table holds the C entry points and nothing of the implementation
    (no __device_stub__ kernels, no internal C++ helpers)"""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", E.PRODUCT_LIB], text=True)
    syms = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    assert syms == _declared_symbols(), sorted(set(syms) ^ set(_declared_symbols()))


def test_version_string_is_gfx950(product_lib):
    product_lib.rfq_version.restype = C.c_char_p
    assert b"gfx950" in product_lib.rfq_version()


def test_create_fails_loudly_without_gpu(product_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = product_lib.rfq_create(C.byref(h), 0)
    assert rc == -1 and not h.value          # RFQ_E_NO_DEVICE: no CPU fallback
    from repaq_amd import RfqCodec, RfqError
    with pytest.raises(RfqError):
        RfqCodec(device=0, library=E.PRODUCT_LIB)


def test_product_sources_do_not_reference_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "repaq_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                t = open(os.path.join(base, f), errors="replace").read()
                if "liboracle" in t or "rfq_oracle" in t or "rfqo_" in t:
                    bad.append(f)
    assert not bad, bad


def test_bench_names_only_kernels_that_exist():
    """bench.py maps stage timers to kernel names (roofline.kernel, the traffic tables): every name must be a kernel of the sources (two deleted kernels were still listed)"""
    src = ""
    for base, _, files in os.walk(os.path.join(ROOT, "repaq_amd", "csrc")):
        for f in files:
            if f.endswith((".h", ".hip")):
                src += open(os.path.join(base, f), errors="replace").read()
    names = set(re.findall(r'"(k_[a-z0-9_]+[a-z0-9])"', open(os.path.join(ROOT, "bench.py")).read()))
    missing = sorted(n for n in names if not re.search(r"\b%s\b" % n, src))
    assert names and not missing, missing
