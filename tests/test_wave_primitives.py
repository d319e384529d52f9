"""The wave scans / reductions of rfq_common.h (DPP row shifts + row broadcasts on gfx950, shuffles under the SIMT interpreter) against a serial
reference, through rfq_selftest_wave: every kernel's prefix sums, streak scans and reductions are built on them, and the interpreter cannot see the
DPP forms - the GPU half of this file is what pins them."""
import random

import pytest

import _engine as E

M32, M64 = (1 << 32) - 1, (1 << 64) - 1


def _s32(x):
    x &= M32
    return x - (1 << 32) if x >> 31 else x


def _s64(x):
    x &= M64
    return x - (1 << 64) if x >> 63 else x


def _patterns():
    rng = random.Random(20260928)
    pats = [[0] * 64, [1] * 64, [M64] * 64, list(range(64)), list(range(63, -1, -1)), [1 << (i % 64) for i in range(64)],
            [(i * 0x9E3779B97F4A7C15) & M64 for i in range(64)], [0x80000000] * 64, [0x7FFFFFFF + (i & 1) for i in range(64)]]
    for k in range(64):                                                   # a single non-zero lane, at every lane: what a wrong row / bank mask would drop
        v = [0] * 64; v[k] = 0xF00DFACE12345678 ^ k; pats.append(v)
    for k in (15, 16, 31, 32, 47, 48):                                    # steps at the row borders of 16 lanes
        pats.append([5 if i <= k else 0xFFFFFFF0 for i in range(64)])
    for _ in range(40):
        pats.append([rng.getrandbits(64) for _ in range(64)])
    for _ in range(20):
        pats.append([rng.getrandbits(rng.choice((1, 4, 16, 31))) for _ in range(64)])
    return pats


def _check(codec):
    pats = _patterns()
    flat = [v for p in pats for v in p]
    got = codec.selftest_wave(flat)
    for pi, p in enumerate(pats):
        a = [v & M32 for v in p]
        s32 = s64 = 0; m32 = None; m64 = None
        u4 = [0, 0, 0, 0]
        for l in range(64):
            o = got[64 * pi + l]; tag = (pi, l)
            s32 = (s32 + a[l]) & M32; s64 = (s64 + p[l]) & M64
            m32 = _s32(a[l]) if m32 is None else max(m32, _s32(a[l])); m64 = _s64(p[l]) if m64 is None else max(m64, _s64(p[l]))
            for k, x in enumerate((a[l], a[l] >> 3, a[l] ^ 0x5A5A, p[l] >> 32)):
                u4[k] = (u4[k] + x) & M32
            assert o[0] == s32, tag
            assert o[1] == s64, tag
            assert _s64(o[2]) == m32, tag
            assert _s64(o[3]) == m64, tag
            assert o[4] == sum(a) & M32, tag
            assert o[5] == min(a) and o[6] == max(a), tag
            land = lor = a[0]
            for x in a: land &= x; lor |= x
            assert o[7] == (land << 32) | lor, tag
            assert o[8] == (a[l - 1] if l else 0xABCD1234), tag
            assert o[9] == a[63], tag
            assert o[10] == min(p), tag
            assert o[11] == (((u4[0] + u4[1] + u4[2]) & M32) << 32) | u4[3], tag


def test_wave_primitives_on_simt_emulation():
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.build_emu())
    try:
        _check(c)
    finally:
        c.close()


@pytest.mark.gpu
def test_wave_primitives_dpp_on_gpu():
    import torch
    assert torch.cuda.is_available()
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=E.PRODUCT_LIB)
    try:
        assert "gfx950" in c.version()
        _check(c)
    finally:
        c.close()
