"""The host driver's blocked-gzip writer and reader (repaq_amd/csrc/host/repaq_hip_main.cpp: ByteSink / ByteSource), driven directly
through a small harness: text sizes around the 0xff00-byte members, sink pieces and source reads of any size (a read smaller than a member
spills it, a tiny read-ahead refills in the middle of members), files that change from blocked to plain gzip and back."""
import gzip
import os
import random
import struct
import subprocess
import zlib

import pytest

import _engine as E

HERE = os.path.dirname(os.path.abspath(__file__))
BLOCK = 0xFF00


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    E.build_emu()
    exe = str(tmp_path_factory.mktemp("gzh") / "gz_harness")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(E.ROOT, "include"), os.path.join(HERE, "gz_harness.cpp"), "-o", exe,
                           "-L" + E.EMU_DIR, "-lrfq_emu", "-Wl,-rpath," + E.EMU_DIR, "-lpthread", "-lz"])
    return exe


def members(z):
    """sizes of the BGZF members of z (asserts the layout)"""
    out, off = [], 0
    while off < len(z):
        assert z[off:off + 4] == b"\x1f\x8b\x08\x04" and z[off + 10:off + 16] == b"\x06\x00BC\x02\x00", off
        n = (z[off + 16] | (z[off + 17] << 8)) + 1
        out.append(n); off += n
    assert off == len(z)
    return out


def text_of(n, seed):
    r = random.Random(seed)
    alpha = b"ACGTN@+FF:,#\n"
    return bytes(r.choice(alpha) for _ in range(min(n, 4096))) * (n // 4096 + 1) if n > 4096 else bytes(r.choice(alpha) for _ in range(n))


def bgzf(data, level=6):
    out = b""
    for i in range(0, len(data), BLOCK):
        blk = data[i:i + BLOCK]; c = zlib.compressobj(level, zlib.DEFLATED, -15); zd = c.compress(blk) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(zd) + 8 - 1) + zd + struct.pack("<II", zlib.crc32(blk), len(blk))
    return out


@pytest.mark.parametrize("n", [0, 1, BLOCK - 1, BLOCK, BLOCK + 1, 3 * BLOCK, 3 * BLOCK + 5, (1 << 20) + 7, 5_000_003])
def test_writer_output_is_blocked_gzip_of_the_text(harness, tmp_path, n):
    text = text_of(n, n)[:n]
    for piece in (1 if n <= 70000 else 997, 1000, BLOCK, BLOCK + 1, 1 << 20):
        p = str(tmp_path / "w.fq.gz")
        subprocess.run([harness, "w", p, str(piece), "3"], input=text, check=True)
        z = open(p, "rb").read()
        assert gzip.decompress(z) == text
        m = members(z)
        assert m[-1] == 28 and len(m) == (n + BLOCK - 1) // BLOCK + 1          # full members, the short last one, the empty end marker


@pytest.mark.parametrize("n", [0, 1, BLOCK, BLOCK + 1, 4 * BLOCK + 9, (1 << 20) + 7, 3_000_001])
def test_reader_inflates_blocked_gzip_at_any_read_size(harness, tmp_path, n):
    text = text_of(n, 7 * n + 1)[:n]
    p = str(tmp_path / "r.fq.gz"); open(p, "wb").write(bgzf(text) + bgzf(b"")[:0] + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    for cap in (1 if n <= 70000 else 4093, 100 if n <= 300000 else 65537, 65536, 1 << 20):
        for zb in ("64", "70000", None):
            env = dict(os.environ); env.pop("RFQ_GZ_BUF", None)
            if zb:
                env["RFQ_GZ_BUF"] = zb
            r = subprocess.run([harness, "r", p, str(cap)], capture_output=True, env=env)
            assert r.returncode == 0 and r.stdout == text, (n, cap, zb, r.stderr[-200:])


def test_reader_follows_a_file_that_changes_gzip_flavour(harness, tmp_path):
    a, b, c = text_of(200_000, 1)[:200_000], text_of(150_001, 2)[:150_001], text_of(70_000, 3)[:70_000]
    p = str(tmp_path / "m.fq.gz")
    open(p, "wb").write(bgzf(a) + gzip.compress(b, 1) + bgzf(c))              # blocked, then one plain member, then blocked again (read by zlib from the change on)
    for cap in (1000, 1 << 20):
        r = subprocess.run([harness, "r", p, str(cap)], capture_output=True)
        assert r.returncode == 0 and r.stdout == a + b + c
    open(p, "wb").write(gzip.compress(a, 1) + bgzf(c))                         # plain first: zlib's reader throughout
    r = subprocess.run([harness, "r", p, "65536"], capture_output=True)
    assert r.returncode == 0 and r.stdout == a + c


def test_reader_refuses_a_damaged_member(harness, tmp_path):
    text = text_of(3 * BLOCK, 5)[:3 * BLOCK]; z = bytearray(bgzf(text))
    z[len(z) // 2] ^= 0x55
    p = str(tmp_path / "bad.fq.gz"); open(p, "wb").write(bytes(z))
    r = subprocess.run([harness, "r", p, "65536"], capture_output=True)
    assert r.returncode != 0 and b"Error to read gzip file" in r.stderr
