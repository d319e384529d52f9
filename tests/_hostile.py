"""Hostile .rfq images for the decoder.  The reference has exactly ONE bounds check on this path (src/rfqcodec.cpp:1041) and RfqChunk::read trusts
every length it reads (src/rfqchunk.cpp:161-228); the engine must not: whatever the bytes say, rfq_decode_batch returns an error or some text within a time bound,
never faults, and the SAME context decodes a good image correctly afterwards (no sticky state).

Mutant classes, all seeded:
  flip      bit flips anywhere in the image (1 - 3 per mutant)
  header    random bytes in the file header's fields behind the magic / version (read-length bytes, flags, name2 rule, N quality, overlap shift, bins, table)
  fixed     a chunk's fixed fields (mSize, reads, flags, sequence / quality / N-position sizes) set to edge values or random bytes
  lengths   random bytes in a chunk's length arrays (read lengths, name1 / name2 / strand lengths) and in the coordinate streams' size words
  quality   the quality payload's own length table (u32 per coded value) and its first stream bytes scribbled on
  truncate  the image cut at every section boundary of its first and last chunk (and a few bytes either side)
  index     the good image with a chunk index that lies (shifted, swapped, beyond the end, not monotonic, too many entries)
Used on the GPU (tests/test_gpu_hostile.py: every mode, full counts, in a child process - a fault would take the process down, and the test says so) and under
the SIMT interpreter (tests/test_emu_hostile.py: a bounded subset; tools/hostile_asan.sh runs it against an AddressSanitizer build of the same sources)."""
import random
import struct
import time

import _oracle as O
import _sections as S

ALLOWED = {-3: "ARG", -5: "DATA", -6: "FORMAT", -7: "UNPINNED", -8: "NOSPACE", -9: "STATE"}      # never -1 (device lost) / -2 (a HIP call failed)
EDGE32 = [0, 1, 2, 0x7F, 0xFF, 0x100, 0xFFFF, 0x10000, 0x7FFFFFFF, 0x80000000, 0xFFFFFFF0, 0xFFFFFFFF]


def images():
    """(label, image, split_pe, expected text(s)) - three shapes: SE with match masks, PE interleaved chunks with overlaps and N, variable read lengths"""
    out = []
    a, _ = O.gen(O.NOVA_SE150, 260, seed=91, nppm=4000)
    out.append(("se150", O.encode_file(a, b"", O.SE, 10000), False, a))
    a, b = O.gen(O.NOVA_PE150, 150, seed=92, nppm=6000)
    out.append(("pe150", O.encode_file(a, b, O.PE_TWO_FILES, 12000), True, (a, b)))
    a, _ = O.gen(O.SE_VAR, 220, seed=93)
    out.append(("se_var", O.encode_file(a, b"", O.SE, 8000), False, a))
    return out


def _put32(b, o, v):
    if o + 4 <= len(b):
        struct.pack_into("<I", b, o, v & 0xFFFFFFFF)


def mutants(img: bytes, seed: int, counts=None, tame=False):
    """[(label, image bytes, chunk index or None)].  tame: a chunk's read count is never raised beyond 300,000 (the interpreter walks every read a chunk
    claims - an image whose flags say "all the same" may claim 4 G reads in 30 bytes, which the GPU shrugs off and the interpreter does not)"""
    c = dict(flip=60, header=20, fixed=40, lengths=30, quality=20, truncate=1, index=12)
    c.update(counts or {})
    rng = random.Random(seed)
    h, chunks = S.parse(img)
    out = []
    for i in range(c["flip"]):
        b = bytearray(img)
        for _ in range(rng.randrange(1, 4)):
            k = rng.randrange(len(b)); b[k] ^= 1 << rng.randrange(8)
        out.append(("flip%d" % i, bytes(b), None))
    for i in range(c["header"]):
        b = bytearray(img)
        for _ in range(rng.randrange(1, 4)):
            k = rng.randrange(9, h.len); b[k] = rng.randrange(256)
        out.append(("header%d" % i, bytes(b), None))
    for i in range(c["fixed"]):
        b = bytearray(img); ch = chunks[rng.randrange(len(chunks))]
        field = rng.choice([0, 4, 8, 10, 14] + ([18] if ch.fixed > 18 else []))
        if field == 8:
            struct.pack_into("<H", b, ch.off + 8, rng.randrange(65536))
        elif rng.random() < 0.6:
            _put32(b, ch.off + field, rng.choice(EDGE32 + [struct.unpack_from("<I", img, ch.off + field)[0] + d for d in (-2, -1, 1, 2, 255)]))
        else:
            _put32(b, ch.off + field, rng.getrandbits(32) >> rng.randrange(0, 28))
        out.append(("fixed%d@%d+%d" % (i, ch.off, field), bytes(b), None))
    if tame:
        def reads_ok(b):
            return all(ch.off + 8 > len(b) or struct.unpack_from("<I", b, ch.off + 4)[0] <= 300000 for ch in chunks)
        out = [m for m in out if reads_ok(m[1])]
    for i in range(c["lengths"]):
        b = bytearray(img); ch = chunks[rng.randrange(len(chunks))]
        lo, hi = ch.off + ch.fixed, ch.off + ch.coords_end
        for _ in range(rng.randrange(1, 6)):
            k = rng.randrange(lo, max(lo + 1, hi)); b[k] = rng.choice([0, 1, 0x7F, 0x80, 0xFF, rng.randrange(256)])
        out.append(("lengths%d@%d" % (i, ch.off), bytes(b), None))
    for i in range(c["quality"]):
        b = bytearray(img); ch = chunks[rng.randrange(len(chunks))]
        lo = ch.off + ch.qual_off; nn = max(1, len(h.normal))
        if rng.random() < 0.5:
            _put32(b, lo + 4 * rng.randrange(nn), rng.choice(EDGE32 + [ch.qual_size, ch.qual_size + 1, ch.qual_size - 1]))
        else:
            for _ in range(rng.randrange(1, 8)):
                k = rng.randrange(lo, max(lo + 1, lo + min(ch.qual_size, 4 * nn + 64))); b[k] = rng.choice([0xFF, 0xE0, 0xFF, 0xC0, 0x80, rng.randrange(256)])
        out.append(("quality%d@%d" % (i, ch.off), bytes(b), None))
    if c["truncate"]:
        cuts = set()
        for ch in (chunks[0], chunks[-1]):
            for m in ch.marks:
                for d in (-1, 0, 1):
                    cuts.add(ch.off + m + d)
        cuts |= {0, 1, 3, 8, 9, 16, h.len - 1, h.len, h.len + 1, h.len + 11, h.len + 12, h.len + 17, h.len + 18, len(img) - 1}
        for k in sorted(x for x in cuts if 0 <= x < len(img)):
            out.append(("truncate@%d" % k, img[:k], None))
    offs = [ch.off for ch in chunks] + [chunks[-1].off + chunks[-1].total]
    lies = [[o + 1 for o in offs], [max(0, o - 1) for o in offs], offs[:1] + offs[2:], offs[:-1] + [len(img) + 4096], offs[:1] + offs[2:1:-1] + offs[3:],
            [offs[0]] * len(offs), offs + [offs[-1] + 7] * 3, [offs[0], offs[-1]], [offs[0]] + [o + 3 for o in offs[1:-1]] + [offs[-1]], [0] + offs[1:],
            [offs[0]] + [rng.randrange(len(img)) for _ in offs[1:-1]] + [offs[-1]], [offs[0]] + sorted(rng.randrange(offs[0], len(img)) for _ in offs[1:-1]) + [offs[-1]]]
    for i, t in enumerate(lies[:c["index"]]):
        if len(t) >= 2:
            out.append(("index%d" % i, img, t))
    return out


def run(codec, modes=((),), counts=None, seed=7, time_bound_s=60.0, good_every=1, log=None, only=None, tame=False):
    """Every mutant of every image under every mode (a mode = ((option, value), ...)); after every `good_every`-th mutant the good image must decode to the
    expected text on the same context.  Returns a summary dict; raises AssertionError on a forbidden error code, a wrong good decode, or a call over the bound."""
    from repaq_amd import RfqError
    summary = {"mutants": 0, "errors": {}, "decoded": 0, "slowest_s": 0.0, "slowest": None, "good_checks": 0}
    for label, img, split, want in images():
        if only and label not in only:
            continue
        muts = mutants(img, seed, counts, tame)
        for mode in modes:
            for name, value in mode:
                codec.set_option(name, value)
            try:
                for k, (mlabel, mimg, index) in enumerate(muts):
                    t0 = time.perf_counter()
                    try:
                        # (every other mutant into buffers of the caller - the size the good image needs -: the emitter is then launched ahead of the host's look at the status)
                        caps = {"out_caps": ((len(want[0]) if split else len(want)) + 4096, (len(want[1]) if split else 0) + 4096)} if k % 2 else {}
                        codec.decode_bytes(mimg, split_pe=split, **caps, **({"chunk_off": index} if index else {}))
                        summary["decoded"] += 1; what = "decoded"
                    except RfqError as e:
                        # (an allocation the device cannot make is a refusal, not a fault: a few bytes of image may claim more text than the device holds)
                        oom = e.code == -2 and "out of memory" in e.message.lower()
                        assert e.code in ALLOWED or oom, "%s / %s / %s: error code %d (%s) - a HIP call failed or the device is gone" % (label, mode, mlabel, e.code, e.message)
                        what = "OOM" if oom else ALLOWED[e.code]
                        summary["errors"][what] = summary["errors"].get(what, 0) + 1
                    dt = time.perf_counter() - t0
                    if dt > summary["slowest_s"]:
                        summary["slowest_s"], summary["slowest"] = round(dt, 3), "%s/%s" % (label, mlabel)
                    assert dt < time_bound_s, "%s / %s / %s took %.1f s" % (label, mode, mlabel, dt)
                    summary["mutants"] += 1
                    if log:
                        log("%s %s %s -> %s (%.3f s)" % (label, "+".join("%s=%s" % m for m in mode) or "default", mlabel, what, dt))
                    if k % good_every == 0 or k == len(muts) - 1:
                        got = codec.decode_bytes(img, split_pe=split)
                        assert got == want, "%s / %s: the good image decodes differently after mutant %s" % (label, mode, mlabel)
                        summary["good_checks"] += 1
            finally:
                for name, _ in mode:
                    codec.set_option(name, None)
    return summary
