"""Seeded random FASTQ inputs for differential tests of the HIP path against the oracle: shapes the hand-made goldens and the generated
configs do not sweep systematically — read lengths around the 16-byte group sizes of the LDS copy loops (1, 15, 16, 17, 31, 33 ...),
names that parse / do not parse / are longer than the staged rows, strand lines with text, 1-41 quality values with exceptions,
N-rich reads, pairs that overlap by a random amount (forward or reverse), reads of random length, several chunks or one."""
import random

import _oracle as O

COMP = {65: 84, 84: 65, 67: 71, 71: 67, 78: 78}


def _rc(s: bytes) -> bytes:
    return bytes(COMP[b] for b in reversed(s))


def case(seed: int):
    """-> (fq1, fq2, paired, chunk_bases)"""
    r = random.Random(seed)
    paired = r.choice([O.SE, O.SE, O.PE_TWO_FILES, O.PE_INTERLEAVED])
    fixed = r.random() < 0.6
    L = r.choice([1, 2, 7, 15, 16, 17, 31, 32, 33, 48, 100, 151, 151, 250, 300])
    big = r.random() < 0.25                                      # enough bases for several chunks at -k 100
    n = r.randint(700, 1500) * max(1, 160 // max(L, 40)) if big else r.randint(1, 120)
    quals = bytes(r.sample(range(34, 76), r.choice([1, 2, 4, 4, 8, 41 if r.random() < 0.3 else 6])))
    qw = [r.random() ** 3 + 0.01 for _ in quals]
    exc = r.random() < 0.2
    nprob = r.choice([0.0, 0.0, 0.002, 0.15])
    style = r.choice(["illumina", "illumina", "illumina_nocomment", "free", "same", "long"])
    strand_text = r.random() < 0.2
    lane = r.randint(1, 8); tile = r.randint(1101, 2678); x = r.randint(1000, 30000); y = r.randint(1000, 200000)
    recs1, recs2 = [], []
    for i in range(n):
        ln = L if fixed else r.randint(1, L)
        seq = bytearray(r.choice(b"ACGT") for _ in range(ln))
        for k in range(ln):
            if r.random() < nprob:
                seq[k] = 78
        q = bytearray(r.choices(quals, qw, k=ln))
        if exc and r.random() < 0.3:
            q[r.randrange(ln)] = r.randrange(33, 80)
        if r.random() < 0.6:
            x += r.randint(0, 40)
        else:
            x = r.randint(1000, 30000); y += r.randint(1, 300)
        if r.random() < 0.02:
            tile += 1; y = r.randint(1000, 3000)
        if style == "free":
            name = b"@r%d.%s" % (i, bytes(r.choice(b"abcXYZ_-") for _ in range(r.randint(0, 20))))
        elif style == "same":
            name = b"@same"
        else:
            name = b"@M0%d:%d:FC%s:%d:%d:%d:%d" % (r.randint(1, 3), 26, b"H3YTW" * (8 if style == "long" else 1), lane, tile, x, y)
        c1 = c2 = b""
        if style in ("illumina", "long"):
            idx = bytes(r.choice(b"ACGT") for _ in range(r.choice([0, 6, 8])))
            c1 = b" 1:N:0:" + idx; c2 = b" 2:N:0:" + idx
        st = b"+" + (name[1:] if strand_text else b"")
        recs1.append(name + c1 + b"\n" + bytes(seq) + b"\n" + st + b"\n" + bytes(q) + b"\n")
        if paired != O.SE:
            # the mate: reverse complement of a window that overlaps R1's tail (or head) by a random amount, or an unrelated read
            ln2 = L if fixed else r.randint(1, L)
            mode = r.random()
            if mode < 0.5 and ln >= 12:
                ov = r.randint(min(12, ln), ln); tail = bytes(seq[ln - ov:]) + bytes(r.choice(b"ACGT") for _ in range(max(0, ln2 - ov)))
                s2 = _rc(tail[:ln2]) if len(tail) >= ln2 else _rc(tail + bytes(r.choice(b"ACGT") for _ in range(ln2 - len(tail))))
            elif mode < 0.65 and ln >= 12:
                ov = r.randint(min(12, ln), ln); head = bytes(r.choice(b"ACGT") for _ in range(max(0, ln2 - ov))) + bytes(seq[:ov])
                s2 = _rc(head[-ln2:]) if len(head) >= ln2 else _rc(bytes(r.choice(b"ACGT") for _ in range(ln2 - len(head))) + head)
            else:
                s2 = bytes(r.choice(b"ACGT") for _ in range(ln2))
            s2 = bytes(s2[:ln2]); q2 = bytes(r.choices(quals, qw, k=len(s2)))
            n2 = name if r.random() > 0.03 else name + b"x"
            recs2.append(n2 + c2 + b"\n" + s2 + b"\n" + st + b"\n" + q2 + b"\n")
    if paired == O.PE_INTERLEAVED:
        fq1 = b"".join(a + b for a, b in zip(recs1, recs2)); fq2 = b""
    else:
        fq1 = b"".join(recs1); fq2 = b"".join(recs2)
    if r.random() < 0.3 and fq1:
        fq1 = fq1[:-1]                                           # no final line break
    if r.random() < 0.2 and fq2:
        fq2 = fq2[:-1]
    # reader quirks (src/fastqreader.cpp:94-196): "\r\n" line ends; one blank line after a record (swallowed); two (reading stops there)
    t = r.random()
    if t < 0.10:
        fq1 = fq1.replace(b"\n", b"\r\n"); fq2 = fq2.replace(b"\n", b"\r\n")
    elif t < 0.16 and recs1:
        k = r.randrange(len(recs1)); cut = sum(len(x) for x in recs1[:k + 1]) if paired != O.PE_INTERLEAVED else None
        if cut is not None and cut <= len(fq1):
            fq1 = fq1[:cut] + (b"\n" if r.random() < 0.5 else b"\n\n") + fq1[cut:]
    return fq1, fq2, paired, 100_000


def check(codec, encode, seed):
    """encode == oracle (or the same error text), decode(oracle image) == oracle decode."""
    from repaq_amd import RfqError
    fq1, fq2, paired, cb = case(seed)
    try:
        want = O.encode_file(fq1, fq2, paired, cb)
    except O.OracleError as e:
        try:
            encode(codec, fq1, fq2, paired, cb)
        except RfqError as g:
            # (the one input class both sides REFUSE rather than reproduce - the reference overflows its heap there, App. C Q6 - is worded differently)
            unpinned = "1.5x scratch buffer" in g.message and "1.5x scratch buffer" in str(e)
            assert unpinned or g.message.strip() == str(e).strip(), (seed, g.message, str(e))
            return "error"
        raise AssertionError("seed %d: the oracle refuses this input (%s), the engine encoded it" % (seed, e))
    got = encode(codec, fq1, fq2, paired, cb)
    assert got == want, "seed %d: image differs (%d vs %d bytes)" % (seed, len(got), len(want))
    if want:
        split = paired != O.SE
        assert codec.decode_bytes(want, split_pe=split) == O.decode_file(want, split), "seed %d: decode differs" % seed
        if split:
            assert codec.decode_bytes(want, split_pe=False) == O.decode_file(want, False), "seed %d: interleaved decode differs" % seed
    return "ok"


MiB = 1 << 20


def block_case(seed: int):
    """-> (fq1, fq2, paired, chunk_bases): inputs of one to three reader blocks (the reference reads 1 MiB at a time, src/fastqreader.cpp:31-46)
    whose ENDS and block edges are what varies - sizes of exactly k MiB and a byte or two off, a final line break or none, a last record cut
    off at a random byte, "\r\n" and blank lines laid across a block edge, mate files of different length.  The line-break bits of the chunks,
    where reading stops and what the tail chunk holds all hang on these."""
    r = random.Random(7000 + seed)
    paired = r.choice([O.SE, O.SE, O.PE_TWO_FILES, O.PE_TWO_FILES, O.PE_INTERLEAVED])
    prof = r.choice([O.NOVA_SE150, O.SE_VAR]) if paired == O.SE else O.NOVA_PE150
    blocks = r.choice([1, 1, 2, 3])
    per = 360 if prof != O.SE_VAR else 250
    reads = (blocks * MiB) // per + 600
    a, b = O.gen(prof, reads if paired != O.PE_INTERLEAVED else reads // 2, seed=900 + seed, nppm=r.choice([0, 20, 3000]), interleaved=paired == O.PE_INTERLEAVED)

    def shape(t: bytes) -> bytes:
        # the size: exactly k MiB, a byte or two beside it, or anywhere; reached by cutting (a cut lands where it lands: mid-name, mid-sequence,
        # on a line break ...) or, for a clean end, by cutting at a record boundary
        target = blocks * MiB + r.choice([0, 0, 0, -2, -1, 1, 2, 17, r.randint(-5000, 5000), r.randint(3, MiB // 2)])
        target = max(400, min(target, len(t)))
        how = r.random()
        if how < 0.35:
            t = t[:target]                                            # cut anywhere
        elif how < 0.7:
            e = t.rfind(b"\n@", 0, target) + 1                        # whole records ...
            t = t[:e]
            if r.random() < 0.5:
                t = t[:-1]                                            # ... without the final line break
            if r.random() < 0.5 and len(t) < target and target - len(t) < 150:
                k = t.index(b"\n")                                    # ... padded (inside the first name) to the target size
                t = t[:k] + b"p" * (target - len(t)) + t[k:]
        else:
            e = t.rfind(b"\n@", 0, target) + 1
            j = t.find(b"\n", e) + 1                                  # a record's name line kept, its sequence line cut short
            t = t[:min(len(t), j + r.randint(0, 60))]
        # line-end quirks across a block edge
        q = r.random()
        if q < 0.2 and len(t) > MiB:
            edge = MiB * r.randint(1, max(1, len(t) // MiB))
            lo = t.rfind(b"\n", 0, min(edge, len(t) - 1)); hi = t.find(b"\n", min(edge, len(t) - 1))
            if lo > 0 and hi > 0:
                seg = t[lo - 400 if lo > 400 else 0:hi + 400].replace(b"\n", b"\r\n")
                t = t[:lo - 400 if lo > 400 else 0] + seg + t[hi + 400:]
        elif q < 0.3 and len(t) > MiB:
            edge = MiB * r.randint(1, max(1, len(t) // MiB))
            lo = t.rfind(b"\n@", 0, min(edge, len(t) - 1))
            if lo > 0:
                t = t[:lo + 1] + b"\n" * r.choice([1, 1, 2]) + t[lo + 1:]
        return t
    if paired == O.PE_TWO_FILES:
        fq1, fq2 = shape(a), shape(b)
        if r.random() < 0.4:                                          # one mate file much shorter
            cut = r.randint(len(fq2) // 3, len(fq2))
            if r.random() < 0.5:
                fq2 = fq2[:cut]
            else:
                fq1 = fq1[:min(len(fq1), cut)]
    else:
        fq1, fq2 = shape(a), b""
    return fq1, fq2, paired, r.choice([100_000, 100_000, 250_000])


def check_block(codec, encode, seed):
    """block_case(seed): encode == oracle (or the same refusal), decode(oracle image) == oracle decode."""
    from repaq_amd import RfqError
    fq1, fq2, paired, cb = block_case(seed)
    try:
        want = O.encode_file(fq1, fq2, paired, cb)
    except O.OracleError as e:
        try:
            encode(codec, fq1, fq2, paired, cb)
        except RfqError as g:
            same_refusal = any(k in g.message and k in str(e) for k in ("1.5x scratch buffer", "quality line", "shorter than"))
            assert same_refusal or g.message.strip() == str(e).strip(), (seed, g.message, str(e))
            return "error"
        raise AssertionError("block seed %d: the oracle refuses this input (%s), the engine encoded it" % (seed, e))
    got = encode(codec, fq1, fq2, paired, cb)
    assert got == want, "block seed %d: image differs (%d vs %d bytes, first difference at %d)" % (
        seed, len(got), len(want), next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1))
    if want:
        split = paired != O.SE
        assert codec.decode_bytes(want, split_pe=split) == O.decode_file(want, split), "block seed %d: decode differs" % seed
    return "ok"


def long_case(seed: int):
    """-> (fq1, fq2, paired, chunk_bases): reads of 300 .. 30000 bases beside short ones - rows longer than the overlap search packs, records
    longer than a gather / emit tile (the byte-wise paths), the decoder's materialising path (reads > 2000 bases)."""
    r = random.Random(5000 + seed)
    paired = r.choice([O.SE, O.PE_TWO_FILES, O.PE_INTERLEAVED])
    n = r.randint(3, 60)
    quals = bytes(r.sample(range(35, 75), r.choice([2, 4, 6])))
    recs1, recs2 = [], []
    for i in range(n):
        L = r.choice([300, 700, 1999, 2000, 2001, 2500, 6000, 7000, 14000, 30000]) if r.random() < 0.5 else r.randint(1, 400)
        seq = bytes(r.choice(b"ACGTN" if r.random() < 0.05 else b"ACGT") for _ in range(L)); q = bytes(r.choices(quals, k=L))
        name = b"@M01:26:FCX:1:%d:%d:%d 1:N:0:AC" % (1101 + i // 7, 1000 + i * 3, 2000 + i)
        recs1.append(name + b"\n" + seq + b"\n+\n" + q + b"\n")
        if paired != O.SE:
            L2 = r.choice([L, r.randint(1, max(1, L))])
            if r.random() < 0.5 and min(L, L2) >= 12:
                ov = r.randint(12, min(L, L2))
                tail = seq[L - ov:] + bytes(r.choice(b"ACGT") for _ in range(max(0, L2 - ov)))
                s2 = _rc(bytes(b if b in COMP else 65 for b in tail[:L2]))
            else:
                s2 = bytes(r.choice(b"ACGT") for _ in range(L2))
            recs2.append(name.replace(b" 1:", b" 2:") + b"\n" + s2 + b"\n+\n" + bytes(r.choices(quals, k=len(s2))) + b"\n")
    if paired == O.PE_INTERLEAVED:
        return b"".join(a + b for a, b in zip(recs1, recs2)), b"", paired, 100_000
    return b"".join(recs1), b"".join(recs2), paired, 100_000


def qual_case(seed: int):
    """-> (fq1, fq2, paired, chunk_bases): 1 .. 93 distinct quality values with flat to very skewed weights (the header's table rules,
    src/rfqheader.cpp:130-237: up to 64 streams, raw qualities beyond), values that first appear after chunk 0 (exception records), N bases
    whose quality is or is not the N quality."""
    r = random.Random(9000 + seed)
    paired = r.choice([O.SE, O.SE, O.PE_TWO_FILES, O.PE_INTERLEAVED])
    n = r.choice([5, 40, 400, 1500])
    nq = r.choice([1, 2, 3, 12, 40, 62, 63, 64, 65, 66, 80, 93])
    quals = bytes(r.sample(range(33, 127), nq)); w = [r.random() ** r.choice([1, 3, 6]) + 1e-3 for _ in quals]
    L = r.choice([36, 75, 100, 151])
    late = r.random() < 0.4
    recs1, recs2 = [], []
    for i in range(n):
        pool, pw = (quals[:max(1, nq // 3)], w[:max(1, nq // 3)]) if (late and i < n // 2) else (quals, w)
        seq = bytes(r.choice(b"ACGT") for _ in range(L)); q = bytearray(r.choices(pool, pw, k=L))
        if r.random() < 0.1:
            k = r.randrange(L); seq = seq[:k] + b"N" + seq[k + 1:]
            if r.random() < 0.5:
                q[k] = r.choice(quals)
        name = b"@M01:26:FCX:1:%d:%d:%d 1:N:0:AC" % (1101 + i // 70, 1000 + i * 3, 2000 + i)
        recs1.append(name + b"\n" + seq + b"\n+\n" + bytes(q) + b"\n")
        if paired != O.SE:
            s2 = bytes(r.choice(b"ACGT") for _ in range(L))
            recs2.append(name.replace(b" 1:", b" 2:") + b"\n" + s2 + b"\n+\n" + bytes(r.choices(pool, pw, k=L)) + b"\n")
    if paired == O.PE_INTERLEAVED:
        return b"".join(a + b for a, b in zip(recs1, recs2)), b"", paired, 100_000
    return b"".join(recs1), b"".join(recs2), paired, 100_000


def overlap_case(seed: int):
    """-> (fq1, fq2, paired, chunk_bases): pairs made to stress the overlap search (src/rfqcodec.cpp:1391-1438): low-complexity reads (homopolymers,
    di- / tri-nucleotide repeats, a short motif with a few point changes) whose 8-base heads recur all along the mate - many candidates pass the
    filter and fail in full, or succeed at the SMALLEST o only -, planted overlaps of every length from 12 up with 0 .. 2 mismatches (one mismatch
    kills a candidate), N inside or outside the overlap, lengths around the 32-base pack groups and the 256-base rows; one file in eight holds a
    lower-case or other character (a pair that leaves the 2-bit rows; the reference then refuses the file: both sides must)."""
    r = random.Random(13000 + seed)
    paired = r.choice([O.PE_TWO_FILES, O.PE_INTERLEAVED])
    n = r.randint(20, 160)
    quals = bytes(r.sample(range(35, 75), 4))
    lens = [12, 13, 20, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 150, 151, 200, 250, 255, 256, 257, 300]
    odd = r.random() < 0.12; odd_at = r.randrange(n)

    def lowc(L):
        t = r.random()
        if t < 0.25:
            return bytes([r.choice(b"ACGT")]) * L
        if t < 0.5:
            m = bytes(r.choice(b"ACGT") for _ in range(r.choice([2, 3, 4, 5, 7])))
            return (m * (L // len(m) + 1))[:L]
        if t < 0.75:
            m = bytes(r.choice(b"ACGT") for _ in range(r.choice([8, 9, 12, 16, 24])))
            b = bytearray((m * (L // len(m) + 1))[:L])
            for _ in range(r.randint(0, 3)):
                b[r.randrange(L)] = r.choice(b"ACGT")
            return bytes(b)
        return bytes(r.choice(b"ACGT") for _ in range(L))

    recs1, recs2 = [], []
    for i in range(n):
        L1 = r.choice(lens) if r.random() < 0.7 else r.randint(1, 300)
        s1 = bytearray(lowc(L1))
        L2 = L1 if r.random() < 0.5 else (r.choice(lens) if r.random() < 0.7 else r.randint(1, 300))
        mode = r.random(); m = min(L1, L2)
        if mode < 0.45 and m >= 12:                               # R1 tail == RC(R2) head
            ov = r.choice([12, 13, m, m - 1 if m > 12 else 12, r.randint(12, m)])
            rc2 = bytearray(bytes(s1[L1 - ov:]) + lowc(max(1, L2 - ov)))[:L2]
        elif mode < 0.7 and m >= 12:                              # RC(R2) tail == R1 head
            ov = r.choice([12, m, r.randint(12, m)])
            rc2 = bytearray(lowc(max(1, L2 - ov)) + bytes(s1[:ov]))[-L2:]
        elif mode < 0.85:
            rc2 = bytearray(s1[:L2] if L2 <= L1 else bytes(s1) + lowc(L2 - L1))   # the same low-complexity text: candidates everywhere
        else:
            rc2 = bytearray(lowc(L2))
        if len(rc2) < L2:
            rc2 = rc2 + bytearray(lowc(L2 - len(rc2)))
        for _ in range(r.choice([0, 0, 0, 1, 2])):                 # point changes after the overlap was planted
            b = s1 if r.random() < 0.5 else rc2
            b[r.randrange(len(b))] = r.choice(b"ACGT")
        t = r.random()
        if t < 0.15:
            b = s1 if r.random() < 0.5 else rc2; b[r.randrange(len(b))] = 78                 # N
        elif odd and i == odd_at:                                                              # (the reference refuses such a file as a whole - after the search has met the pair)
            b = s1 if r.random() < 0.5 else rc2; b[r.randrange(len(b))] = r.choice(b"acgtnRYKM.")
        s2 = bytes(COMP.get(c, c) if c in COMP else c for c in reversed(bytes(rc2)))
        name = b"@M01:26:FCX:1:%d:%d:%d 1:N:0:AC" % (1101 + i // 50, 1000 + i * 3, 2000 + i)
        recs1.append(name + b"\n" + bytes(s1) + b"\n+\n" + bytes(r.choices(quals, k=len(s1))) + b"\n")
        recs2.append(name.replace(b" 1:", b" 2:") + b"\n" + s2 + b"\n+\n" + bytes(r.choices(quals, k=len(s2))) + b"\n")
    if paired == O.PE_INTERLEAVED:
        return b"".join(a + b for a, b in zip(recs1, recs2)), b"", paired, 100_000
    return b"".join(recs1), b"".join(recs2), paired, 100_000


def check_gen(codec, encode, gen, seed):
    """gen(seed): encode == oracle (or both refuse), decode(oracle image) == oracle decode."""
    from repaq_amd import RfqError
    fq1, fq2, paired, cb = gen(seed)
    try:
        want = O.encode_file(fq1, fq2, paired, cb)
    except O.OracleError as e:
        try:
            encode(codec, fq1, fq2, paired, cb)
        except RfqError:
            return "error"
        raise AssertionError("%s seed %d: the oracle refuses this input (%s), the engine encoded it" % (gen.__name__, seed, e))
    got = encode(codec, fq1, fq2, paired, cb)
    assert got == want, "%s seed %d: image differs" % (gen.__name__, seed)
    split = paired != O.SE
    assert codec.decode_bytes(want, split_pe=split) == O.decode_file(want, split), "%s seed %d: decode differs" % (gen.__name__, seed)
    return "ok"
