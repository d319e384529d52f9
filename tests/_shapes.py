"""Inputs whose read lengths are uniform, or uniform but for ONE read, or uniform per file only: the closed-form prefixes / cuts of the encoder
(k_lens_uniform, k_fill_pq, k_partition's closed branch; round 6: the one-pass gather's optimistic cut) must hold exactly where they apply and
give way everywhere else (the coupling between "every read has L bases" and "every unit has U bases" was implied, not enforced).
Used by tests/test_emu_encode.py (small, interpreter) and tests/test_gpu_encode.py (larger, MI355X); expected images come from the oracle."""
import random

import _oracle as O


def _name(rng, i, mate):
    return b"@A00%d:%d:HXYZ%dDSXX:%d:%d:%d:%d %d:N:0:ACGTACGT" % (rng.randrange(100, 999) if i == 0 else 123, 45, 7, 1 + i // 4000 % 4, 1101 + i // 1000 % 50,
                                                                 rng.randrange(1000, 32000), rng.randrange(1000, 60000), mate)


def _rec(rng, name, n, quals=b"F:,#"):
    seq = bytes(rng.choice(b"ACGT") for _ in range(n))
    if n and rng.random() < 0.02:
        k = rng.randrange(n); seq = seq[:k] + b"N" + seq[k + 1:]
    q = bytes(quals[0] if rng.random() < 0.85 else rng.choice(quals) for _ in range(n))
    q = bytes(quals[3] if seq[i:i + 1] == b"N" else q[i] for i in range(n))
    return name + b"\n" + seq + b"\n+\n" + q + b"\n"


def fastq(lens1, lens2=None, seed=1, interleaved=False):
    """SE (lens2 None) or PE text with the given read lengths; names as a NovaSeq writes them (R1 / R2 of a pair differ in the read number only)"""
    rng = random.Random(seed)
    f1, f2 = [], []
    for i, n in enumerate(lens1):
        nm = _name(rng, i, 1)
        f1.append(_rec(rng, nm, n))
        if lens2 is not None:
            f2.append(_rec(rng, nm.replace(b" 1:N", b" 2:N"), lens2[i]))
    if lens2 is not None and interleaved:
        return b"".join(a + b for a, b in zip(f1, f2)), b""
    return b"".join(f1), b"".join(f2)


def cases(n=900, cb=20000):
    """(label, fq1, fq2, paired, chunk_bases): n reads / pairs, chunks of cb bases"""
    L = 150
    out = []
    def se(label, lens, seed):
        a, _ = fastq(lens, None, seed); out.append((label, a, b"", O.SE, cb))
    def pe(label, l1, l2, seed, inter=False):
        a, b = fastq(l1, l2, seed, inter); out.append((label, a, b, O.PE_INTERLEAVED if inter else O.PE_TWO_FILES, cb))
    se("se_uniform", [L] * n, 11)
    se("se_last_read_shorter", [L] * (n - 1) + [L - 1], 12)
    se("se_first_read_shorter", [L - 1] + [L] * (n - 1), 13)
    se("se_one_read_in_the_middle_longer", [L] * (n // 2) + [L + 1] + [L] * (n - n // 2 - 1), 14)
    se("se_second_read_differs", [L, L - 7] + [L] * (n - 2), 15)
    pe("pe_uniform", [L] * n, [L] * n, 21)
    pe("pe_r1_100_r2_50_uniform_units", [100] * n, [50] * n, 22)
    pe("pe_last_r2_shorter", [100] * n, [100] * (n - 1) + [99], 23)
    pe("pe_last_r1_shorter", [100] * (n - 1) + [99], [100] * n, 24)
    pe("pe_first_r2_longer", [100] * n, [101] + [100] * (n - 1), 25)
    pe("pe_units_uniform_reads_not", [100, 50] * (n // 2), [50, 100] * (n // 2), 26)
    pe("pe_interleaved_in_uniform", [L] * n, [L] * n, 27, True)
    pe("pe_interleaved_in_last_mate_shorter", [L] * n, [L] * (n - 1) + [L - 3], 28, True)
    return out
