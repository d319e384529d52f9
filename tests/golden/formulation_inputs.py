"""Inputs of tests/test_gpu_formulations.py that fqgen has no profile for - built here, deterministically, so that the golden maker (this container,
reference binary) and the GPU test (no reference there) see the same bytes.  Data builders only; nothing of the reference is stored."""
import random


def _rec(name, seq, qual, strand="+"):
    return "%s\n%s\n%s\n%s\n" % (name, seq, strand, qual)


def crlf_mid(n=4000, seed=91):
    """'\\r\\n' line ends (src/fastqreader.cpp:94-156: the normalising path), variable read lengths, N bases, a lone '\\r' line end now and then."""
    rnd = random.Random(seed); out = []
    for i in range(n):
        ln = rnd.choice([150, 150, 151, 100, 76, 36, 250])
        seq = "".join(rnd.choice("ACGT") if rnd.random() > 0.004 else "N" for _ in range(ln))
        qual = "".join("#" if c == "N" else rnd.choice("FFFFFFFF:,") for c in seq)
        r = _rec("@A00250:26:H3YTWDSXX:1:%d:%d:%d 1:N:0:ACGTAC" % (1101 + i // 1500, 1000 + 3 * i, 2000 + i // 5), seq, qual)
        out.append(r.replace("\n", "\r" if i % 97 == 5 else "\r\n"))
    return "".join(out).encode()


def long_reads(n=260, seed=92):
    """Reads that do not fit the tile gather's staged-text buffer (two records must fit 23.5 KB: reads of 12 - 40 kB do not) among ordinary ones: the
    byte-wise gather (k_gather + k_packbytes) and, on decode, reads of more than 2000 bases (the expanded path)."""
    rnd = random.Random(seed); out = []
    for i in range(n):
        ln = rnd.choice([150, 151, 2001, 3000, 12000, 12001, 16384, 40000]) if i % 3 == 0 else rnd.randrange(30, 400)
        seq = "".join(rnd.choice("ACGT") if rnd.random() > 0.002 else "N" for _ in range(ln))
        qual = "".join("#" if c == "N" else rnd.choice("FFFFFF:,") for c in seq)
        out.append(_rec("@m64011_190830_220126/%d/ccs np=%d" % (4194370 + 17 * i, 3 + i % 11) if i % 2 else "@A00250:26:H3YTWDSXX:1:1101:%d:%d 1:N:0:ACGT" % (1000 + i, 2000 + i), seq, qual))
    return "".join(out).encode()


INPUTS = {"crlf_mid": (crlf_mid, 0, 100), "long_reads": (long_reads, 0, 100)}      # name -> (builder, paired, -k)
