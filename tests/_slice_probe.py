"""Run in a subprocess with RFQ_SLICE_BYTES / RFQ_SLICE_BASES set (the library reads them once): the sliced encode / decode paths of calls
that would not fit 32-bit offsets, exercised on small inputs.  argv: library path."""
import sys

import _engine as E
import _oracle as O


def main(lib):
    from repaq_amd import RfqCodec
    c = RfqCodec(device=0, library=lib)
    n = 0
    for label, prof, units, seed, cb, paired, kw in (("se150", O.NOVA_SE150, 3000, 2, 20000, O.SE, {}), ("se_var_nonl", O.SE_VAR, 3000, 3, 15000, O.SE, dict(nonl=1)),
                                                      ("pe150", O.NOVA_PE150, 1500, 4, 20000, O.PE_TWO_FILES, dict(nonl=2)),
                                                      ("pe150_il", O.NOVA_PE150, 1500, 4, 20000, O.PE_INTERLEAVED, dict(interleaved=True)),
                                                      ("bgi", O.BGI_PE100, 1200, 5, 10000, O.PE_TWO_FILES, dict(n_quals=40))):
        fq1, fq2 = O.gen(prof, units, seed=seed, **kw)
        want = O.encode_file(fq1, fq2, paired, cb)
        got = E.encode(c, fq1, fq2, paired, cb)
        assert got == want, "%s: sliced encode differs from the oracle" % label
        # CRLF text: every slice goes through the normalising path
        crlf1, crlf2 = fq1.replace(b"\n", b"\r\n"), fq2.replace(b"\n", b"\r\n")
        assert E.encode(c, crlf1, crlf2, paired, cb) == O.encode_file(crlf1, crlf2, paired, cb), "%s: sliced CRLF encode differs" % label
        split = paired != O.SE
        back = c.decode_bytes(want, split_pe=split)
        assert back == O.decode_file(want, split), "%s: sliced decode differs from the oracle" % label
        if paired == O.PE_TWO_FILES:
            assert back == (fq1, fq2)
        # the plan pass over slices: chunk ends in the caller's coordinates
        d1 = c.dev_put(fq1); d2 = c.dev_put(fq2) if paired == O.PE_TWO_FILES else None
        r, e1, e2 = c.scan(d1, len(fq1), d2, len(fq2) if d2 else 0, paired, cb, final=True)
        offs = O.chunk_table(want)
        assert r.n_chunks == len(offs) - 1 and e1 == sorted(e1) and e1[-1] == len(fq1), label
        c.dev_free(d1)
        if d2:
            c.dev_free(d2)
        n += 1
    # mates of different length (the reference truncates to the shorter file): the short file ends inside a slice while the other goes on
    fq1, fq2 = O.gen(O.NOVA_PE150, 3000, seed=9)
    short = b"\n".join(fq2.split(b"\n")[: 4 * 300]) + b"\n"
    for a, b in ((fq1, short), (short.replace(b"/2", b"/1") if b"/2" in short else fq1[: len(short)].rsplit(b"\n@", 1)[0] + b"\n", fq2)):
        try:
            want = O.encode_file(a, b, O.PE_TWO_FILES, 20000)
        except Exception:
            continue
        assert E.encode(c, a, b, O.PE_TWO_FILES, 20000) == want, "sliced encode of mates of different length differs from the oracle"
        n += 1
    c.close()
    print("SLICES_OK", n)


if __name__ == "__main__":
    main(sys.argv[1])
