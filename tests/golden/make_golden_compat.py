#!/usr/bin/env python3
"""Golden vectors for `--bug_compat` (rfq_decode_args.bug_compat): what the REFERENCE BINARY (oracle/_ref/repaq, built from /root/reference by oracle/Makefile)
writes when it decompresses images in which chunks that are not the last carry a NO_LINE_BREAK bit - the chunk behind such a chunk is lost, and
decompressPE also drops the flagged chunk's R2 text when the R1 bit is set (src/repaq.cpp:303-325, 376-403).  Inputs are generated (tests/_oracle.gen) and
encoded with -k 100; the fixture holds sizes and md5s of the reference's own decode output.  Run in the build container: python tests/golden/make_golden_compat.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402

# (name, profile, reads, seed, how the input is paired, generator options[, decode with two outputs - default: whenever the input is paired])
CASES = [("se_nonl", O.NOVA_SE150, 6000, 1, O.SE, dict(nonl=1)), ("se_nonl_small", O.NOVA_SE150, 1500, 12, O.SE, dict(nonl=1)),
         ("pe_nonl_r1_one_output", O.NOVA_PE150, 2300, 9, O.PE_TWO_FILES, dict(nonl=1), False), ("pe_nonl_both_one_output", O.NOVA_PE150, 2300, 13, O.PE_TWO_FILES, dict(nonl=3), False),
         ("pe_nonl_r2", O.NOVA_PE150, 20000, 4, O.PE_TWO_FILES, dict(nonl=2)),
         ("pe_nonl_r1", O.NOVA_PE150, 8000, 5, O.PE_TWO_FILES, dict(nonl=1)), ("pe_nonl_both", O.NOVA_PE150, 8000, 6, O.PE_TWO_FILES, dict(nonl=3)),
         ("pe_nonl_r1_small", O.NOVA_PE150, 2300, 9, O.PE_TWO_FILES, dict(nonl=1)), ("bgi_nonl_both", O.BGI_PE100, 5000, 10, O.PE_TWO_FILES, dict(nonl=3, n_quals=40))]

if __name__ == "__main__":
    assert O.have_ref(), "oracle/_ref/repaq is needed (make -C oracle ref)"
    out = {}
    for case in CASES:
        name, prof, reads, seed, paired, kw = case[:6]
        fq1, fq2 = O.gen(prof, reads, seed=seed, **kw)
        rfq = O.ref_encode(fq1, fq2, paired, k=100)
        assert rfq == O.encode_file(fq1, fq2, paired, 100_000)
        split = case[6] if len(case) > 6 else paired != O.SE
        ref = O.ref_decode(rfq, split_pe=split)
        texts = ref if split else (ref,)
        offs = O.chunk_table(rfq)
        flagged = sum(1 for o in offs[:-2] if (rfq[o + 8] | (rfq[o + 9] << 8)) & 0x0C00)      # chunks that are not the last and carry a NO_LINE_BREAK bit
        out[name] = {"profile": prof, "reads": reads, "seed": seed, "paired": paired, "kw": kw, "split": bool(split), "rfq_md5": hashlib.md5(rfq).hexdigest(), "chunks": len(offs) - 1, "flagged_not_last": flagged,
                     "ref_decode_len": [len(t) for t in texts], "ref_decode_md5": [hashlib.md5(t).hexdigest() for t in texts],
                     "input_len": [len(fq1)] + ([len(fq2)] if split else []), "ref_roundtrip": list(texts) == ([fq1, fq2] if split else [fq1])}
        print(name, out[name]["chunks"], flagged, out[name]["ref_decode_len"], out[name]["input_len"], out[name]["ref_roundtrip"])
    json.dump(out, open(os.path.join(HERE, "compat.json"), "w"), indent=1, sort_keys=True)
