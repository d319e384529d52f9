#!/usr/bin/env python3
"""Golden maker for tests/golden/formulations.json.  Runs ONLY in the build container (needs oracle/_ref/repaq, the reference compiled in place by
oracle/Makefile): the reference binary's .rfq (md5, size) and its own decode (md5) of the inputs tests/golden/formulation_inputs.py builds.  The
JSON is data; tests/test_gpu_formulations.py forces every RFQ_* formulation of the HIP path against it."""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import _oracle as O
from formulation_inputs import INPUTS


def main():
    assert O.have_ref(), "build the reference first: make -C oracle"
    out = {}
    for name, (build, paired, k) in INPUTS.items():
        fq = build()
        rfq = O.ref_encode(fq, b"", paired, k)
        dec = O.ref_decode(rfq, False)
        assert O.encode_file(fq, b"", paired, max(100, k) * 1000) == rfq, name      # (the oracle agrees: it is the checker on the GPU box)
        out[name] = {"in_md5": hashlib.md5(fq).hexdigest(), "in_len": len(fq), "paired": paired, "k": k, "rfq_md5": hashlib.md5(rfq).hexdigest(), "rfq_len": len(rfq),
                     "decode_md5": hashlib.md5(dec).hexdigest(), "decode_len": len(dec)}
        print(name, len(fq), len(rfq), len(dec))
    json.dump(out, open(os.path.join(HERE, "formulations.json"), "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
