#!/usr/bin/env python3
"""A/B timing of encode (and decode) variants on ONE resident workload, inside one process: each variant = a set of environment switches the
library reads per call (RFQ_GATHER, RFQ_TUNE, RFQ_G2_KSHIFT, ...).  Prints per-variant stage times (HIP events inside the library).
usage (GPU box): python tools/ab_encode.py [--workload cfg2] [--units N] [--steps 3] "NAME:K=V,K=V" ...      (a bare NAME = no switches)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2"); ap.add_argument("--units", type=int, default=0); ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--encode-only", action="store_true"); ap.add_argument("--check", action="store_true", help="parity of every variant (rfq md5 against the golden, decode == input)"); ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    import torch
    from repaq_amd import RfqCodec
    dev = torch.device("cuda", 0); codec = RfqCodec(device=0)
    w = B.Workload(codec, dev, a.workload, a.units, None, 1000, decode=not a.encode_only)
    keys = set()
    for v in a.variants or ["base"]:
        name, _, kv = v.partition(":")
        env = dict(x.split("=", 1) for x in kv.split(",") if x)
        for k in keys: os.environ.pop(k, None)
        for k, val in env.items(): os.environ[k] = val; keys.add(k)
        w.stage = {}; w.enc_s = w.dec_s = 0.0
        par = w.check() if a.check else None
        dt = w.run(a.steps, 1, torch.cuda.synchronize, lambda: None)
        st = {k: round(val / a.steps, 3) for k, val in w.stage.items()}
        enc = sum(val for k, val in st.items() if not k.startswith("dec:")); dec = sum(val for k, val in st.items() if k.startswith("dec:"))
        print(json.dumps({"variant": name, "env": env, "ms_per_step": round(dt / a.steps * 1e3, 3), "enc_ms": round(enc, 3), "dec_ms": round(dec, 3), "stage_ms": st, "parity": par}), flush=True)
    codec.close()


if __name__ == "__main__":
    main()
