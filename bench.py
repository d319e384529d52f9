#!/usr/bin/env python3
"""bench.py — raw-FASTQ MB/s of the RFQ hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
encode FASTQ -> .rfq (and, once the decode kernels are in, decode .rfq -> FASTQ of the same batch).
Workload at N=1: BASELINE.json configs[1] — synthetic NovaSeq SE150, 1 GB of FASTQ (2.8 M reads, fqgen profile 0,
seed 2), default chunk size (-k 1000).  N>1: every rank owns one such batch (chunks are independent once the header
exists; no collective on the data path) -> weak scaling; value = all ranks' FASTQ bytes / max-over-ranks time.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def cpu_baseline(fq1: bytes, reads: int):
    """Reference repaq (oracle/_ref/repaq, single thread, -O3 as its Makefile builds it) on the same workload, on this
    box's host cores; falls back to the plain-C port (oracle/liboracle.so) when the reference binary did not travel."""
    import _oracle as O
    cores = 1
    mb = len(fq1) / 1e6
    if O.have_ref():
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            p = os.path.join(d, "in.fq"); o = os.path.join(d, "out.rfq"); q = os.path.join(d, "back.fq")
            with open(p, "wb") as f:
                f.write(fq1)
            t0 = time.perf_counter(); subprocess.check_call([O.REF_BIN, "-c", "-i", p, "-o", o]); t1 = time.perf_counter()
            subprocess.check_call([O.REF_BIN, "-d", "-i", o, "-o", q]); t2 = time.perf_counter()
        enc, dec = mb / (t1 - t0), mb / (t2 - t1)
        return {"value": round(mb * 2 / (t2 - t0), 1), "unit": "MB/s", "cores": cores, "kind": "reference",
                "sample": "whole workload: %d reads / %.0f MB, repaq -c then -d, files in /tmp" % (reads, mb),
                "encode_MBps": round(enc, 1), "decode_MBps": round(dec, 1), "host_cpus": os.cpu_count()}
    t0 = time.perf_counter(); rfq = O.encode_file(fq1, b"", O.SE, 1_000_000); t1 = time.perf_counter()
    O.decode_file(rfq, False); t2 = time.perf_counter()
    return {"value": round(mb * 2 / (t2 - t0), 1), "unit": "MB/s", "cores": cores, "kind": "port",
            "sample": "whole workload: %d reads / %.0f MB, in-memory oracle encode then decode" % (reads, mb),
            "encode_MBps": round(mb / (t1 - t0), 1), "decode_MBps": round(mb / (t2 - t1), 1), "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=2_800_000, help="reads per GPU (2.8 M x 357 B = 1 GB of FASTQ)")
    ap.add_argument("--chunk-kb", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--encode-only", action="store_true")
    ap.add_argument("--pe", action="store_true", help="profiling aid: PE150 two-file workload (configs[2] shape, reads/2 pairs) instead of the SE150 headline workload")
    ap.add_argument("--bgi", action="store_true", help="profiling aid: BGI-style PE100 two-file workload (configs[4] shape: long names, 40 quality values, N runs)")
    ap.add_argument("--no-verify", action="store_true", help="skip the parity assertions (kernel ablation runs with RFQ_TUNE set)")
    args = ap.parse_args()

    import torch
    from repaq_amd import dist as D
    rank, world, local = D.env_rank()
    # test aid for 1-GPU boxes: RFQ_BENCH_SINGLE_DEVICE=1 puts every rank on GPU 0 and rendezvouses over gloo, so that the N>1 control
    # flow (barriers, max-over-ranks time, aggregate value) can be exercised where RCCL refuses two ranks on one device
    single = os.environ.get("RFQ_BENCH_SINGLE_DEVICE") == "1"
    if single:
        local = 0
    if world > 1:
        D.init("gloo" if single else "nccl", device=None if single else torch.device("cuda", local))   # "nccl" is RCCL on ROCm; barrier / max / sum only
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)" % (args.gpus, world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import _oracle as O
    from repaq_amd import RfqCodec, SE
    codec = RfqCodec(device=local)     # raises loudly without the HIP library / a GPU: there is no fallback

    seed = 2 + rank
    from repaq_amd import PE_TWO_FILES
    if args.bgi:
        args.pe = True
        fq1, fq2 = O.gen(O.BGI_PE100, args.reads // 2, seed=seed + 2, nppm=20, n_quals=40)
    elif args.pe:
        fq1, fq2 = O.gen(O.NOVA_PE150, args.reads // 2, seed=seed + 1, nppm=20)
    else:
        fq1, fq2 = O.gen(O.NOVA_SE150, args.reads, seed=seed, nppm=20)
    n = len(fq1) + len(fq2)
    d_fq = torch.frombuffer(bytearray(fq1), dtype=torch.uint8).to(dev)
    d_fq2 = torch.frombuffer(bytearray(fq2), dtype=torch.uint8).to(dev) if fq2 else None
    paired = PE_TWO_FILES if args.pe else SE
    chunk_bases = max(100, args.chunk_kb) * 1000

    have_decode = not args.encode_only
    state = {"rfq_len": 0, "chunks": 0, "stage_ms": {}, "dec_ms": 0.0, "enc_ms": 0.0}

    def step(collect):
        codec.clearHeader()
        t0 = time.perf_counter()
        r = codec.encode(d_fq.data_ptr(), len(fq1), d_fq2.data_ptr() if d_fq2 is not None else None, len(fq2), paired, chunk_bases)
        t1 = time.perf_counter()
        if collect:
            for name, ms in codec.timings():
                state["stage_ms"][name] = state["stage_ms"].get(name, 0.0) + ms
        state["rfq_len"], state["chunks"] = r.rfq_len, r.n_chunks
        t2 = t1
        if state.get("decode_ok", True) and have_decode:
            try:
                d = codec.decode(r.d_rfq, r.rfq_len, split_pe=bool(args.pe))
                t2 = time.perf_counter()
                state["decode_ok"] = True; state["dec_n"] = d.n1; state["d_fq"] = d.d_fq1
                if collect:
                    for name, ms in codec.timings():
                        state["stage_ms"]["dec:" + name] = state["stage_ms"].get("dec:" + name, 0.0) + ms
            except Exception as e:
                if "not built yet" not in str(e):
                    raise
                state["decode_ok"] = False
        if collect:
            state["enc_ms"] += (t1 - t0) * 1e3; state["dec_ms"] += (t2 - t1) * 1e3
        return r

    # parity of the measured configuration (rank 0): md5 of the .rfq against the reference's golden md5
    r = step(False)
    parity = "unchecked"
    if rank == 0 and not args.no_verify:
        got = codec.dev_get(r.d_rfq, r.rfq_len)
        md5 = hashlib.md5(got).hexdigest()
        gold = [g for g in json.load(open(os.path.join(ROOT, "tests", "golden", "generated.json")))
                if not args.pe and g["profile"] == O.NOVA_SE150 and g["reads"] == args.reads and g["seed"] == seed and g["nppm"] == 20 and g["k"] == args.chunk_kb and not g["nonl"]]
        if gold:
            assert md5 == gold[0]["rfq_md5"], "GPU .rfq md5 %s != reference golden %s" % (md5, gold[0]["rfq_md5"])
            parity = "rfq md5 == reference golden (%s)" % md5
        else:
            want = O.encode_file(fq1, fq2, paired, chunk_bases)
            assert got == want, "GPU .rfq differs from the oracle"
            parity = "rfq bytes == oracle (%s)" % md5
        if state.get("decode_ok"):
            back = codec.dev_get(state["d_fq"], state["dec_n"])
            assert back == fq1, "decode round trip differs from the input FASTQ"
            parity += "; decode == input"

    for _ in range(args.warmup):
        step(False)
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    dt, total_bytes = D.reduce_max_sum(dt, n, device=None if single else dev)      # MAX over ranks of the time, SUM of the bytes

    if rank == 0:
        K = args.steps
        passes = 2 if state.get("decode_ok") else 1          # FASTQ bytes consumed by encode + produced by decode
        value = total_bytes * passes * K / dt / 1e6
        stage = {k: v / K for k, v in state["stage_ms"].items()}
        enc_stage = {k: v for k, v in stage.items() if not k.startswith("dec:")}
        dec_stage = {k: v for k, v in stage.items() if k.startswith("dec:")}
        dom = max(stage, key=stage.get) if stage else None    # the longest stage of either direction (HIP events on the codec's stream)
        alg = float(n + state["rfq_len"])                    # B_fastq + B_rfq per batch, either direction
        roof = None
        if dom:
            # HBM bytes of the dominant stage from the committed PMC passes (tools/pmc_summary.py: FETCH_SIZE and WRITE_SIZE collected in
            # separate rocprofv3 --pmc runs of this same command, KB units, FETCH_SIZE x2 on gfx950) — only valid for the default workload
            traffic = None
            STAGE_KERNELS = {"index": ["k_nl_bitmap", "k_line_offsets", "k_line_tail"], "read_table+cut": ["k_read_table", "k_unit_len", "k_partition"],
                             "chunk_flags+overlap": ["k_chunk_flags_se", "k_chunk_flags_pe", "k_chunk_flags_a", "k_chunk_flags_b", "k_overlap", "k_pv_in", "k_scan_reduce<U4>", "k_scan_apply<U4>", "k_chunk_bases"],
                             "gather": ["k_gather", "k_stream_plan", "k_chunk_layout"], "pos_coder": ["k_pos_coder<0>", "k_pos_coder<1>", "k_pos_coder<2>"],
                             "coords+layout": ["k_coords"], "assemble": ["k_assemble", "k_assemble_names"], "header": ["k_hdr_stats", "k_hdr_pass2"],
                             "dec:walk": ["k_dec_spec_walk", "k_dec_parse"], "dec:read_table": ["k_dec_readtab"],
                             "dec:streams": ["k_dec_bases", "k_dec_fill", "k_dec_unpack", "k_dec_coords", "k_dec_pos_sum", "k_dec_pos_link", "k_dec_pos_emit", "k_dec_except"],
                             "dec:textlen": ["k_dec_textlen"], "dec:emit": ["k_dec_emit"]}
            pj = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            if args.reads == 2_800_000 and args.chunk_kb == 1000 and not args.pe and os.path.exists(pj):
                pmc = json.load(open(pj))
                ks = [k for k in STAGE_KERNELS.get(dom, []) if k in pmc]
                traffic = int(sum(pmc[k]["fetch_bytes"] + pmc[k]["write_bytes"] for k in ks)) if ks else None
            ach = alg / (stage[dom] * 1e-3) / 1e9
            enc_ms, dec_ms = sum(enc_stage.values()), sum(dec_stage.values())
            roof = {"bound": "hbm", "kernel": "+".join(STAGE_KERNELS.get(dom, [dom])), "stage": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": round(stage[dom], 4),
                    "whole_encode_frac": round(alg / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if enc_ms else None,
                    "whole_decode_frac": round(alg / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dec_ms else None}
        out = {
            "metric": "raw FASTQ MB/s encode+decode" if state.get("decode_ok") else "raw FASTQ MB/s encode (decode pending)",
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": (("configs[4] shape (profiling aid): synthetic BGI-style PE100, 40 quality values, %.2f GB FASTQ per GPU (fqgen profile 3, %d pairs), -k %d" if args.bgi else
                                     "configs[2] shape (profiling aid): synthetic NovaSeq PE150 two files, %.2f GB FASTQ per GPU (fqgen profile 1, %d pairs), -k %d")
                                    % (n / 1e9, args.reads // 2, args.chunk_kb)) if args.pe else
                                   "configs[1]: synthetic NovaSeq SE150 %.2f GB FASTQ per GPU (fqgen profile 0, %d reads, seed 2+rank, N 20 ppm), -k %d"
                                   % (n / 1e9, args.reads, args.chunk_kb), "chunks_per_gpu": state["chunks"], "rfq_over_fastq": round(state["rfq_len"] / n, 4),
                       "encode_MBps_per_gpu": round(n * K / (state["enc_ms"] * 1e-3) / 1e6, 1) if state["enc_ms"] else None,
                       "decode_MBps_per_gpu": round(n * K / (state["dec_ms"] * 1e-3) / 1e6, 1) if state.get("decode_ok") and state["dec_ms"] else None,
                       "parity": parity, "stage_ms": {k: round(v, 3) for k, v in stage.items()}},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline and not args.pe:
            out["cpu_baseline"] = cpu_baseline(fq1, args.reads)
        print(json.dumps(out))
    codec.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
