#!/usr/bin/env python3
"""bench.py — raw-FASTQ MB/s of the RFQ hot path on MI355X (BASELINE.json metric: encode + decode, bit-exact .rfq vs the reference).

A step = one pass of the hot path over one batch of synthetic input that is already resident in HBM: FASTQ -> .rfq
(rfq_encode_batch) and .rfq -> FASTQ (rfq_decode_batch) of the same batch, both results left in HBM.

N = 1 (the driver's headline run): BASELINE.json configs[2] — synthetic NovaSeq PE150, two files of 4.0 GB (`-i/-I`, fqgen profile 1,
11.2 M pairs, seed 3, -k 1000), one batch.  The .rfq's md5 is checked against the reference's own output (tests/golden/big.json,
made by tests/golden/make_golden_big.py with the compiled reference), the decoded mates against the inputs, inside the run.
Secondary lines (same JSON object, "secondary"): configs[1] (SE150 1 GB, md5 vs the reference golden) and the configs[4] shape
(BGI-style PE100, 40 quality values).

N > 1: ONE configs[3]-shaped PE150 input of N x (2 x 8 GB) encoded chunk-parallel by the N GPUs: see run_multi() (static resident shares + a plan step);
--queue: the same input through a host work queue - rank-agnostic batches pulled from a shared counter - see run_queue() (also with --gpus 1: the whole input on one GPU).

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

# stage (HIP-event pair inside librfq_hip) -> the kernels it brackets (names as rocprofv3 reports them)
STAGE_KERNELS = {"index": ["k_line_index", "k_line_tail"], "index_2pass": ["k_nl_bitmap", "k_line_offsets", "k_line_tail"], "lens+cut": ["k_read_lens", "k_lens_uniform", "k_fill_pq", "k_partition"],
                 "chunk_flags": ["k_chunk_flags_a", "k_chunk_flags_b", "k_chunk_bases"],
                 "gather": ["k_gather2", "k_mask_bounds", "k_chunk_flags_b", "k_stream_plan"], "gather_bytes": ["k_overlap", "k_overlap_apply", "k_pv_in", "k_gather", "k_packbytes", "k_stream_plan", "k_chunk_layout"],
                 # (tile gather: the overlap search on the loose slots, the stored prefix, the sequence packer and the N streams run on the second stream beside the coder)
                 "pos_coder": ["k_pos_coder", "k_pos_coder_list", "k_overlap", "k_chunk_prefix", "k_seqpack", "k_chunk_layout", "k_coords", "k_rare_cleanup"],
                 "coords+layout": ["k_pos_sizes", "k_chunk_layout"], "assemble": ["k_assemble", "k_assemble_names"], "header": ["k_read_table", "k_hdr_stats", "k_hdr_pass2", "k_dense_order"],
                 "dec:walk": ["k_dec_table", "k_dec_rebase", "k_dec_gw_find", "k_dec_gw_walk", "k_dec_gw_stitch", "k_dec_parse", "k_dec_summary"], "dec:read_table": ["k_dec_readtab2", "k_dec_readtab"],
                 "dec:streams": ["k_dec_coords", "k_dec_pos_sum2", "k_dec_pos_link2", "k_dec_pos_off", "k_dec_pos_list", "k_dec_textlen2", "k_dec_textlen",   # (fused path: the text lengths run beside the list chain)
                                 # (reads longer than 2000 bases, -k values whose chunks exceed 4096 records, and legacy RLE files take the materialising path)
                                 "k_dec_bases", "k_dec_fill", "k_dec_unpack", "k_dec_pos_sum", "k_dec_pos_link", "k_dec_pos_emit", "k_dec_except", "k_dec_rle"],
                 "dec:textlen": ["k_dec_textlen"], "dec:emit": ["k_dec_emit3"], "dec:emit_expanded": ["k_dec_emit"]}

# stages that are ONE kernel (the roofline object is about a kernel: the longest of these; "pos_coder" is a phase of two streams - the coder beside the
# overlap / prefix / sequence-packer chain - and is reported as `longest_stage`)
KERNEL_STAGES = ("dec:emit", "dec:emit_expanded", "gather")


def dominant_kernel_stage(stage):
    ks = [k for k in stage if k in KERNEL_STAGES]
    return max(ks, key=stage.get) if ks else max(stage, key=stage.get)


WORKLOADS = {
    # key: (label, fqgen profile, units (reads or pairs), seed, extra gen kwargs, paired)
    "cfg2": ("configs[2]: synthetic NovaSeq PE150 two files (-i/-I), 2 x %.2f GB FASTQ (fqgen profile 1, %d pairs, seed %d, N 20 ppm), -k %d, encode + decode", 1, 11_200_000, 3, {}),
    "cfg1": ("configs[1]: synthetic NovaSeq SE150 %.2f GB FASTQ (fqgen profile 0, %d reads, seed %d, N 20 ppm), -k %d, encode + decode", 0, 2_800_000, 2, {}),
    "cfg4": ("configs[4] shape: synthetic BGI-style PE100 two files, long names, 40 quality values, N runs, 2 x %.2f GB FASTQ (fqgen profile 3, %d pairs, seed %d), -k %d, encode + decode", 3, 1_400_000, 6, {"nppm": 0, "n_quals": 40}),
}


def offset_of_record(arr, rec):
    """Byte offset of record `rec` in a '\\n'-terminated FASTQ held in a numpy array (4 lines per record)."""
    import numpy as np
    want = 4 * rec
    if want == 0:
        return 0
    pos, seen, slab = 0, 0, 1 << 26
    while pos < arr.size:
        nl = np.flatnonzero(arr[pos:pos + slab] == 10)
        if seen + nl.size >= want:
            return pos + int(nl[want - seen - 1]) + 1
        seen += nl.size; pos += slab
    return int(arr.size)


def cpu_baseline(a1, a2, paired, units, sample_units):
    """Reference repaq (oracle/_ref/repaq: the reference's own sources compiled by oracle/Makefile, single thread, -O3 as its Makefile
    builds it) on a BOUNDED SAMPLE of the same workload - its first `sample_units` reads / pairs - on this box's host cores; falls back
    to the plain-C port (oracle/liboracle.so) when the reference binary did not travel."""
    import _oracle as O
    su = min(units, sample_units)
    c1 = offset_of_record(a1, su); c2 = offset_of_record(a2, su) if paired else 0
    mb = (c1 + c2) / 1e6
    what = "first %d %s of the workload (%.0f MB of FASTQ)" % (su, "pairs" if paired else "reads", mb)
    if O.have_ref():
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            p1, p2, o, q1, q2 = (os.path.join(d, n) for n in ("r1.fq", "r2.fq", "o.rfq", "b1.fq", "b2.fq"))
            a1[:c1].tofile(p1)
            if paired:
                a2[:c2].tofile(p2)
            enc = [O.REF_BIN, "-c", "-i", p1, "-o", o] + (["-I", p2] if paired else [])
            dec = [O.REF_BIN, "-d", "-i", o, "-o", q1] + (["-O", q2] if paired else [])
            t0 = time.perf_counter(); subprocess.check_call(enc); t1 = time.perf_counter()
            subprocess.check_call(dec); t2 = time.perf_counter()
        return {"value": round(mb * 2 / (t2 - t0), 1), "unit": "MB/s", "cores": 1, "kind": "reference",
                "sample": what + ": repaq -c then -d, one thread, files in /tmp",
                "encode_MBps": round(mb / (t1 - t0), 1), "decode_MBps": round(mb / (t2 - t1), 1), "host_cpus": os.cpu_count()}
    f1 = a1[:c1].tobytes(); f2 = a2[:c2].tobytes() if paired else b""
    t0 = time.perf_counter(); rfq = O.encode_file(f1, f2, O.PE_TWO_FILES if paired else O.SE, 1_000_000); t1 = time.perf_counter()
    O.decode_file(rfq, bool(paired)); t2 = time.perf_counter()
    return {"value": round(mb * 2 / (t2 - t0), 1), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": what + ": in-memory oracle encode then decode, one thread",
            "encode_MBps": round(mb / (t1 - t0), 1), "decode_MBps": round(mb / (t2 - t1), 1), "host_cpus": os.cpu_count()}


def cpu_baseline_all_cores(a1, a2, paired, units, shard_units, max_procs=32):
    """The same reference binary on min(32, host cores) processes at once, each on its own shard of the workload (`shard_units` reads / pairs,
    consecutive shards from the start of the input): what the host's cores deliver together - chunks are independent, so a chunk-parallel CPU
    run would look like this.  A stated baseline, never the target."""
    import _oracle as O
    if not O.have_ref():
        return None
    procs = max(1, min(max_procs, os.cpu_count() or 1, units // max(1, shard_units)))
    cuts1 = [offset_of_record(a1, i * shard_units) for i in range(procs + 1)]
    cuts2 = [offset_of_record(a2, i * shard_units) for i in range(procs + 1)] if paired else None
    mb = (cuts1[-1] + (cuts2[-1] if paired else 0)) / 1e6
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        enc, dec = [], []
        for i in range(procs):
            p1, p2, o, q1, q2 = (os.path.join(d, "%s%d" % (n, i)) for n in ("r1_", "r2_", "o_", "b1_", "b2_"))
            a1[cuts1[i]:cuts1[i + 1]].tofile(p1)
            if paired:
                a2[cuts2[i]:cuts2[i + 1]].tofile(p2)
            enc.append([O.REF_BIN, "-c", "-i", p1, "-o", o] + (["-I", p2] if paired else []))
            dec.append([O.REF_BIN, "-d", "-i", o, "-o", q1] + (["-O", q2] if paired else []))
        t0 = time.perf_counter()
        for p in [subprocess.Popen(c) for c in enc]:
            assert p.wait() == 0
        t1 = time.perf_counter()
        for p in [subprocess.Popen(c) for c in dec]:
            assert p.wait() == 0
        t2 = time.perf_counter()
    return {"value": round(mb * 2 / (t2 - t0), 1), "unit": "MB/s", "cores": procs, "kind": "reference",
            "sample": "%d concurrent repaq processes, each -c then -d on its own %d %s of the workload (%.0f MB of FASTQ in all), files in /tmp" % (procs, shard_units, "pairs" if paired else "reads", mb),
            "encode_MBps": round(mb / (t1 - t0), 1), "decode_MBps": round(mb / (t2 - t1), 1), "host_cpus": os.cpu_count()}


def golden_md5(key, units, seed, k):
    """The reference's .rfq md5 for this exact input, if a golden was made for it (tests/golden/*.json), else None."""
    try:
        if key == "cfg2":
            g = json.load(open(os.path.join(ROOT, "tests", "golden", "big.json"))).get("cfg2")
            if g and g["pairs"] == units and g["seed"] == seed and k == g["k"]:
                return g["rfq_md5"]
        if key == "cfg4":
            g = json.load(open(os.path.join(ROOT, "tests", "golden", "big.json"))).get("cfg4")
            if g and g["pairs"] == units and g["seed"] == seed and k == g["k"] and g["n_quals"] == 40:
                return g["rfq_md5"]
        if key == "cfg1":
            for g in json.load(open(os.path.join(ROOT, "tests", "golden", "generated.json"))):
                if g["profile"] == 0 and g["reads"] == units and g["seed"] == seed and g["nppm"] == 20 and g["k"] == k and not g["nonl"]:
                    return g["rfq_md5"]
    except (OSError, KeyError, ValueError):
        pass
    return None


class Workload:
    """One synthetic input resident in HBM + the encode/decode step over it."""

    def __init__(self, codec, dev, key, units, seed, chunk_kb, decode=True):
        import torch
        import _oracle as O
        from repaq_amd import SE, PE_TWO_FILES
        label, prof, dunits, dseed, kw = WORKLOADS[key]
        self.key, self.units, self.seed, self.k = key, units or dunits, (dseed if seed is None else seed), chunk_kb
        self.paired = prof in (1, 3)
        self.a1, self.a2 = O.gen_np(prof, self.units, seed=self.seed, **kw)
        self.n1, self.n2 = int(self.a1.size), int(self.a2.size) if self.paired else 0
        self.n = self.n1 + self.n2
        self.label = label % ((self.n / (2e9 if self.paired else 1e9)), self.units, self.seed, chunk_kb)
        self.t1 = torch.from_numpy(self.a1).to(dev); self.t2 = torch.from_numpy(self.a2).to(dev) if self.paired else None
        self.o1 = torch.empty(self.n1 + 64, dtype=torch.uint8, device=dev); self.o2 = torch.empty(self.n2 + 64, dtype=torch.uint8, device=dev) if self.paired else None
        self.codec, self.mode, self.cb, self.do_decode = codec, (PE_TWO_FILES if self.paired else SE), max(100, chunk_kb) * 1000, decode
        self.stage = {}; self.enc_s = 0.0; self.dec_s = 0.0; self.rfq_len = 0; self.chunks = 0; self.r = None; self.step_s = []

    def step(self, collect):
        c = self.codec
        c.clearHeader()
        t0 = time.perf_counter()
        r = c.encode(self.t1.data_ptr(), self.n1, self.t2.data_ptr() if self.paired else None, self.n2, self.mode, self.cb)
        t1 = time.perf_counter()
        if collect:
            for name, ms in c.timings():
                self.stage[name] = self.stage.get(name, 0.0) + ms
        self.r, self.rfq_len, self.chunks = r, r.rfq_len, r.n_chunks
        t2 = t1
        if self.do_decode:
            # the decode gets the image and nothing else, like a .rfq file gives it (the format has no chunk index, src/rfqchunk.cpp:161-228): the library finds
            # the chunk starts itself (guess and verify: k_dec_gw_find / walk / stitch + the verifying parse).  A host that has just encoded the image could pass
            # its chunk offsets on (rfq_decode_args.h_chunk_off): that rate is reported beside this one as decode_MBps_indexed (value = what a file allows)
            d = c.decode(r.d_rfq, r.rfq_len, split_pe=self.paired, d_out1=self.o1.data_ptr(), cap1=self.n1 + 64,
                         d_out2=self.o2.data_ptr() if self.paired else None, cap2=(self.n2 + 64) if self.paired else 0)
            t2 = time.perf_counter()
            self.d = d
            if collect:
                for name, ms in c.timings():
                    self.stage["dec:" + name] = self.stage.get("dec:" + name, 0.0) + ms
        if collect:
            self.enc_s += t1 - t0; self.dec_s += t2 - t1; self.step_s.append(t2 - t0)
        return r

    def check(self):
        """Parity of what is being measured: .rfq md5 against the reference's golden (or the oracle on small inputs), decode == input."""
        import torch
        import _oracle as O
        r = self.step(False)
        got = self.codec.dev_get(r.d_rfq, r.rfq_len)
        md5 = hashlib.md5(got).hexdigest()
        gold = golden_md5(self.key, self.units, self.seed, self.k)
        if gold:
            assert md5 == gold, "GPU .rfq md5 %s != reference golden %s" % (md5, gold)
            parity = "rfq md5 == reference golden (%s)" % md5
        elif self.n <= 300_000_000:
            want = O.encode_file(self.a1.tobytes(), self.a2.tobytes() if self.paired else b"", self.mode, self.cb)
            assert got == want, "GPU .rfq differs from the oracle"
            parity = "rfq bytes == oracle (%s)" % md5
        else:
            # no golden at this size: the leading chunks against the oracle's encoding of a file prefix (the cut rule and every chunk
            # image but the prefix's tail chunk are the same for the prefix and the whole file: chunks are independent)
            u = min(self.units, 200_000)
            c1 = offset_of_record(self.a1, u); c2 = offset_of_record(self.a2, u) if self.paired else 0
            want = O.encode_file(self.a1[:c1].tobytes(), self.a2[:c2].tobytes() if self.paired else b"", self.mode, self.cb)
            offs = O.chunk_table(want); cut = offs[-2] if len(offs) > 2 else 0
            assert cut > 0 and got[:cut] == want[:cut], "GPU .rfq differs from the oracle in the leading chunks"
            parity = "leading %d chunks == oracle, rfq md5 %s (no reference golden at this size)" % (len(offs) - 2, md5)
        if self.do_decode:
            assert self.d.n1 == self.n1 and torch.equal(self.o1[:self.n1], self.t1), "decoded R1 differs from the input FASTQ"
            if self.paired:
                assert self.d.n2 == self.n2 and torch.equal(self.o2[:self.n2], self.t2), "decoded R2 differs from the input FASTQ"
            parity += "; decode == input (both mates)" if self.paired else "; decode == input"
        return parity

    def run(self, steps, warmup, sync, barrier):
        for _ in range(warmup):
            self.step(False)
        barrier(); sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(True)
        sync(); barrier()
        return time.perf_counter() - t0

    def decode_indexed(self, steps, sync):
        """Decode throughput WITH the encoder's chunk index handed over (rfq_decode_args.h_chunk_off: what a host that has just encoded the image, or walked the
        chunk headers while the image was on its way to the GPU, can pass; every extent is still verified on the device).  The headline decode runs without it.
        Returns (MB/s, walk stage ms)."""
        c = self.codec; r = self.r
        kw = dict(split_pe=self.paired, d_out1=self.o1.data_ptr(), cap1=self.n1 + 64, d_out2=self.o2.data_ptr() if self.paired else None, cap2=(self.n2 + 64) if self.paired else 0,
                  chunk_off=r.h_chunk_off, n_chunks=r.n_chunks)
        c.decode(r.d_rfq, r.rfq_len, **kw); sync()
        walk = 0.0; t0 = time.perf_counter()
        for _ in range(steps):
            c.decode(r.d_rfq, r.rfq_len, **kw)
            walk += dict(c.timings()).get("walk", 0.0)
        sync()
        return round(self.n * steps / (time.perf_counter() - t0) / 1e6, 1), round(walk / steps, 3)

    def summary(self, steps):
        stage = {k: v / steps for k, v in self.stage.items()}
        enc_ms = sum(v for k, v in stage.items() if not k.startswith("dec:")); dec_ms = sum(v for k, v in stage.items() if k.startswith("dec:"))
        return stage, enc_ms, dec_ms


def roofline_of(w, stage, enc_ms, dec_ms, traffic_key, live=None, live_note=None):
    """The dominant kernel (longest HIP-event stage of either direction) against the HBM roofline, on ALGORITHMIC bytes (SURVEY.md
    §8(d): B_fastq + B_rfq per direction of a batch)."""
    if not stage:
        return None
    dom = dominant_kernel_stage(stage); longest = max(stage, key=stage.get)
    alg = float(w.n + w.rfq_len)
    traffic = None; kernels = None; source = None
    if live:
        kernels, source = live, live_note
    else:
        pjs = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
        pj = os.path.join(ROOT, "profiles", pjs[-1]) if pjs else ""
        if pj and os.path.exists(pj):
            pmc = json.load(open(pj)).get(traffic_key)
            if pmc and pmc.get("units") == w.units:
                kernels = pmc["kernels"]
                source = "NOT measured in this run (%s): the committed PMC passes of this workload, profiles/%s" % (live_note or "not asked for", pjs[-1])
    def kernel_traffic(names):
        ks = [k for k in names if kernels and k in kernels]
        return int(sum(kernels[k]["fetch_bytes"] + kernels[k]["write_bytes"] for k in ks)) if ks else None
    traffic = kernel_traffic(STAGE_KERNELS.get(dom, []))
    per_step = None
    if kernels and live:
        tot = {"encode": 0.0, "decode": 0.0}
        for k, v in kernels.items():
            for side, share in side_of(k).items():
                tot[side] += share * (v["fetch_bytes"] + v["write_bytes"])
        top = sorted(((k, v["fetch_bytes"], v["write_bytes"]) for k, v in kernels.items() if k.startswith("k_")), key=lambda t: -(t[1] + t[2]))[:12]
        per_step = {"encode_bytes": int(tot["encode"]), "decode_bytes": int(tot["decode"]), "encode_per_fastq_byte": round(tot["encode"] / w.n, 3), "decode_per_fastq_byte": round(tot["decode"] / w.n, 3),
                    "algorithmic_per_fastq_byte": round(alg / w.n, 3), "top_kernels_MB": {k: [round(f / 1e6, 1), round(wr / 1e6, 1)] for k, f, wr in top}}
    # the longest single kernel of the ENCODE side next to the dominant one (which has been the emitter of the decode side)
    enc_dom = max((k for k in stage if k in KERNEL_STAGES and not k.startswith("dec:")), key=stage.get, default=None)
    enc_kernel = None
    if enc_dom:
        a = alg / (stage[enc_dom] * 1e-3) / 1e9
        enc_kernel = {"kernel": STAGE_KERNELS.get(enc_dom, [enc_dom])[0], "stage": enc_dom, "avg_launch_ms": round(stage[enc_dom], 4), "achieved": round(a, 1), "frac": round(a / HBM_PEAK_GBS, 4),
                      "traffic": kernel_traffic(STAGE_KERNELS.get(enc_dom, [])[:1])}
    ach = alg / (stage[dom] * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": STAGE_KERNELS.get(dom, [dom])[0], "stage": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "longest_stage": {"stage": longest, "ms": round(stage[longest], 4), "kernels": STAGE_KERNELS.get(longest, [longest])},
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": source, "hbm_traffic_per_step": per_step, "encode_kernel": enc_kernel,
            "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": round(stage[dom], 4),
            "note": "achieved = one direction's algorithmic bytes (B_fastq + B_rfq) / the HIP-event time of the longest single kernel (its stage holds nothing else of weight): an upper bound for that kernel; "
                    "longest_stage = the longest timed phase, which may be several kernels on two streams; "
                    "whole_*_frac divide the same bytes by the whole direction's device time; traffic = that kernel's HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KB), "
                    "see traffic_source; hbm_traffic_per_step = the same counters summed over all kernels of a direction; encode_kernel = the encode side's longest single kernel",
            "whole_encode_frac": round(alg / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if enc_ms else None,
            "whole_decode_frac": round(alg / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dec_ms else None}


def pmc_traffic_live(workload, chunk_kb, units, seed, timeout_s=300):
    """HBM traffic per kernel of ONE step of `workload`, measured now: this script re-executes itself twice under rocprofv3 - `--pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE`, each in its own run with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (KB units; FETCH_SIZE x 2 on
    gfx950: tools/pmc_summary.py) - on one warm-up and one timed step, and sums every kernel's counters over a step.  None when rocprofv3 is not there or a
    pass fails (the caller then falls back to the committed profiles/*_pmc_traffic.json and says so)."""
    import glob
    import shutil
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_summary as P
    except Exception as e:                               # noqa: BLE001
        return None, "tools/pmc_summary.py: %s" % e
    got = {}
    work = tempfile.mkdtemp(prefix="rfq_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", workload,
                   "--steps", "1", "--warmup", "1", "--chunk-kb", str(chunk_kb), "--no-cpu-baseline", "--no-secondary", "--no-verify", "--no-pmc"]
            if units:
                cmd += ["--units", str(units)]
            if seed is not None:
                cmd += ["--seed", str(seed)]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            got[counter] = P.per_kernel(dbs[0], counter, last_step=True)
    except Exception as e:                               # noqa: BLE001
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    passes = 1.0                                         # the child runs the step twice (one warm-up, one timed); the counters of the LAST step are taken (a warm-up step
                                                         # of a fresh context may repeat a batch whose arenas it sized too small)
    ker = {}
    for k in set(got["FETCH_SIZE"]) | set(got["WRITE_SIZE"]):
        nf, vf = got["FETCH_SIZE"].get(k, (0, 0.0)); nw, vw = got["WRITE_SIZE"].get(k, (0, 0.0))
        ker[k] = {"fetch_bytes": 2.0 * vf * 1024 / passes, "write_bytes": vw * 1024 / passes, "launches_per_step": max(nf, nw) / passes}
    return ker, "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two child runs of this script, --kernel-trace only, one step each; KB units, FETCH_SIZE x 2 on gfx950)"


def blocks_of(step_s, bytes_per_step, nblocks=4):
    """The timed steps cut into `nblocks` consecutive blocks: each block's rate, their median and spread (one number per run hides the
    box's clock behaviour; `value` stays the contract's K steps in one bracket)."""
    if len(step_s) < nblocks:
        nblocks = max(1, len(step_s))
    if not step_s:
        return None
    per = len(step_s) // nblocks
    rates = [round(bytes_per_step * per / sum(step_s[i * per:(i + 1) * per]) / 1e6, 1) for i in range(nblocks)]
    srt = sorted(rates); med = srt[len(srt) // 2] if len(srt) % 2 else round((srt[len(srt) // 2 - 1] + srt[len(srt) // 2]) / 2, 1)
    return {"steps_per_block": per, "MBps": rates, "median_MBps": med, "min_MBps": srt[0], "max_MBps": srt[-1]}


def smi_clocks(local):
    """Current shader / memory clock of the device as the SMI reports them right after the timed steps (sysfs first, rocm-smi as a fallback); None where
    neither can be read (an unprivileged container)."""
    import glob
    import re
    out = {}
    try:
        import torch
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        for key, f in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk")):
            for path in glob.glob("/sys/bus/pci/devices/%s/%s" % (bdf, f)):
                cur = [l for l in open(path).read().splitlines() if l.strip().endswith("*")]
                if cur:
                    out[key + "_MHz"] = int(re.search(r"(\d+)\s*Mhz", cur[0], re.I).group(1))
    except Exception:                                    # noqa: BLE001
        pass
    if not out:
        try:
            r = subprocess.run(["rocm-smi", "-d", str(local), "--showclocks", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20, text=True)
            j = json.loads(r.stdout)
            for card in j.values():
                for k, v in card.items():
                    m = re.search(r"\((\d+)Mhz\)", str(v))
                    if m and "sclk" in k.lower():
                        out["sclk_MHz"] = int(m.group(1))
                    if m and "mclk" in k.lower():
                        out["mclk_MHz"] = int(m.group(1))
        except Exception:                                # noqa: BLE001
            pass
    return out or None


def side_of(kernel):
    """which direction a kernel belongs to, for the per-step traffic totals (scans: the encode uses the u64 / u32 ones - which return at once on reads of one
    length - and the decode one small U4 scan over the chunks' text totals; round 4's decode ran two U4 scans and a u32 one over every read)"""
    if kernel.startswith("k_dec_"):
        return {"decode": 1.0}
    if kernel.startswith("k_scan_"):
        return {"decode": 1.0} if "U4" in kernel else {"encode": 1.0}
    if kernel.startswith("k_") :
        return {"encode": 1.0}
    return {}


def line_of(w, steps, dt, total_bytes, parity):
    stage, enc_ms, dec_ms = w.summary(steps)
    passes = 2 if w.do_decode else 1
    return {"workload": w.label, "value_MBps": round(total_bytes * passes * steps / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 3),
            "chunks": w.chunks, "rfq_over_fastq": round(w.rfq_len / w.n, 4),
            "encode_MBps": round(w.n * steps / w.enc_s / 1e6, 1) if w.enc_s else None, "decode_MBps": round(w.n * steps / w.dec_s / 1e6, 1) if w.dec_s else None,
            "parity": parity, "stage_ms": {k: round(v, 3) for k, v in stage.items()}}, stage, enc_ms, dec_ms


SEG_PAIRS, SEG_SEED0, SEGS_PER_GPU, HEAD_PAIRS = 2_800_000, 4000, 8, 4000


def peak_rss_gb():
    import resource
    return round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2)          # (ru_maxrss is in KB on Linux)


def segments_to_device(dev, seg_ids, seg_pairs, head_seed=None, head_pairs=0, threads=8):
    """Segments `seg_ids` of the logical input (segment s = fqgen profile 1, `seg_pairs` pairs, seed SEG_SEED0 + s), then the first `head_pairs` pairs of segment
    `head_seed`, laid end to end in ONE device buffer per stream.  The generator's threads hand each finished segment to the main thread, which copies it to its
    place on the device and drops it: at most `threads` + 1 segments (2 x 1 GB each at full size) are alive on the host at any time - round 4 concatenated all of
    a rank's segments on the host first (np.concatenate: twice the share, 32 GB per rank at configs[3] size).  Returns (t1, t2, [(n1, n2) per segment],
    bytes of the head in each stream)."""
    import collections
    import torch
    from concurrent.futures import ThreadPoolExecutor
    import _oracle as O
    cap = (len(seg_ids) * seg_pairs + head_pairs) * 360 + 4096                          # (a NovaSeq-profile record is at most 359 bytes: see _oracle.gen_np)
    t1 = torch.empty(cap, dtype=torch.uint8, device=dev); t2 = torch.empty(cap, dtype=torch.uint8, device=dev)
    sizes, o1, o2 = [], 0, 0
    jobs = [(SEG_SEED0 + s_, seg_pairs) for s_ in seg_ids] + ([(SEG_SEED0 + head_seed, head_pairs)] if head_pairs else [])
    with ThreadPoolExecutor(max(1, threads)) as ex:
        pend = collections.deque(); it = iter(jobs)
        def feed():
            while len(pend) < max(1, threads):
                j = next(it, None)
                if j is None:
                    return
                pend.append(ex.submit(lambda sd, n_: O.gen_np(O.NOVA_PE150, n_, seed=sd), j[0], j[1]))
        feed()
        while pend:
            a, b = pend.popleft().result()
            feed()
            t1[o1:o1 + a.size].copy_(torch.from_numpy(a)); t2[o2:o2 + b.size].copy_(torch.from_numpy(b))
            sizes.append((int(a.size), int(b.size))); o1 += int(a.size); o2 += int(b.size)
            del a, b
    head = sizes.pop() if head_pairs else (0, 0)
    return t1, t2, sizes, head


def pin_to_gpu_numa(local):
    """Bind this process to the CPUs of the NUMA node its GPU hangs off (set-up work - generating the rank's text, the host side of hipMemcpy - then stays off
    the other sockets' memory).  Returns the node, or None when the topology cannot be read (nothing is changed then)."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return node
    except Exception:                                    # noqa: BLE001
        pass
    return None


def run_multi(args, rank, world, local):
    """N > 1: ONE configs[3]-shaped input - NovaSeq PE150, N x (2 x 8 GB) - encoded chunk-parallel and decoded again, every GPU working on
    the part of the text that is resident in its own HBM.

    The logical input is the concatenation of 8 N segments (segment s = fqgen profile 1, 2.8 M pairs, seed 4000 + s; N = 8: 2 x 64 GB).
    Rank r generates segments 8r .. 8r+7 (its share: 2 x 8 GB) plus the first 4000 pairs of segment 8r+8 (the head the chunk that
    straddles the share boundary runs into).  Setup (untimed): the plan chain (repaq_amd.dist.plan_shares: where each rank's first
    chunk starts - one small host message per rank), and rank 0's header to every rank (<= 272 bytes).  A step (timed): every rank
    encodes its byte range with rfq_encode_batch (the call runs slice by slice inside the library: >= 4 GiB per stream) and decodes
    its own chunk images back; no collective, no RCCL - rendezvous is gloo over the host.  The N images in rank order ARE the .rfq:
    rank 0 checks every chunk of it (crc32 + size, gathered over the host) against the reference's own encoding of the same logical
    input (tests/golden/cfg3.json), every rank checks its decoded text against its input."""
    import struct
    import zlib
    import numpy as np
    import torch
    import torch.distributed as dist
    import _oracle as O
    from repaq_amd import RfqCodec, PE_TWO_FILES, dist as D
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    codec = RfqCodec(device=local)
    seg_pairs, per = args.seg_pairs, args.segs_per_gpu
    if args.strong:                                   # ONE fixed input (--segs-per-gpu segments in all) split over the ranks
        assert per % world == 0, "--strong: --segs-per-gpu (%d: the segments of the whole input) must be a multiple of the number of GPUs" % per
        per = per // world
    cb = max(100, args.chunk_kb) * 1000
    # ---- this rank's share (+ the head of the next one), generated straight into one buffer per stream
    # (set-up, untimed: the generator is a C loop that drops the GIL - the rank's segments are made on several threads, on the cores next to its GPU)
    numa = pin_to_gpu_numa(local)
    t_gen = time.perf_counter()
    gen_threads = max(1, min(per, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)) // max(1, world if numa is None else 1), 8))
    t1, t2, sizes, head = segments_to_device(dev, range(per * rank, per * rank + per), seg_pairs, head_seed=per * (rank + 1), head_pairs=(min(HEAD_PAIRS, seg_pairs) if rank < world - 1 else 0),
                                             threads=gen_threads)
    share1, share2 = sum(x[0] for x in sizes), sum(x[1] for x in sizes)
    avail1, avail2 = share1 + head[0], share2 + head[1]
    gen_s = time.perf_counter() - t_gen
    lens = [None] * world
    dist.all_gather_object(lens, (share1, share2))
    off1, off2 = sum(x[0] for x in lens[:rank]), sum(x[1] for x in lens[:rank])
    # ---- plan chain, header
    plan = {}
    torch.cuda.synchronize(); D.barrier()
    cut1, cut2, n1, n2 = D.plan_shares(codec, rank, world, t1.data_ptr(), share1, avail1, t2.data_ptr(), share2, avail2, PE_TWO_FILES, cb, stats=plan)
    plan_ms = D.reduce_max_sum(plan.get("plan_ms", 0.0) / 1e3, 0)[0] * 1e3      # (untimed set-up, like opening the files: reported, max over ranks)
    last = rank == world - 1
    o1 = torch.empty(n1 + 64, dtype=torch.uint8, device=dev); o2 = torch.empty(n2 + 64, dtype=torch.uint8, device=dev)
    state = {"stage": {}, "enc_s": 0.0, "dec_s": 0.0}

    def step(collect):
        if rank == 0:
            codec.clearHeader()
        t_a = time.perf_counter()
        r = codec.encode(t1.data_ptr() + cut1, n1, t2.data_ptr() + cut2, n2, PE_TWO_FILES, cb, final=last, emit_header=(rank == 0),
                         file_off1=off1 + cut1, file_off2=off2 + cut2, flush_all=not last)
        t_b = time.perf_counter()
        if collect:
            for name, ms in codec.timings():
                state["stage"][name] = state["stage"].get(name, 0.0) + ms
        d = None
        if not args.encode_only:
            d = codec.decode(r.d_rfq, r.rfq_len, has_header=(rank == 0), split_pe=True, final=last, d_out1=o1.data_ptr(), cap1=n1 + 64, d_out2=o2.data_ptr(), cap2=n2 + 64,
                             chunk_off=r.h_chunk_off, n_chunks=r.n_chunks)
            if collect:
                for name, ms in codec.timings():
                    state["stage"]["dec:" + name] = state["stage"].get("dec:" + name, 0.0) + ms
        t_c = time.perf_counter()
        if collect:
            state["enc_s"] += t_b - t_a; state["dec_s"] += t_c - t_b
        return r, d

    if rank == 0:
        step(False)                                  # makes the header from chunk 0 (RfqCodec::makeHeader, src/rfqcodec.cpp:59-145)
    hdr = D.share_header(codec)
    # ---- parity of what is being measured
    r, d = step(False)
    parity = "unchecked"
    if not args.no_verify:
        assert r.consumed1 == n1 and r.consumed2 == n2, "rank %d: the range was not encoded whole" % rank
        img = codec.dev_get(r.d_rfq, r.rfq_len)
        offs = [r.h_chunk_off[i] for i in range(r.n_chunks + 1)]
        mine = [(zlib.crc32(img[offs[i]:offs[i + 1]]) & 0xFFFFFFFF, offs[i + 1] - offs[i]) for i in range(r.n_chunks)]
        del img
        if d is not None:
            assert d.n1 == n1 and d.n2 == n2 and torch.equal(o1[:n1], t1[cut1:cut1 + n1]) and torch.equal(o2[:n2], t2[cut2:cut2 + n2]), "rank %d: decoded text differs from the input" % rank
        allc = [None] * world
        dist.all_gather_object(allc, mine)
        if rank == 0:
            chunks = [c for part in allc for c in part]
            parity = "%d chunks over %d ranks" % (len(chunks), world)
            try:
                g = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg3.json")))
            except OSError:
                g = None
            if g and g["seg_pairs"] == seg_pairs and g["seed0"] == SEG_SEED0 and g["segments"] >= per * world and cb == 1_000_000:
                assert hdr.hex() == g["header_hex"], "header differs from the reference's"
                G = g["group"]; whole = len(chunks) if g["segments"] == per * world else len(chunks) - 1       # (a shorter input ends in a tail chunk of its own)
                if g["segments"] == per * world:
                    assert len(chunks) == g["n_chunks"], "chunk count differs from the reference's"
                ok = 0
                for gi in range(whole // G):
                    hh = hashlib.md5()
                    for c in chunks[gi * G:(gi + 1) * G]:
                        hh.update(struct.pack("<II", c[0], c[1]))
                    assert hh.hexdigest()[:16] == g["group_md5"][gi], "chunks %d..%d differ from the reference's" % (gi * G, gi * G + G - 1)
                    ok += G
                parity += ": header and %d chunk images (crc32 + size, groups of %d) == reference golden" % (ok, G)
            else:
                parity += " (no reference golden for this shape)"
            parity += "; every rank: decode == its input text" if d is not None else ""
    # ---- timed
    for _ in range(args.warmup):
        step(False)
    D.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize(); D.barrier()
    dt = time.perf_counter() - t0
    my_dt = dt
    dt, total = D.reduce_max_sum(dt, n1 + n2)
    rfq_total = D.reduce_max_sum(0.0, r.rfq_len)[1]
    # every rank's own line (a straggler must be visible in the one JSON object rank 0 prints)
    K_ = args.steps
    mine_line = {"rank": rank, "gpu": local, "numa_node": numa, "fastq_bytes": n1 + n2, "chunks": r.n_chunks, "s_per_step": round(my_dt / K_, 5),
                 "encode_MBps": round((n1 + n2) * K_ / state["enc_s"] / 1e6, 1) if state["enc_s"] else None, "decode_MBps": round((n1 + n2) * K_ / state["dec_s"] / 1e6, 1) if state["dec_s"] else None,
                 "setup_generate_s": round(gen_s, 1), "gen_threads": gen_threads, "peak_host_rss_gb": peak_rss_gb(), "stage_ms": {k: round(v / K_, 3) for k, v in state["stage"].items()}}
    ranks = [None] * world
    dist.all_gather_object(ranks, mine_line)
    if rank == 0:
        K = args.steps; passes = 1 if args.encode_only else 2
        stage = {k: v / K for k, v in state["stage"].items()}
        enc_ms = sum(v for k, v in stage.items() if not k.startswith("dec:")); dec_ms = sum(v for k, v in stage.items() if k.startswith("dec:"))
        alg = float(n1 + n2 + r.rfq_len); dom = dominant_kernel_stage(stage) if stage else None
        out = {"metric": "raw FASTQ MB/s encode+decode" if passes == 2 else "raw FASTQ MB/s encode",
               "value": round(total * passes * K / dt / 1e6, 1), "unit": "MB/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
               "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "configs[3] shape: ONE synthetic NovaSeq PE150 input of %d x (2 x %.2f GB) = 2 x %.1f GB FASTQ (%d segments: fqgen profile 1, %d pairs, seed %d + s), -k %d, "
                                      "chunk-parallel over %d GPUs (each encodes + decodes the byte range resident in its HBM; plan + header over the host, no RCCL)"
                                      % (world, share1 / 1e9, total / 2e9, per * world, seg_pairs, SEG_SEED0, args.chunk_kb, world),
                          "rfq_over_fastq": round(rfq_total / total, 4), "parity": parity,
                          "plan": plan.get("plan"), "plan_ms": round(plan_ms, 2), "strong": bool(args.strong), "ranks": ranks,
                          "rank0": {"chunks": r.n_chunks, "encode_MBps": round((n1 + n2) * K / state["enc_s"] / 1e6, 1), "decode_MBps": round((n1 + n2) * K / state["dec_s"] / 1e6, 1) if state["dec_s"] else None,
                                    "stage_ms": {k: round(v, 3) for k, v in stage.items()}}},
               "roofline": None if not dom else {"bound": "hbm", "kernel": STAGE_KERNELS.get(dom, [dom])[0], "stage": dom, "achieved": round(alg / (stage[dom] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                                 "unit": "GB/s", "frac": round(alg / (stage[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "note": "rank 0's share, per GPU; see the N = 1 line for the definitions",
                                                 "whole_encode_frac": round(alg / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if enc_ms else None,
                                                 "whole_decode_frac": round(alg / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dec_ms else None}}
        print(json.dumps(out))
    codec.close()
    dist.destroy_process_group()


def run_queue(args, rank, world, local, emit=True):
    """--queue: the host work queue north_star names, in bench form.  ONE configs[3]-shaped logical input (--segs-per-gpu x N segments; N = 8: 2 x 64 GB) is resident in
    the HBM of EVERY GPU (128 GB of text at N = 8: it fits 288 GB), planned once (rfq_scan_batch: where every chunk ends), and cut into batches of --queue-chunks whole
    chunks.  A step: every rank pulls batch numbers from ONE shared counter (the rendezvous store's atomic add; a local counter for N = 1) until none are left, and
    encodes (+ decodes) each batch it got - rank-agnostic, no static shares, no data-path collective; the next ticket is fetched while the GPU works on the current
    batch.  Parity (untimed pass): the (crc32, size) of every chunk image, gathered by batch number, against the reference's table for the whole input
    (tests/golden/cfg3.json) - with N = 1 this is the whole configs[3] input through ONE GPU - and every decoded batch against its text.
    Counterpart in the product: repaq_hip --devices (do_compress_multi / do_decompress_multi), which also moves the text over PCIe."""
    import struct
    import threading
    import zlib
    import torch
    import torch.distributed as dist
    from repaq_amd import RfqCodec, PE_TWO_FILES, dist as D
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # --queue-workers contexts per GPU, each on a host thread of its own (one rfq_ctx per (host thread, GPU): include/rfq_hip.h): a batch's chain of small kernels
    # and its host round trips - about half of a 256-chunk batch's 1.4 ms, profiles/r06_b_queue_timeline_256.txt - run under another batch's large kernels
    W = max(1, args.queue_workers)
    codecs = [RfqCodec(device=local) for _ in range(W)]
    codec = codecs[0]
    seg_pairs, nseg = args.seg_pairs, args.segs_per_gpu * (1 if args.strong else world)
    cb = max(100, args.chunk_kb) * 1000
    numa = pin_to_gpu_numa(local)
    t_gen = time.perf_counter()
    gen_threads = max(1, min(nseg, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)) // max(1, world if numa is None else 1), 8))
    t1, t2, sizes, _ = segments_to_device(dev, range(nseg), seg_pairs, threads=gen_threads)
    n1, n2 = sum(x[0] for x in sizes), sum(x[1] for x in sizes)
    gen_s = time.perf_counter() - t_gen
    # ---- the plan: every chunk's end in both streams (every rank scans the same text: set-up, untimed)
    torch.cuda.synchronize(); D.barrier()
    t_plan = time.perf_counter()
    sr, e1, e2 = codec.scan(t1.data_ptr(), n1, t2.data_ptr(), n2, PE_TWO_FILES, cb, final=True)
    plan_ms = (time.perf_counter() - t_plan) * 1e3
    nc = sr.n_chunks; B = max(1, args.queue_chunks); nb = (nc + B - 1) // B
    cut1 = [0] + [e1[min(nc, (b + 1) * B) - 1] for b in range(nb)]; cut2 = [0] + [e2[min(nc, (b + 1) * B) - 1] for b in range(nb)]
    cut1[-1], cut2[-1] = n1, n2                                                   # (the last batch runs to the end of the input: final)
    big1 = max(cut1[b + 1] - cut1[b] for b in range(nb)); big2 = max(cut2[b + 1] - cut2[b] for b in range(nb))
    outs = [(torch.empty(big1 + 64, dtype=torch.uint8, device=dev), torch.empty(big2 + 64, dtype=torch.uint8, device=dev)) for _ in range(W)]
    store = dist.distributed_c10d._get_default_store() if world > 1 else None
    local_ctr = {}; ctr_lock = threading.Lock()

    def take(key):                                                                # the shared counter: the next batch nobody has taken yet
        if store is not None:
            return int(store.add(key, 1)) - 1
        with ctr_lock:
            local_ctr[key] = local_ctr.get(key, -1) + 1
            return local_ctr[key]

    state = {"stage": {}, "enc_s": 0.0, "dec_s": 0.0, "batches": 0, "bytes": 0}
    st_lock = threading.Lock()

    def one(b, collect, check=None, w=0):
        codec = codecs[w]; o1, o2 = outs[w]
        last = b == nb - 1
        a1, a2 = cut1[b], cut2[b]; m1, m2 = cut1[b + 1] - a1, cut2[b + 1] - a2
        t_a = time.perf_counter()
        r = codec.encode(t1.data_ptr() + a1, m1, t2.data_ptr() + a2, m2, PE_TWO_FILES, cb, final=last, emit_header=(b == 0), file_off1=a1, file_off2=a2, flush_all=not last)
        t_b = time.perf_counter()
        stages = list(codec.timings()) if collect else []
        d = None
        if not args.encode_only:
            d = codec.decode(r.d_rfq, r.rfq_len, has_header=(b == 0), split_pe=True, final=last, d_out1=o1.data_ptr(), cap1=big1 + 64, d_out2=o2.data_ptr(), cap2=big2 + 64)
            if collect:
                stages += [("dec:" + name, ms) for name, ms in codec.timings()]
        if collect:
            t_c = time.perf_counter()
            with st_lock:
                for name, ms in stages:
                    state["stage"][name] = state["stage"].get(name, 0.0) + ms
                state["enc_s"] += t_b - t_a; state["dec_s"] += t_c - t_b; state["batches"] += 1; state["bytes"] += m1 + m2
        if check is not None:
            assert r.consumed1 == m1 and r.consumed2 == m2, "batch %d was not encoded whole" % b
            img = codec.dev_get(r.d_rfq, r.rfq_len); offs = [r.h_chunk_off[i] for i in range(r.n_chunks + 1)]
            check[b] = [(zlib.crc32(img[offs[i]:offs[i + 1]]) & 0xFFFFFFFF, offs[i + 1] - offs[i]) for i in range(r.n_chunks)]
            if d is not None:
                assert d.n1 == m1 and d.n2 == m2 and torch.equal(o1[:m1], t1[a1:a1 + m1]) and torch.equal(o2[:m2], t2[a2:a2 + m2]), "batch %d: decoded text differs from the input" % b
        return r

    def drain(key, collect, check=None):
        """pull batches until the counter runs past the last one: every worker thread of this rank with its own context; a worker's next ticket is on its way while
        it works on the current batch"""
        errs = []

        def worker(w):
            try:
                torch.cuda.set_device(local)
                nxt = {"b": take(key)}
                while nxt["b"] < nb:
                    b = nxt["b"]
                    th = threading.Thread(target=lambda: nxt.__setitem__("b", take(key))) if store is not None else None
                    if th:
                        th.start()
                    one(b, collect, check, w)
                    if th:
                        th.join()
                    else:
                        nxt["b"] = take(key)
            except BaseException as e:                                            # noqa: BLE001
                errs.append(e)
        if W == 1:
            worker(0)
        else:
            ths = [threading.Thread(target=worker, args=(w,)) for w in range(W)]
            for t_ in ths:
                t_.start()
            for t_ in ths:
                t_.join()
        if errs:
            raise errs[0]

    # ---- header: batch 0 on rank 0 makes it (RfqCodec::makeHeader), every rank sets it
    if rank == 0:
        codec.clearHeader(); one(0, False)
    hdr = D.share_header(codec)
    for c_ in codecs[1:]:
        c_.setHeader(hdr)
    # ---- parity of what is being measured: one untimed pass over the queue
    parity = "unchecked"
    if not args.no_verify:
        mine = {}
        drain("verify", False, mine)
        allc = [None] * world
        if world > 1:
            dist.all_gather_object(allc, mine)
        else:
            allc = [mine]
        if rank == 0:
            got = {}
            for part in allc:
                got.update(part)
            assert sorted(got) == list(range(nb)), "the queue did not hand out every batch exactly once"
            chunks = [c for b in range(nb) for c in got[b]]
            parity = "%d chunks in %d batches over %d rank(s)" % (len(chunks), nb, world)
            try:
                g = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg3.json")))
            except OSError:
                g = None
            if g and g["seg_pairs"] == seg_pairs and g["seed0"] == SEG_SEED0 and g["segments"] >= nseg and cb == 1_000_000:
                assert hdr.hex() == g["header_hex"], "header differs from the reference's"
                G = g["group"]; whole = len(chunks) if g["segments"] == nseg else len(chunks) - 1
                if g["segments"] == nseg:
                    assert len(chunks) == g["n_chunks"], "chunk count differs from the reference's"
                ok = 0
                for gi in range(whole // G):
                    hh = hashlib.md5()
                    for c in chunks[gi * G:(gi + 1) * G]:
                        hh.update(struct.pack("<II", c[0], c[1]))
                    assert hh.hexdigest()[:16] == g["group_md5"][gi], "chunks %d..%d differ from the reference's" % (gi * G, gi * G + G - 1)
                    ok += G
                parity += ": header and %d chunk images (crc32 + size, groups of %d) == reference golden" % (ok, G)
            else:
                parity += " (no reference golden for this shape)"
            parity += "; every batch: decode == its input text" if not args.encode_only else ""
    # ---- timed: K passes over the queue
    for w_ in range(args.warmup):
        drain("warm%d" % w_, False)
    D.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k_ in range(args.steps):
        drain("step%d" % k_, True)
        D.barrier()                                                               # (a pass ends when the queue is empty AND every rank is done with what it took)
    torch.cuda.synchronize(); D.barrier()
    dt = time.perf_counter() - t0
    dt = D.reduce_max_sum(dt, 0)[0]
    # the encode direction alone, by the wall clock (with several worker contexts side by side a worker's own call times overlap: only a pass of its own tells the encode rate)
    enc_wall = None
    if not args.encode_only:
        keep_eo, args.encode_only = args.encode_only, True
        D.barrier(); torch.cuda.synchronize()
        t1_ = time.perf_counter()
        for k_ in range(args.steps):
            drain("enc%d" % k_, False)
            D.barrier()
        torch.cuda.synchronize(); D.barrier()
        enc_wall = D.reduce_max_sum(time.perf_counter() - t1_, 0)[0]
        args.encode_only = keep_eo
    K = args.steps; passes = 1 if args.encode_only else 2
    mine_line = {"rank": rank, "gpu": local, "numa_node": numa, "batches_per_step": round(state["batches"] / K, 1), "fastq_bytes_per_step": state["bytes"] // K,
                 "encode_MBps": round(state["bytes"] / state["enc_s"] / 1e6, 1) if state["enc_s"] else None, "decode_MBps": round(state["bytes"] / state["dec_s"] / 1e6, 1) if (state["dec_s"] and not args.encode_only) else None,
                 "setup_generate_s": round(gen_s, 1), "gen_threads": gen_threads, "peak_host_rss_gb": peak_rss_gb(), "stage_ms": {k: round(v / K, 3) for k, v in state["stage"].items()}}
    ranks = [None] * world
    if world > 1:
        dist.all_gather_object(ranks, mine_line)
    else:
        ranks = [mine_line]
    if rank == 0:
        total = n1 + n2
        out = {"metric": "raw FASTQ MB/s encode+decode" if passes == 2 else "raw FASTQ MB/s encode", "value": round(total * passes * K / dt / 1e6, 1), "unit": "MB/s", "n_gpus": world,
               "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
               "dtype": "u8", "data": "synthetic",
               "config": {"workload": "configs[3] shape through the host work QUEUE: ONE synthetic NovaSeq PE150 input of 2 x %.1f GB FASTQ (%d segments: fqgen profile 1, %d pairs, seed %d + s), -k %d, "
                                      "resident in every GPU's HBM; batches of %d chunks pulled from one shared counter by %d rank(s) x %d worker context(s) (encode%s per batch; plan + header over the host, no RCCL)"
                                      % (total / 2e9, nseg, seg_pairs, SEG_SEED0, args.chunk_kb, B, world, W, "" if args.encode_only else " + decode"),
                          "queue": True, "workers_per_gpu": W, "encode_MBps_wall": round(total * K / (enc_wall if enc_wall else dt) / 1e6, 1), "batches": nb, "chunks": nc, "parity": parity, "plan": "scan of the whole input on every rank", "plan_ms": round(plan_ms, 2), "strong": bool(args.strong), "ranks": ranks},
               "roofline": None}
        if emit:
            print(json.dumps(out))
    for c_ in codecs:
        c_.close()
    if world > 1:
        dist.destroy_process_group()
    return out if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2", help="N=1 headline workload (default: BASELINE.json configs[2])")
    ap.add_argument("--units", type=int, default=0, help="reads (SE) / pairs (PE) of the headline workload; 0 = the config's own size")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--chunk-kb", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="reads / pairs of the workload the CPU baseline is timed on")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] / configs[4] secondary lines")
    ap.add_argument("--no-queue-line", action="store_true", help="skip the host-work-queue secondary line (one GPU's share of configs[3] through bench.py --queue)")
    ap.add_argument("--encode-only", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the parity assertions (A/B runs of a switch or a macro that changes nothing but time)")
    ap.add_argument("--seg-pairs", type=int, default=SEG_PAIRS, help="N>1: pairs per segment of the logical input (test aid: smaller inputs)")
    ap.add_argument("--segs-per-gpu", type=int, default=SEGS_PER_GPU, help="N>1: segments per GPU share (configs[3]: 8 x 2.8 M pairs = 2 x 8 GB per GPU)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic in this run (two extra passes of one step under rocprofv3 --pmc): take the committed profiles/*_pmc_traffic.json")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)      # the counter passes: one warm-up + one step, nothing else
    ap.add_argument("--queue", action="store_true", help="the host work queue: the whole logical input resident on every GPU, batches of --queue-chunks chunks pulled from one shared counter (see run_queue); works with --gpus 1 too")
    ap.add_argument("--queue-chunks", type=int, default=1024, help="--queue: chunks per batch (profiles/r06_d_queue_sweep_workers.txt: 256 / 512 / 1024 chunks x 1 / 2 / 3 worker contexts)")
    ap.add_argument("--queue-workers", type=int, default=2, help="--queue: worker contexts (host threads) per GPU")
    ap.add_argument("--strong", action="store_true", help="N>1: strong scaling - the input is --segs-per-gpu segments IN ALL (default 8 = 2 x 8 GB), split over the N GPUs")
    args = ap.parse_args()

    import torch
    from repaq_amd import dist as D
    rank, world, local = D.env_rank()
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)" % (args.gpus, world)
    # test aid for 1-GPU boxes: RFQ_BENCH_SINGLE_DEVICE=1 puts every rank on GPU 0, so that the N>1 control flow can be exercised there
    single = os.environ.get("RFQ_BENCH_SINGLE_DEVICE") == "1"
    if single:
        local = 0
    if world > 1:
        D.init("gloo")          # host-side rendezvous only (barriers, max-over-ranks time, the <= 272-byte header, chunk hashes): no RCCL on this path
    if args.queue:
        return run_queue(args, rank, world, local)
    if world > 1:
        return run_multi(args, rank, world, local)

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from repaq_amd import RfqCodec
    codec = RfqCodec(device=local)     # raises loudly without the HIP library / a GPU: there is no fallback

    sync = torch.cuda.synchronize
    w = Workload(codec, dev, args.workload, args.units, args.seed, args.chunk_kb, decode=not args.encode_only)
    parity = "unchecked" if args.no_verify else w.check()
    dt = w.run(args.steps, args.warmup, sync, lambda: None)
    head, stage, enc_ms, dec_ms = line_of(w, args.steps, dt, w.n, parity)
    if args.pmc_child:
        print(json.dumps({"pmc_child": True, "value": head["value_MBps"]})); codec.close(); return
    indexed = w.decode_indexed(args.steps, sync) if w.do_decode else None
    live, live_note = (None, "--no-pmc") if args.no_pmc else pmc_traffic_live(args.workload, args.chunk_kb, args.units, args.seed)
    out = {
        "metric": "raw FASTQ MB/s encode+decode" if w.do_decode else "raw FASTQ MB/s encode",
        "value": head["value_MBps"], "unit": "MB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": w.label, "chunks_per_gpu": w.chunks, "rfq_over_fastq": head["rfq_over_fastq"],
                   "encode_MBps_per_gpu": head["encode_MBps"], "decode_MBps_per_gpu": head["decode_MBps"],
                   # the headline decode finds the chunk starts itself (a .rfq file has no index); the same decode handed the encoder's chunk offsets:
                   "decode_MBps_indexed": indexed[0] if indexed else None, "walk_ms_indexed": indexed[1] if indexed else None,
                   "value_MBps_indexed": round(2 * w.n / (w.n / (head["encode_MBps"] * 1e6) + w.n / (indexed[0] * 1e6)) / 1e6, 1) if indexed and head["encode_MBps"] else None,
                   # the K timed steps again as blocks (every step ends with the host holding its result, so a step's wall time is its own): median block rate, spread, device clocks
                   "blocks": blocks_of(w.step_s, w.n * (2 if w.do_decode else 1)), "clocks": smi_clocks(local),
                   "parity": parity, "stage_ms": head["stage_ms"]},
        "roofline": roofline_of(w, stage, enc_ms, dec_ms, args.workload, live, live_note),
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w.a1, w.a2, w.paired, w.units, args.cpu_sample)
        allc = cpu_baseline_all_cores(w.a1, w.a2, w.paired, w.units, max(1, args.cpu_sample // 8))
        if allc:
            out["cpu_baseline_all_cores"] = allc
    if not args.no_secondary and args.workload == "cfg2" and not args.units:
        sec = {}
        del w
        torch.cuda.empty_cache()
        for key in ("cfg1", "cfg4"):
            w2 = Workload(codec, dev, key, 0, None, args.chunk_kb, decode=not args.encode_only)
            p2 = "unchecked" if args.no_verify else w2.check()
            dt2 = w2.run(3, 1, sync, lambda: None)
            l2, st2, e2, d2 = line_of(w2, 3, dt2, w2.n, p2)
            # (the secondary workloads' HBM traffic is measured in this run too - VERDICT r4 #4: it used to be quoted from a committed JSON)
            lv2, note2 = (None, "--no-pmc") if args.no_pmc else pmc_traffic_live(key, args.chunk_kb, 0, None)
            l2["roofline"] = roofline_of(w2, st2, e2, d2, key, lv2, note2)
            sec[key] = l2
            del w2
            torch.cuda.empty_cache()
        out["secondary"] = sec
        if not args.no_queue_line:
            # the host work queue on this GPU (what a multi-GPU run multiplies): one GPU's share of the configs[3] input - 8 segments, 2 x 8 GB - in batches of
            # --queue-chunks chunks pulled from the counter by --queue-workers contexts; its own parity pass (every chunk image against the reference's table)
            import copy
            qa = copy.copy(args); qa.steps, qa.warmup = 3, 1
            codec.close(); codec = None
            q = run_queue(qa, 0, 1, local, emit=False)
            sec["queue"] = {"workload": q["config"]["workload"], "value_MBps": q["value"], "ms_per_step": q["ms_per_step"], "batches": q["config"]["batches"], "chunks": q["config"]["chunks"],
                            "queue_chunks": args.queue_chunks, "workers_per_gpu": q["config"]["workers_per_gpu"], "encode_MBps": q["config"]["encode_MBps_wall"], "parity": q["config"]["parity"], "plan_ms": q["config"]["plan_ms"],
                            "note": "value_MBps: encode + decode of every batch, wall clock; encode_MBps: a pass of encodes alone, wall clock"}
    print(json.dumps(out))
    if codec is not None:
        codec.close()


if __name__ == "__main__":
    main()
