#!/usr/bin/env python3
"""Copy one measurement round (gpurun_out/<tag>/, written by tools/measure_round.sh) into the tracked profiles/ directory as
profiles/<round>_<letter>_* and refresh profiles/<round>_pmc_traffic.json, the per-kernel HBM traffic bench.py reports as roofline.traffic.
usage: python tools/publish_profiles.py <tag> <letter> [round=r02] [units of the cfg2 PMC run = 11200000]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, letter = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r02"
units = int(sys.argv[4]) if len(sys.argv) > 4 else 11200000
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
pairs = [("bench.json.log", "bench.json.log"), ("bench_under_rocprof.json.log", "bench_under_rocprof.json.log"),
         ("trace_kernel_stats.txt", "kernel_stats_cfg2.txt"), ("trace_cfg1_kernel_stats.txt", "kernel_stats_cfg1.txt"),
         ("trace_cfg4_kernel_stats.txt", "kernel_stats_cfg4.txt"), ("pmc_traffic.txt", "pmc_traffic_cfg2.txt"), ("pmc_traffic_cfg1.txt", "pmc_traffic_cfg1.txt"), ("pmc_traffic_cfg4.txt", "pmc_traffic_cfg4.txt"),
         ("e2e_cli.txt", "e2e_cli.txt"), ("e2e_pe.txt", "e2e_pe.txt"), ("sq_counters.txt", "sq_counters.txt"),
         ("multi/gpus2_single_device.json.log", "gpus2_single_device.json.log"), ("multi/gpus4_single_device.json.log", "gpus4_single_device.json.log"), ("multi/gpus2_strong_single_device.json.log", "gpus2_strong_single_device.json.log"),
         ("queue/queue_gpus1_2seg.json.log", "queue_gpus1_2seg.json.log"), ("queue/queue_gpus2_single_device.json.log", "queue_gpus2_single_device.json.log"),
         ("queue/queue_cfg3_whole_one_gpu.json.log", "queue_cfg3_whole_one_gpu.json.log")]
for a, b in pairs:
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, "%s_%s_%s" % (rnd, letter, b)))
        print("profiles/%s_%s_%s" % (rnd, letter, b))
pj = os.path.join(src, "pmc_traffic.json")
if os.path.exists(pj):
    prev = os.path.join(dst, "%s_pmc_traffic.json" % rnd)
    out = json.load(open(prev)) if os.path.exists(prev) else {}          # (a kernels-only set re-measures configs[2] alone: the other workloads' entries stay)
    out["cfg2"] = {"kernels": json.load(open(pj)), "units": units,
                    "source": "profiles/%s_%s_pmc_traffic_cfg2.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, KB units, FETCH_SIZE x2 on gfx950)" % (rnd, letter)}
    for wl, wunits in (("cfg1", 2800000), ("cfg4", 1400000)):            # (the secondary workloads' own PMC passes, at bench.py's sizes)
        pw = os.path.join(src, "pmc_traffic_%s.json" % wl)
        if os.path.exists(pw):
            out[wl] = {"kernels": json.load(open(pw)), "units": wunits, "source": "profiles/%s_%s_pmc_traffic_%s.txt" % (rnd, letter, wl)}
    json.dump(out, open(os.path.join(dst, "%s_pmc_traffic.json" % rnd), "w"), indent=1, sort_keys=True)
    print("profiles/%s_pmc_traffic.json" % rnd)
