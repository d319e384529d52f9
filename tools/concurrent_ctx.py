#!/usr/bin/env python3
"""Do two independent batches on two contexts of ONE GPU finish sooner side by side than one after the other?  (Round 4 found that two kernels of this path on two
streams take the sum of their times when both are VALU-bound or both memory-bound; an index pass - memory-bound - beside a gather - VALU-bound - was never timed.)
Two halves of the PE150 workload, each on its own context / host thread: sequential vs concurrent, encode and decode.  usage: python tools/concurrent_ctx.py [pairs_per_half]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import _oracle as O
from repaq_amd import RfqCodec, PE_TWO_FILES

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 5_600_000
dev = torch.device("cuda:0")
halves = []
for h in range(2):
    a1, a2 = O.gen_np(1, pairs, seed=3 + h)
    c = RfqCodec(device=0)
    t1, t2 = torch.from_numpy(a1).to(dev), torch.from_numpy(a2).to(dev)
    o1, o2 = torch.empty(a1.size + 64, dtype=torch.uint8, device=dev), torch.empty(a2.size + 64, dtype=torch.uint8, device=dev)
    halves.append(dict(c=c, t1=t1, t2=t2, o1=o1, o2=o2, n1=int(a1.size), n2=int(a2.size), r=None))
nbytes = sum(h["n1"] + h["n2"] for h in halves)


def enc(h):
    h["c"].clearHeader()
    h["r"] = h["c"].encode(h["t1"].data_ptr(), h["n1"], h["t2"].data_ptr(), h["n2"], PE_TWO_FILES, 1_000_000)


def dec(h):
    r = h["r"]
    h["c"].decode(r.d_rfq, r.rfq_len, split_pe=True, d_out1=h["o1"].data_ptr(), cap1=h["n1"] + 64, d_out2=h["o2"].data_ptr(), cap2=h["n2"] + 64, chunk_off=r.h_chunk_off, n_chunks=r.n_chunks)


def timed(fn, concurrent, reps=8):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if concurrent:
            th = [threading.Thread(target=fn, args=(h,)) for h in halves]
            for t in th: t.start()
            for t in th: t.join()
        else:
            for h in halves: fn(h)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


for h in halves: enc(h); dec(h)
for name, fn in (("encode", enc), ("decode", dec)):
    s = timed(fn, False); c = timed(fn, True)
    print("%s: 2 x %.2f GB  sequential %.2f ms (%.0f GB/s)  concurrent %.2f ms (%.0f GB/s)  speed-up %.3f" % (name, nbytes / 2e9, s * 1e3, nbytes / s / 1e9, c * 1e3, nbytes / c / 1e9, s / c))
# staggered: the second context starts when the first is about a third through (its index and cut are done) - what a two-slice pipeline inside one call would do
for name, fn, lag in (("encode", enc, 0.0022), ("decode", dec, 0.0012)):
    best = 1e9
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a = threading.Thread(target=fn, args=(halves[0],)); b = threading.Thread(target=fn, args=(halves[1],))
        a.start(); time.sleep(lag); b.start(); a.join(); b.join()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%s staggered by %.1f ms: %.2f ms (%.0f GB/s)" % (name, lag * 1e3, best * 1e3, nbytes / best / 1e9))
