#!/usr/bin/env python3
"""Is a kernel's time the same when the GPU idles between steps?  (Round 5: after the decoder's main chain lost 0.6 ms of light kernels, k_dec_emit3 - unchanged in what it does - measured
0.2 ms slower inside the bench loop.  A power / clock effect of the denser loop, or the kernel?)  Decode steps of the headline workload back to back and with a pause between them; the
stage times come from the library's HIP events.  usage: [RFQ_HIP_LIBRARY=...] python tools/emit_gap.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import bench
from repaq_amd import RfqCodec
dev = torch.device("cuda:0"); codec = RfqCodec(device=0)
w = bench.Workload(codec, dev, "cfg2", 0, None, 1000, decode=True)
r = w.step(False)
for gap in (0.0, 0.03, 0.0, 0.03):
    acc = {}; n = 12
    for i in range(n + 2):
        d = codec.decode(r.d_rfq, r.rfq_len, split_pe=True, d_out1=w.o1.data_ptr(), cap1=w.n1 + 64, d_out2=w.o2.data_ptr(), cap2=w.n2 + 64, chunk_off=r.h_chunk_off, n_chunks=r.n_chunks)
        if i >= 2:
            for k, v in codec.timings(): acc[k] = acc.get(k, 0.0) + v
        if gap: torch.cuda.synchronize(); time.sleep(gap)
    print("%s gap %2.0f ms: " % (os.environ.get("RFQ_HIP_LIBRARY", "default"), gap * 1e3) + "  ".join("%s %.3f" % (k, v / n) for k, v in acc.items()))
