"""Ablation of k_gather on the bench workload (GPU box): RFQ_TUNE bits 16 = no counting, 32 = no stores, 128 = no compose.  The image is invalid
while a bit is set (the call may even report a corrupt stream): only the gather stage's HIP-event time is read.  usage: python tools/abl_gather.py"""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch, _oracle as O
from repaq_amd import RfqCodec, PE_TWO_FILES
a, b = O.gen_np(O.NOVA_PE150, 5600000, seed=3)
t1 = torch.from_numpy(a).cuda(); t2 = torch.from_numpy(b).cuda()
c = RfqCodec(device=0)
best = None
for it in range(3):
    c.clearHeader()
    try:
        c.encode(t1.data_ptr(), t1.numel(), t2.data_ptr(), t2.numel(), PE_TWO_FILES, 1000000)
    except Exception as e:
        pass
    tm = dict(c.timings())
    g = tm.get("gather"); best = g if best is None or (g and g < best) else best
print("RESULT", os.environ.get("RFQ_TUNE", "0"), best, {k: round(v, 3) for k, v in tm.items()})
''' % (ROOT, ROOT)
for tune in (0, 16, 32, 48, 128, 176):
    env = dict(os.environ, RFQ_TUNE=str(tune))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    print([l for l in r.stdout.splitlines() if l.startswith("RESULT")] or r.stderr[-300:])
