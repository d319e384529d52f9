import sys, os
ROOT=os.environ.get("GRAFT_REPO_ROOT",".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, _oracle as O
from repaq_amd import RfqCodec, PE_TWO_FILES
a, b = O.gen_np(O.NOVA_PE150, 5600000, seed=3)
t1 = torch.from_numpy(a).cuda(); t2 = torch.from_numpy(b).cuda()
c = RfqCodec(device=0)
best = None
for it in range(3):
    c.clearHeader()
    try: c.encode(t1.data_ptr(), t1.numel(), t2.data_ptr(), t2.numel(), PE_TWO_FILES, 1000000)
    except Exception as e: pass
    tm = dict(c.timings()); g = tm.get("read_table+cut"); best = g if best is None or (g and g < best) else best
print("RESULT", os.environ.get("RFQ_HIP_LIBRARY","default").split("/")[-1], best, {k: round(v, 3) for k, v in tm.items()})
