"""Host <-> device link rate of the box (page-locked buffers, one stream / two streams): the ceiling of anything that is handed host buffers."""
import time
import torch
n = 1 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory(); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda"); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, fn in (("H2D 1 GiB", lambda: d.copy_(h, non_blocking=True)), ("D2H 1 GiB", lambda: h.copy_(d, non_blocking=True))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); print("%s: %.1f GB/s" % (name, 5 * n / (time.perf_counter() - t0) / 1e9))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize(); print("H2D + D2H at once: %.1f GB/s each way" % (5 * n / (time.perf_counter() - t0) / 1e9))
# 16 MB pieces back to back on one stream (the driver's staging blocks)
m = 16 << 20
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(0, n, m): d[i:i + m].copy_(h[i:i + m], non_blocking=True)
torch.cuda.synchronize(); print("H2D 64 x 16 MiB pieces: %.1f GB/s" % (n / (time.perf_counter() - t0) / 1e9))
