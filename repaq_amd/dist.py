"""Multi-GPU plumbing: one process per GPU, no data-path collective (chunks are independent once the header exists).

`torch.distributed` is used for exactly three things: the barrier around the timed region, the max-over-ranks time and the
sum of bytes — plus `share_header`, the one tiny exchange a chunk-parallel encode of a SINGLE file needs (rank 0 makes the
header from chunk 0, every other rank sets it; RfqCodec::setHeader, src/rfqcodec.cpp:16-18).  Backend "nccl" (= RCCL) on GPUs,
"gloo" in the CPU tests."""
import os
import time


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend="nccl", device=None):
    import torch.distributed as dist
    rank, world, _ = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_max_sum(seconds, nbytes, device=None):
    """(max over ranks of seconds, sum over ranks of nbytes)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds), float(nbytes)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    b = torch.tensor([float(nbytes)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(b, op=dist.ReduceOp.SUM)
    return float(t.item()), float(b.item())


def share_header(codec, device=None):
    """Rank 0 holds a header (made by its first encode); every other rank receives the <= 272 bytes and sets it."""
    import torch
    import torch.distributed as dist
    rank, world, _ = env_rank()
    if world == 1:
        return codec.header()
    buf = torch.zeros(17 + 255 + 1, dtype=torch.uint8, device=device)
    if rank == 0:
        h = codec.header()
        buf[0] = len(h) - 17
        buf[1:1 + len(h)] = torch.tensor(list(h), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=0)
    h = bytes(buf[1:1 + 17 + int(buf[0])].tolist())
    if rank != 0:
        codec.setHeader(h)
    return h


def split_chunk_ranges(n_chunks, world):
    """Contiguous, near-equal ranges of chunk indices: [(begin, end)] per rank (the host work queue in its static form)."""
    base, rem = divmod(n_chunks, world)
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < rem else 0)
        out.append((b, e)); b = e
    return out


def timed(fn, steps):
    """barrier + fn() x steps + barrier; returns this rank's seconds (callers add the device sync inside fn or around)."""
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier()
    return time.perf_counter() - t0
