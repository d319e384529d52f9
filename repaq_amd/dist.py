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


def plan_shares(codec, rank, world, d1, share1, avail1, d2=None, share2=0, avail2=0, paired=0, chunk_bases=1_000_000):
    """The distributed plan of a chunk-parallel encode of ONE input whose text is spread over the ranks.

    Rank r holds, resident in its HBM, its own share of every stream (`share*` bytes: whole records, the logical file is the shares in
    rank order) followed by the first bytes of the next rank's share (`avail*` >= `share*` bytes readable in all; the last rank has
    none).  Chunks are cut greedily from the start of the file (Repaq::compress, src/repaq.cpp:546-553), so where rank r's first chunk
    starts depends on every earlier share: the ranks run the plan pass (rfq_scan_batch: line index, read lengths, cut rule - no
    coding) one after the other, each from its own first cut over its share plus the head of the next, and hand the offset of the chunk
    boundary that falls at or behind the share's end to the next rank.  One small host message per rank; no data-path collective.

    Returns (cut1, cut2, len1, len2): this rank encodes bytes [cut, cut + len) of its resident buffers - with flush_all on every rank
    but the last (its range ends on a chunk boundary), final on the last."""
    import torch
    import torch.distributed as dist
    two = paired == 1
    cut = [0, 0]; own = [share1, share2]
    for src in range(world):
        nxt = torch.zeros(2, dtype=torch.int64)
        if rank == src:
            last = src == world - 1
            n1 = (share1 if last else avail1) - cut[0]; n2 = ((share2 if last else avail2) - cut[1]) if two else 0
            if not last:
                r, e1, e2 = codec.scan(d1 + cut[0], n1, (d2 + cut[1]) if two else None, n2, paired, chunk_bases, final=False)
                k = next((i for i, e in enumerate(e1) if e >= share1 - cut[0]), None)
                if k is None:
                    raise RuntimeError("rank %d: the head of the next share (%d bytes) does not reach the end of the chunk that straddles the share boundary" % (rank, avail1 - share1))
                own = [e1[k], e2[k] if two else 0]
                nxt[0] = e1[k] - (share1 - cut[0]); nxt[1] = (e2[k] - (share2 - cut[1])) if two else 0
            else:
                own = [n1, n2]
        if src < world - 1:
            if world > 1:
                dist.broadcast(nxt, src=src)
            if rank == src + 1:
                cut = [int(nxt[0]), int(nxt[1])]
    return cut[0], cut[1], own[0], own[1]


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
