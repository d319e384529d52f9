"""Multi-GPU plumbing: one process per GPU, no data-path collective (chunks are independent once the header exists).

`torch.distributed` is used for exactly three things: the barrier around the timed region, the max-over-ranks time and the
sum of bytes — plus `share_header` and `plan_shares`, the tiny host exchanges a chunk-parallel encode of a SINGLE file needs (rank 0
makes the header from chunk 0, every other rank sets it: RfqCodec::setHeader, src/rfqcodec.cpp:16-18; where every rank's first chunk
starts).  Backend "gloo" everywhere - on the GPU box too: these are a few hundred bytes of host data, there is no data-path collective
and no RCCL traffic."""
import os
import time


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend="gloo", device=None):
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


def _plan_chain(codec, rank, world, d1, share1, avail1, d2, share2, avail2, paired, chunk_bases):
    """The plan as a chain: rank r scans from its own first cut over its share plus the head of the next and hands the chunk boundary that falls
    at or behind its share's end to rank r + 1 - N dependent scans, one small broadcast each.  Works for any input."""
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
                if nxt[0] < 0 or nxt[1] < 0:
                    raise RuntimeError("rank %d: the shares of the two files are not aligned record for record (chunk end %d / %d against shares of %d / %d bytes)" % (rank, e1[k], e2[k] if two else 0, share1 - cut[0], share2 - cut[1]))
            else:
                own = [n1, n2]
        if src < world - 1:
            if world > 1:
                dist.broadcast(nxt, src=src)
            if rank == src + 1:
                cut = [int(nxt[0]), int(nxt[1])]
    return cut[0], cut[1], own[0], own[1]


def plan_shares(codec, rank, world, d1, share1, avail1, d2=None, share2=0, avail2=0, paired=0, chunk_bases=1_000_000, stats=None):
    """The distributed plan of a chunk-parallel encode of ONE input whose text is spread over the ranks.

    Rank r holds, resident in its HBM, its own share of every stream (`share*` bytes: whole records, the logical file is the shares in
    rank order) followed by the first bytes of the next rank's share (`avail*` >= `share*` bytes readable in all; the last rank has
    none).  Chunks are cut greedily from the start of the file (Repaq::compress, src/repaq.cpp:546-553), so where rank r's first chunk
    starts depends on every earlier share - but only through the bases the chunk that is open at the share's start has already taken.

    When all cut units (reads / pairs) of the whole input have the same number of bases - sequencer output, as a rule - that carry follows
    from the NUMBER of units in front of the share: every rank scans its own share once (in parallel), the unit counts are all-gathered, and
    a second scan with the carry (rfq_encode_args.carry_bases) gives the true chunk ends: two scans per rank whatever the number of ranks.
    Otherwise the ranks scan one after the other (_plan_chain).  `stats` (a dict) receives which plan ran and what it took.

    Returns (cut1, cut2, len1, len2): this rank encodes bytes [cut, cut + len) of its resident buffers - with flush_all on every rank
    but the last (its range ends on a chunk boundary), final on the last."""
    import torch.distributed as dist
    t0 = time.perf_counter()
    two = paired == 1
    last = rank == world - 1
    if world == 1:
        if stats is not None:
            stats.update(plan="single", plan_ms=0.0)
        return 0, 0, share1, share2
    # pass 1 (every rank at once): my share alone, to its end: how many units, and whether they all hold the same number of bases
    r, _, _ = codec.scan(d1, share1, d2 if two else None, share2 if two else 0, paired, chunk_bases, final=True)
    upr = 1 if paired == 0 else 2
    mine = (int(r.n_reads) // upr, int(r.unit_bases), int(r.input_ended))
    allv = [None] * world
    dist.all_gather_object(allv, mine)
    lens = {v[1] for v in allv if v[0]}
    uniform = len(lens) == 1 and 0 not in lens and not any(v[2] for v in allv)
    if not uniform:
        out = _plan_chain(codec, rank, world, d1, share1, avail1, d2, share2, avail2, paired, chunk_bases)
        if stats is not None:
            stats.update(plan="chain", plan_ms=(time.perf_counter() - t0) * 1e3)
        return out
    L = lens.pop(); K = -(-chunk_bases // L)                                   # units per chunk
    before = sum(v[0] for v in allv[:rank])
    carry_units = before % K                                                  # units the open chunk already holds
    if last:
        c1 = c2 = 0
        if carry_units:
            r, e1, e2 = codec.scan(d1, share1, d2 if two else None, share2 if two else 0, paired, chunk_bases, final=True, carry_bases=carry_units * L)
            c1, c2 = e1[0], (e2[0] if two else 0)
        out = (c1, c2, share1 - c1, (share2 - c2) if two else 0)
    else:
        r, e1, e2 = codec.scan(d1, avail1, d2 if two else None, avail2 if two else 0, paired, chunk_bases, final=False, carry_bases=carry_units * L)
        c1, c2 = (e1[0], e2[0] if two else 0) if carry_units else (0, 0)       # the chunk open at my start is the previous rank's
        k = next((i for i, e in enumerate(e1) if e >= share1), None)
        if k is None:
            raise RuntimeError("rank %d: the head of the next share (%d bytes) does not reach the end of the chunk that straddles the share boundary" % (rank, avail1 - share1))
        out = (c1, c2, e1[k] - c1, (e2[k] - c2) if two else 0)
    if stats is not None:
        stats.update(plan="parallel", plan_ms=(time.perf_counter() - t0) * 1e3)
    return out


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
