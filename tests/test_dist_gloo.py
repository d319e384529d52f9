"""CPU, world_size 2, gloo: the N>1 plumbing — chunk-parallel encode of ONE file across ranks (header from rank 0, no data-path
collective) reassembles to the one-shot image, and the bench-style reductions agree.  Kernels run on the SIMT-emulation test
library here; on the GPU box the same code path runs under the nccl (= RCCL) backend in bench.py."""
import os
import socket
import subprocess
import sys
import textwrap

import _engine as E

WORKER = textwrap.dedent('''
    import os, sys, pickle
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import torch
    import _oracle as O
    from repaq_amd import RfqCodec, dist as D
    rank, world = D.init("gloo")
    codec = RfqCodec(device=0, library=%(lib)r)
    fq1, _ = O.gen(O.NOVA_SE150, 900, seed=77)
    cb = 15000
    want = O.encode_file(fq1, b"", O.SE, cb)
    offs = O.chunk_table(want)                       # chunk boundaries in the image (the oracle stands in for the plan pass here)
    n_chunks = len(offs) - 1
    # record ranges of each chunk: a chunk ends after the read that reaches cb bases (150-base reads -> 100 reads per chunk)
    rpc = (cb + 149) // 150
    ranges = D.split_chunk_ranges(n_chunks, world)
    b, e = ranges[rank]
    rec_bytes = [i for i, ch in enumerate(fq1) if ch == 10]
    def rec_off(r):                                  # byte offset of record r
        return 0 if r == 0 else rec_bytes[4 * r - 1] + 1
    total_recs = len(rec_bytes) // 4
    if rank == 0:                                    # rank 0 encodes its range first: that makes the header from chunk 0
        part = fq1[rec_off(b * rpc): rec_off(min(e * rpc, total_recs))]
        img = codec.encode_bytes(part, b"", O.SE, cb, emit_header=True)
    hdr = D.share_header(codec)
    if rank != 0:
        part = fq1[rec_off(b * rpc): rec_off(min(e * rpc, total_recs))]
        img = codec.encode_bytes(part, b"", O.SE, cb, emit_header=False)
    gathered = [None] * world
    torch.distributed.all_gather_object(gathered, img)
    tmax, bsum = D.reduce_max_sum(0.5 + rank, len(part))
    if rank == 0:
        assert b"".join(gathered) == want, "chunk-parallel image differs from the one-shot image"
        assert hdr == want[:len(hdr)]
        assert tmax == 0.5 + world - 1 and bsum == len(fq1)
        print("DIST_OK", n_chunks, world)
    torch.distributed.destroy_process_group()
''')


def test_two_rank_chunk_parallel_encode(tmp_path):
    E.build_emu()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": E.ROOT, "lib": E.EMU_LIB})
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIP_EMU_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(outs)
    assert "DIST_OK" in outs[0]
