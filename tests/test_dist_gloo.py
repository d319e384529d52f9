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
    # the plan pass (rfq_scan_batch): every rank finds where each chunk ends in the text, then takes its range of chunks
    d = codec.dev_put(fq1)
    r, ends, _ = codec.scan(d, len(fq1), None, 0, O.SE, cb, final=True)
    codec.dev_free(d)
    n_chunks = r.n_chunks
    assert n_chunks == len(O.chunk_table(want)) - 1
    ranges = D.split_chunk_ranges(n_chunks, world)
    b, e = ranges[rank]
    lo, hi = (ends[b - 1] if b else 0), (len(fq1) if e == n_chunks else ends[e - 1])
    part = fq1[lo:hi]
    def enc(emit_header):
        dp = codec.dev_put(part)
        rr = codec.encode(dp, len(part), None, 0, O.SE, cb, final=(e == n_chunks), emit_header=emit_header, file_off1=lo, flush_all=(e != n_chunks))
        out = codec.dev_get(rr.d_rfq, rr.rfq_len)
        codec.dev_free(dp)
        return out
    if rank == 0:                                    # rank 0 encodes its range first: that makes the header from chunk 0
        img = enc(True)
    hdr = D.share_header(codec)
    if rank != 0:
        img = enc(False)
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
