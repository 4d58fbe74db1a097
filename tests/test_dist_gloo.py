"""The N > 1 path on CPU: world_size 2, gloo.  Each rank codes its contiguous shard of chunks (with the
oracle standing in for the GPU here -- this test is about the sharding / gather logic, not the codec),
the block stream is gathered to rank 0 and must equal the single-process archive."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dsrc_amd import synth
    from dsrc_amd.dist import gather_block_stream, shard_range
    from tests._oracle import Config, Oracle
    o = Oracle(); cfg = Config.from_levels(1, 1)
    chunks = [synth.illumina_fastq(40, first=1 + 40 * k)[:-1] for k in range(5)]
    lo, hi = shard_range(len(chunks), rank, world)
    blocks = [o.compress_block(cfg, c)[0] for c in chunks[lo:hi]]
    payload = torch.frombuffer(bytearray(b"".join(blocks)), dtype=torch.uint8) if blocks else torch.zeros(0, dtype=torch.uint8)
    res = gather_block_stream([len(b) for b in blocks], payload)
    if rank == 0:
        sizes, bufs = res
        stream = b"".join(bytes(t.numpy().tobytes()) for t in bufs)
        want = [o.compress_block(cfg, c)[0] for c in chunks]
        q.put((sizes == [len(b) for b in want], stream == b"".join(want)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range():
    from dsrc_amd.dist import shard_range
    for n in (0, 1, 5, 8, 17):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


@pytest.mark.timeout(180)
def test_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=150)
    for p in procs:
        p.join(60)
    assert ok == (True, True)
