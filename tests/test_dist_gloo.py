"""The N > 1 path on CPU: world_size 2, gloo.  Each rank codes its contiguous shard of chunks (with the
oracle standing in for the GPU here -- this test is about the sharding / state hand-over / gather / archive assembly
logic, not the codec), the block stream is gathered to rank 0, rank 0 assembles header + blocks + footer, and the file
must be the one the unmodified reference wrote with `dsrc c -t1` (md5 committed in tests/golden/state_golden.json) -- on
data whose blocks depend on the state the reference carries from block to block."""
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
    side = dist.new_group(backend="gloo")
    from dsrc_amd import synth
    from dsrc_amd.dist import gather_block_stream, shard_range
    from tests._oracle import Config, Oracle
    import hashlib
    import json
    from dsrc_amd.dist import archive_bytes, exchange_fields_capacity
    from tests.cases import state_dependent_fastq
    o = Oracle()
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "state_golden.json")))
    data = state_dependent_fastq()
    assert hashlib.sha256(data).hexdigest() == G["in_sha256"]
    chunks = [data[s: s + n] for s, n in o.cut_chunks(data, 1 << 20)]
    lo, hi = shard_range(len(chunks), rank, world)
    ok = []
    for a in G["archives"]:
        crc = "-c" in a["flags"]
        d = int(a["flags"][0][2:]); ql = int(a["flags"][1][2:])
        cfg = Config.from_levels(d, ql, False, crc)
        # state hand-over: one number per rank, from the first title of each chunk (number of fields = separators + 1)
        nf = [sum(c[: c.index(b"\n")].count(x) for x in b" ._,=:/-#") + 1 for c in chunks[lo:hi]]
        seed = exchange_fields_capacity(nf)
        blocks = [b for b, _, _ in o.compress_blocks_state(cfg, chunks[lo:hi], fields_cap=seed)]
        payload = torch.frombuffer(bytearray(b"".join(blocks)), dtype=torch.uint8) if blocks else torch.zeros(0, dtype=torch.uint8)
        res = gather_block_stream([len(b) for b in blocks], payload)
        # the same with the footer table on a host-side group of its own (what bench.py --gpus N does next to nccl)
        res2 = gather_block_stream([len(b) for b in blocks], payload, size_group=side)
        if rank == 0:
            assert res2[0] == res[0] and [bytes(t.numpy().tobytes()) for t in res2[1]] == [bytes(t.numpy().tobytes()) for t in res[1]]
            sizes, bufs = res
            arc = archive_bytes(sizes, [bytes(t.numpy().tobytes()) for t in bufs], dna_order=cfg.dna_order, quality_order=cfg.quality_order,
                                lossy=False, crc=crc, tag_flags=0, quality_offset=33, plus_repetition=False, color_space=False)
            ok.append((len(arc), hashlib.md5(arc).hexdigest()) == (a["size"], a["md5"]))
            # the test is sensitive: without the hand-over rank 1 starts from an empty history and the archive differs
            if world > 1 and not crc:
                unseeded = [b for b, _, _ in o.compress_blocks_state(cfg, chunks[shard_range(len(chunks), 1, world)[0]:])]
                ok.append(b"".join(unseeded) != b"".join(bytes(t.numpy().tobytes()) for t in bufs[1:]))
    if rank == 0:
        q.put(tuple(ok))
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
    assert ok and all(ok), ok
