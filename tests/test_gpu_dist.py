"""Multi-GPU path with the GPU codec in it, on the one GPU a test box has:
  * bench.py's N > 1 code path (one rank, nccl): the block stream rank 0 gathers in a step, assembled with dist.archive_bytes,
    is the archive `dsrc-amd c` writes for the same records;
  * two processes (gloo for the two tiny exchanges, the GPU for the blocks): the state hand-over through
    dsrcgpu_title_fields / exchange_fields_capacity / dsrcgpu_set_fields_capacity gives the archive the unmodified reference
    wrote with `dsrc c -t1` (tests/golden/state_golden.json) on data whose blocks depend on that state.
Reference: DsrcFileWriter (src/DsrcFile.cpp:112-170), the worker pool's block order (src/DsrcIo.cpp:25-66)."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "dsrc_amd", "csrc", "dsrc-amd")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


@pytest.mark.timeout(600)
def test_bench_dist_path_gathers_the_cli_archive(tmp_path):
    env = dict(os.environ, DSRC_BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--pipeline", "1", "--blocks", "6", "--steps", "1", "--warmup", "0",
                          "--no-cpu", "--decode-blocks", "0", "--check", "0", "--dump-step", str(tmp_path)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=550)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([x for x in out.stdout.splitlines() if x.startswith("{")][-1])
    assert line["config"]["gather_verified"] is True
    assert line["config"]["per_rank_MB_per_s"] and "one per GPU" in line["config"]["parallelism"]
    cli = str(tmp_path / "cli.dsrc")
    subprocess.check_call([CLI, "c", "-d3", "-q2", str(tmp_path / "step.fastq"), cli])
    assert _md5(str(tmp_path / "gathered.dsrc")) == _md5(cli)
    # two timed steps: the gather of step s runs while step s + 1 is being compressed into the other output buffer (StepGates);
    # rank 0's table and every rank's digest of its own stream are checked after the timed region
    env["MASTER_PORT"] = str(_free_port())
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--pipeline", "2", "--blocks", "8", "--steps", "2", "--warmup", "1",
                          "--no-cpu", "--decode-blocks", "0", "--check", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=550)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([x for x in out.stdout.splitlines() if x.startswith("{")][-1])
    assert line["config"]["gather_verified"] is True and line["steps"] == 2 and "1 ranks in the RCCL group" in line["config"]["parallelism"]


@pytest.mark.timeout(900)
def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks itself.  On the one GPU of a test box both
    ranks sit on GPU 0 (DSRC_BENCH_SAME_GPU: gloo and host copies -- RCCL refuses two ranks on one device); everything else is the
    N > 1 path of the driver's run: per-instance gather threads and groups, the state hand-over, the checked gather, one line from rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DSRC_BENCH_FORCE_DIST")}
    env["DSRC_BENCH_SAME_GPU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pipeline", "2", "--blocks", "8", "--steps", "2", "--warmup", "1",
                          "--no-cpu", "--decode-blocks", "0", "--check", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=850)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [x for x in out.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2
    assert "2 ranks in the" in line["config"]["parallelism"] and line["config"]["gather_verified"] is True
    assert len(line["config"]["per_rank_MB_per_s"]) == 2
    # a group of another size than --gpus is refused, not silently timed
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu"], env=env2, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dsrc_amd._lib import Handle, load
    from dsrc_amd.dist import archive_bytes, exchange_fields_capacity, gather_block_stream, shard_range
    from tests._oracle import Config, Oracle
    from tests.cases import state_dependent_fastq
    L = load()
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "state_golden.json")))
    data = state_dependent_fastq()
    chunks = [data[s: s + n] for s, n in Oracle().cut_chunks(data, 1 << 20)]          # the chunk cutter only (host row a-19)
    lo, hi = shard_range(len(chunks), rank, world)
    ok = []
    for a in G["archives"]:
        crc = "-c" in a["flags"]
        cfg = Config.from_levels(int(a["flags"][0][2:]), int(a["flags"][1][2:]), False, crc)
        nf = []
        for c in chunks[lo:hi]:
            t = c[: c.index(b"\n")]
            nf.append(L.dsrcgpu_title_fields(t, len(t), 0))
        seed = exchange_fields_capacity(nf)
        h = Handle(cfg.dna_order, cfg.quality_order, crc=crc)
        h.set_fields_capacity(seed)
        blocks = [r[0] for r in h.compress_batch(chunks[lo:hi])]
        h.close()
        payload = torch.frombuffer(bytearray(b"".join(blocks)), dtype=torch.uint8) if blocks else torch.zeros(0, dtype=torch.uint8)
        res = gather_block_stream([len(b) for b in blocks], payload)
        if rank == 0:
            sizes, bufs = res
            arc = archive_bytes(sizes, [bytes(t.numpy().tobytes()) for t in bufs], dna_order=cfg.dna_order, quality_order=cfg.quality_order,
                                lossy=False, crc=crc, tag_flags=0, quality_offset=33, plus_repetition=False, color_space=False)
            ok.append((len(arc), hashlib.md5(arc).hexdigest()) == (a["size"], a["md5"]))
    # the hand-over matters on this data: rank 1 without its seed writes other blocks
    seed = exchange_fields_capacity(nf)
    if rank == 1:
        cfg = Config.from_levels(0, 0)
        h = Handle(cfg.dna_order, cfg.quality_order); a_ = [r[0] for r in h.compress_batch(chunks[lo:hi])]; h.close()
        h = Handle(cfg.dna_order, cfg.quality_order); h.set_fields_capacity(seed); b_ = [r[0] for r in h.compress_batch(chunks[lo:hi])]; h.close()
        q.put(("differs", seed != 0 and a_ != b_))
    else:
        q.put(("archives", tuple(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_the_gpu_write_the_t1_archive():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=500) for _ in range(2))
    for p in procs:
        p.join(60)
    assert got["archives"] and all(got["archives"]), got
    assert got["differs"], "state_dependent_fastq no longer depends on the handed-over state"
