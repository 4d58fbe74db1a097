#!/usr/bin/env python3
"""bench.py -- raw FASTQ MB/s through the MI355X block-compression path (BASELINE.json metric).

A *step* = one scheduler pass (dsrcgpu_compress_batch_device) over `--blocks` consecutive 8 MiB
chunks of the synthetic 150 bp Illumina-like data set (BASELINE.json configs[2]: 100 M reads,
-d3 -q2, default -b8), generated on the device by the counter-based generator so that every rank /
step can produce its own shard.  Inputs are resident in HBM when the timed region starts and the
compressed blocks stay in HBM (PCIe is not in `value`).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); blocks are independent, so
ranks take disjoint record ranges (weak scaling, no collective in the data path) and only the
per-block sizes / the compressed stream are gathered to rank 0 after the timed region.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BUF = 8 << 20                     # -b8 (reference default, src/Common.h:156)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def title_len(i: np.ndarray) -> np.ndarray:
    def digits(v):
        d = np.ones(v.shape, dtype=np.int64)
        for k in range(1, 20):
            d += (v >= 10 ** k)
        return d
    x = 1000 + (7 * i) % 20000
    y = 2000 + (13 * i) % 90000
    return 8 + digits(i) + 26 + 1 + 1 + 4 + 1 + digits(x) + 1 + digits(y) + 13


def record_offsets(first: int, count: int) -> np.ndarray:
    i = np.arange(first, first + count, dtype=np.int64)
    size = title_len(i) + 1 + 150 + 1 + 1 + 1 + 150 + 1
    off = np.zeros(count + 1, dtype=np.int64)
    np.cumsum(size, out=off[1:])
    return off


def cut_blocks(off: np.ndarray, nblocks: int):
    """Chunk boundaries as IFastqStreamReader::ReadNextChunk would place them (reference
    src/FastqStream.cpp:18-98): a chunk ends just before the first record that starts after
    byte (start + buf - 8192); its size excludes the final newline."""
    starts, sizes, recs = [], [], []
    r = 0
    for _ in range(nblocks):
        start = off[r]
        pos = start + BUF - 8192
        nxt = int(np.searchsorted(off, pos, side="right"))
        if nxt >= len(off):
            break
        starts.append(int(start)); sizes.append(int(off[nxt] - start - 1)); recs.append(nxt - r)
        r = nxt
    return starts, sizes, recs, r


def cpu_baseline(sample: bytes, d: int, q: int):
    """Reference (oracle/_ref, unmodified DSRC built from /root/reference) multi-threaded on the host
    cores, or our C port on one core when _ref is absent.  Reported beside the GPU number only."""
    import tempfile
    from tests._oracle import Oracle, Ref, have_ref
    # the reference's queues use a 64-bit completion mask: thread counts >= 64 are undefined
    cores = min(os.cpu_count() or 1, 60)
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        src = os.path.join(td, "s.fastq"); dst = os.path.join(td, "s.dsrc")
        with open(src, "wb") as f:
            f.write(sample)
        if have_ref():
            r = Ref()
            t = time.time(); rc = r.compress_file(src, dst, d, q, False, False, 33, 8, cores); dt = time.time() - t
            kind = "reference"
        else:
            o = Oracle(); cores = 1
            t = time.time(); rc = o.compress_file(src, dst, d, q, False, False, 33, 8); dt = time.time() - t
            kind = "port"
        assert rc == 0
    return {"value": round(len(sample) / dt / 1e6, 2), "unit": "MB/s", "cores": cores, "kind": kind,
            "sample": f"{len(sample)} bytes of the same synthetic FASTQ ({len(sample) // BUF + 1} blocks), -d{d} -q{q} -b8, "
                      f"file in tmpfs, {cores} worker threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=int(os.environ.get("DSRC_BENCH_BLOCKS", "256")), help="8 MiB chunks per step per GPU")
    ap.add_argument("--dna", type=int, default=3)
    ap.add_argument("--qua", type=int, default=2)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--check", type=int, default=2, help="blocks of the first step to verify against the oracle")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_
        torch.cuda.set_device(local)
        dist_.init_process_group("nccl", device_id=torch.device("cuda", local))
        dist = dist_

    from dsrc_amd._lib import Handle
    from tests._oracle import Config
    cfg = Config.from_levels(args.dna, args.qua)
    h = Handle(cfg.dna_order, cfg.quality_order, quality_offset=33, device=local)

    recs_per_block = 22300
    per_step_recs = int(args.blocks * recs_per_block * 1.02) + 1000
    total_steps = args.steps + args.warmup
    # rank r, step s covers records [base, base + per_step_recs)
    cap_in = per_step_recs * 384
    d_in = h.dev_alloc(cap_in)
    d_out = h.dev_alloc(cap_in // 2)

    def prepare(step):
        first = 1 + (rank * total_steps + step) * per_step_recs
        nbytes = h.synth_illumina(first, per_step_recs, d_in, cap_in)
        off = record_offsets(first, per_step_recs)
        assert off[-1] == nbytes, (off[-1], nbytes)
        starts, sizes, recs, _ = cut_blocks(off, args.blocks)
        assert len(starts) == args.blocks
        return first, starts, sizes, recs

    def barrier():
        if dist is not None:
            dist.barrier()

    def run(step_info):
        _, starts, sizes, _ = step_info
        return h.compress_batch_device(d_in, starts, sizes, d_out, cap_in // 2)

    def stage(step_info):
        pass

    infos = []
    t_kernel = []; t_rc = []
    in_bytes = 0; out_bytes = 0
    first_out = None
    for s in range(args.warmup):
        info = prepare(s); stage(info); res = run(info)
        if s == 0:
            first_out = (info, res, h.dev_download(d_out, res[0][min(args.check, len(res[0])) - 1] + res[1][min(args.check, len(res[0])) - 1]) if args.check else b"")
            first_chunks = [h.dev_download(d_in + info[1][i], info[2][i]) for i in range(min(args.check, args.blocks))]
    barrier()
    wall = 0.0
    for s in range(args.warmup, total_steps):
        info = prepare(s); stage(info)
        barrier()
        t0 = time.perf_counter()
        res = run(info)
        barrier()
        wall += time.perf_counter() - t0
        ms, rc_ms, nrc = h.last_timing()
        t_kernel.append(ms); t_rc.append(rc_ms)
        in_bytes += sum(info[2]) + len(info[2]); out_bytes += sum(res[1])
        if first_out is None and args.check:
            first_out = (info, res, h.dev_download(d_out, res[0][args.check - 1] + res[1][args.check - 1]))
            first_chunks = [h.dev_download(d_in + info[1][i], info[2][i]) for i in range(min(args.check, args.blocks))]

    # parity spot-check of the first step against the oracle (outside the timed region)
    checked = 0
    if rank == 0 and args.check and first_out is not None:
        from tests._oracle import Oracle
        o = Oracle()
        info, res, blob = first_out
        for i in range(min(args.check, args.blocks)):
            want = o.compress_block(cfg, first_chunks[i])[0]
            got = blob[res[0][i]: res[0][i] + res[1][i]]
            assert got == want, f"bench parity check failed on block {i}"
            checked += 1

    if dist is not None:
        import torch
        t = torch.tensor([wall, float(in_bytes), float(out_bytes)], device="cuda", dtype=torch.float64)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        wall = float(tmax[0]); in_bytes = float(tsum[1]); out_bytes = float(tsum[2])

    if rank == 0:
        value = in_bytes / wall / 1e6
        rc_avg = sum(t_rc) / max(1, len(t_rc))
        k_avg = sum(t_kernel) / max(1, len(t_kernel))
        per_rank_in = in_bytes / world / args.steps; per_rank_out = out_bytes / world / args.steps
        alg_bytes = per_rank_in + per_rank_out            # chunk read once + block written once
        achieved = alg_bytes / (k_avg / 1e3) / 1e9 if k_avg > 0 else 0.0
        line = {
            "metric": "raw FASTQ MB/s compressed (bit-identical .dsrc)", "value": round(value, 1), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u16/u32 integer", "data": "synthetic",
            "config": {"workload": f"Synthetic Illumina 150 bp FASTQ (BASELINE configs[2] shape: 100M-read set), -d{args.dna} -q{args.qua} -b8; "
                                   f"one step = {args.blocks} consecutive 8 MiB blocks per GPU, device-resident",
                       "blocks_per_step": args.blocks, "parallelism": f"blocks sharded over {world} GPU(s), no data-path collective",
                       "ratio_out_in": round(out_bytes / in_bytes, 4), "parity_checked_blocks": checked},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel": "whole batch (all kernels of one scheduler pass, HIP events on the scheduler stream)",
                         "batch_ms": round(k_avg, 2), "k_rc_ms": round(rc_avg, 2)},
        }
        if not args.no_cpu and world == 1:
            sample_blocks = 120
            from dsrc_amd import synth
            nrec = sample_blocks * 22000
            # the same generator on the host would take minutes in numpy; download the device bytes instead
            need = min(sample_blocks, args.blocks)
            info = prepare(0)
            sample = h.dev_download(d_in, info[1][need - 1] + info[2][need - 1] + 1)
            line["cpu_baseline"] = cpu_baseline(sample, args.dna, args.qua)
        print(json.dumps(line))
    h.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
