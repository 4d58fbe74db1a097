"""Multi-GPU sharding of the block path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU test-suite).

Blocks are independent given the chunk boundaries, so rank r compresses a contiguous range of partIds with
no peer traffic.  The only exchange is the one the archive needs: the per-block sizes (they *are* the footer
table, reference src/DsrcFile.cpp:142) are all-gathered, then every rank's compressed stream moves to rank 0
over its direct link (point-to-point send/recv, no ring, no reduction).  Rank-major order is archive order.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_parts: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous partId range [lo, hi) of `rank` (the first n_parts % world ranks get one extra part)."""
    base, extra = divmod(n_parts, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_block_stream(block_sizes: List[int], payload: torch.Tensor, group=None) -> Optional[Tuple[List[int], List[torch.Tensor]]]:
    """Gather every rank's (block_sizes, concatenated blocks) to rank 0.

    payload: 1-D uint8 tensor holding this rank's blocks back to back (on the device for nccl).
    Returns on rank 0: (all block sizes in archive order, [payload tensor of rank 0, 1, ...]); None elsewhere.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = payload.device
    total = int(sum(block_sizes))
    assert payload.numel() >= total
    # 1) footer table: counts, then sizes padded to the largest count
    cnt = torch.tensor([len(block_sizes), total], dtype=torch.int64, device=dev)
    cnts = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c[0]) for c in cnts]; totals = [int(c[1]) for c in cnts]
    mx = max(counts) if counts else 0
    mine = torch.zeros(max(mx, 1), dtype=torch.int64, device=dev)
    if block_sizes:
        mine[: len(block_sizes)] = torch.tensor(block_sizes, dtype=torch.int64, device=dev)
    allsz = [torch.zeros(max(mx, 1), dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allsz, mine, group=group)
    # 2) payloads: direct send to rank 0
    if rank == 0:
        bufs = [payload[:total]] + [torch.empty(totals[r], dtype=torch.uint8, device=dev) for r in range(1, world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r, group) for r in range(1, world) if totals[r] > 0]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        sizes: List[int] = []
        for r in range(world):
            sizes += [int(x) for x in allsz[r][: counts[r]].tolist()]
        return sizes, bufs
    if total > 0:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, payload[:total].contiguous(), 0, group)]):
            w.wait()
    return None
