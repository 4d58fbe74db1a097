"""Multi-GPU sharding of the block path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU test-suite).

Blocks are independent given the chunk boundaries, so rank r compresses a contiguous range of partIds with
no peer traffic.  The only exchange is the one the archive needs: the per-block sizes (they *are* the footer
table, reference src/DsrcFile.cpp:142) are all-gathered, then every rank's compressed stream moves to rank 0
over its direct link (point-to-point send/recv, no ring, no reduction).  Rank-major order is archive order.

What a shard needs from the shards before it is one number: the capacity of the reference's TagStats::fields vector after
all earlier chunks, which is a fold over the first title of every chunk
(dsrc_amd._lib.fields_capacity_fold).  Every rank folds its own chunks, the per-rank results are all-gathered and rank r
seeds its first scheduler instance with the fold of ranks 0..r-1 (exchange_fields_capacity) -- with that the gathered
stream is byte for byte the archive `dsrc c -t1` writes.
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


def gather_block_stream(block_sizes: List[int], payload: torch.Tensor, group=None,
                        recv_bufs: Optional[List[torch.Tensor]] = None, size_group=None) -> Optional[Tuple[List[int], List[torch.Tensor]]]:
    """Gather every rank's (block_sizes, concatenated blocks) to rank 0.

    payload: 1-D uint8 tensor holding this rank's blocks back to back (on the device for nccl).
    recv_bufs: optional, rank 0 only: one pre-allocated uint8 tensor per rank (entry 0 unused) that the peers' streams are
    received into -- a caller that gathers inside a timed loop allocates them once.
    size_group: optional host-side group (gloo) for the footer table.  The block sizes are host integers on every rank (the batch
    call returned them); through the device group they travel as tensors and come back with a device synchronisation per
    all-gather (`int()`, `.tolist()`), which in a loop that overlaps the gather with the next step's compression stalls the
    thread behind kernels it has nothing to do with.  With a host group the table is a few KB over TCP and the device group carries
    the payloads only.
    Returns on rank 0: (all block sizes in archive order, [payload tensor of rank 0, 1, ...]); None elsewhere.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = payload.device
    total = int(sum(block_sizes))
    assert payload.numel() >= total
    # 1) footer table
    if size_group is not None:
        tables = [None] * world
        dist.all_gather_object(tables, [int(x) for x in block_sizes], group=size_group)
        counts = [len(t) for t in tables]; totals = [sum(t) for t in tables]
        allsz = tables
    else:
        # counts, then sizes padded to the largest count
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
        allsz = [t[: counts[r]].tolist() for r, t in enumerate(allsz)]
    # 2) payloads: direct send to rank 0
    if rank == 0:
        if recv_bufs is not None:
            assert all(recv_bufs[r].numel() >= totals[r] for r in range(1, world)), "receive buffer smaller than a peer's stream"
            bufs = [payload[:total]] + [recv_bufs[r][: totals[r]] for r in range(1, world)]
        else:
            bufs = [payload[:total]] + [torch.empty(totals[r], dtype=torch.uint8, device=dev) for r in range(1, world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r, group) for r in range(1, world) if totals[r] > 0]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        sizes: List[int] = []
        for r in range(world):
            sizes += [int(x) for x in allsz[r][: counts[r]]]
        return sizes, bufs
    if total > 0:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, payload[:total].contiguous(), 0, group)]):
            w.wait()
    return None


def exchange_fields_capacity(chunk_first_fields: List[int], group=None, device=None) -> int:
    """Seed for this rank's first scheduler instance.  chunk_first_fields: for every chunk of this rank, in order, the
    number of fields in the title of its first record (dsrcgpu_title_fields).  The capacity after a chunk is
    f(cap, n) = cap doubled (from 1) until it exceeds n - 1, so folding f over any run of chunks only needs the run's
    largest n: one int64 per rank is all-gathered and rank r folds the entries of ranks 0..r-1."""
    world = dist.get_world_size(group); rank = dist.get_rank(group)
    mine = torch.tensor([max(chunk_first_fields) if chunk_first_fields else 0], dtype=torch.int64, device=device)
    allv = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    cap = 0
    for r in range(rank):
        cap = _capacity_after(cap, int(allv[r][0]))
    return cap


def _capacity_after(cap: int, n_fields: int) -> int:
    for i in range(n_fields):
        if i == cap:
            cap = cap * 2 if cap else 1
    return cap


def archive_bytes(block_sizes: List[int], payloads: List[bytes], *, dna_order: int, quality_order: int, lossy: bool, crc: bool,
                  tag_flags: int, quality_offset: int, plus_repetition: bool, color_space: bool) -> bytes:
    """Rank 0: header + gathered blocks + footer = the .dsrc file (DsrcFileWriter, reference src/DsrcFile.cpp:112-170).
    Header: AA 02 00 02, footer size (BE32), footer offset (BE64), records (0), block count (BE64), AA x 8.
    Footer: CC, block sizes (host-endian uint32 array), dataset flags, quality offset, compression flags, orders, -f mask."""
    import struct
    body = b"".join(payloads)
    assert len(body) == sum(block_sizes)
    foot = b"\xCC" + struct.pack("<%dI" % len(block_sizes), *block_sizes)
    foot += bytes([(2 if color_space else 0) | (1 if plus_repetition else 0), quality_offset,
                   (1 if lossy else 0) | (2 if crc else 0), dna_order, quality_order]) + struct.pack(">Q", tag_flags)
    head = b"\xAA\x02\x00\x02" + struct.pack(">IQQQ", len(foot), 40 + len(body), 0, len(block_sizes)) + b"\xAA" * 8
    return head + body + foot
