#!/usr/bin/env python3
"""PCIe-inclusive rate of the C ABI's host entry point (dsrcgpu_compress_batch: host chunks in, host blocks out)
and throughput of the other BASELINE configs.  Not the headline metric (bench.py); results go to profiles/."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dsrc_amd._lib import Handle  # noqa: E402
from dsrc_amd import synth  # noqa: E402
from tests._oracle import Config  # noqa: E402


def illumina_chunks(h, nblocks, first=1):
    recs = int(nblocks * bench.RECS_PER_BLOCK * 1.02) + 1000
    cap = recs * 384
    d = h.dev_alloc(cap)
    n = h.synth_illumina(first, recs, d, cap)
    off = bench.record_offsets(first, recs)
    starts, sizes = bench.cut_blocks(off, nblocks)
    data = h.dev_download(d, n)
    h.dev_free(d)
    return [data[s: s + z] for s, z in zip(starts, sizes)]


def run(name, cfg, chunks, reps=3):
    h = Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc)
    h.compress_batch(chunks)
    t = time.perf_counter()
    for _ in range(reps):
        res = h.compress_batch(chunks)
    dt = (time.perf_counter() - t) / reps
    ms, rc_ms, _ = h.last_timing()
    h.close()
    nin = sum(len(c) + 1 for c in chunks); nout = sum(len(r[0]) for r in res)
    print(json.dumps({"case": name, "blocks": len(chunks), "in_bytes": nin, "out_bytes": nout, "ratio": round(nout / nin, 4),
                      "host_to_host_MBps": round(nin / dt / 1e6, 1), "gpu_batch_ms": round(ms, 1), "k_rc_ms": round(rc_ms, 1)}))


def main():
    h = Handle()
    ill = illumina_chunks(h, 256)
    h.close()
    run("config2 illumina -d0 -q0 (host in/out)", Config.from_levels(0, 0), ill)
    run("config3 illumina -d3 -q2 (host in/out)", Config.from_levels(3, 2), ill)
    run("illumina -d3 -q2 -c (CRC32)", Config.from_levels(3, 2, crc=True), ill)
    run("illumina -d2 -q1 -l (lossy)", Config.from_levels(2, 1, lossy=True), ill)
    ion = synth.iontorrent_fastq(150000)
    from tests._oracle import Oracle
    cuts = Oracle().cut_chunks(ion, 8 << 20)
    chunks = [ion[s: s + z] for s, z in cuts]
    run("config5 454/Ion-Torrent-like -d2 -q1 -l", Config.from_levels(2, 1, lossy=True), chunks)


if __name__ == "__main__":
    main()
