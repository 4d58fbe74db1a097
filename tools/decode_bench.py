#!/usr/bin/env python3
"""Decompression alone on the GPU: `distinct` synthetic 8 MiB chunks are compressed once, then `blocks` blocks (the distinct ones
repeated) are decoded in one dsrcgpu_decompress_batch_device pass, everything resident in HBM.  Prints one JSON line per pass.
Used to take rocprofv3 kernel traces / PMC counters of the decoder without the compression bench around it:

    python tools/decode_bench.py --blocks 2400 --distinct 300 -d 3 -q 2 --passes 2
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

import bench  # noqa: E402  (record geometry of the synthetic set)
from dsrc_amd._lib import Handle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=2400)
    ap.add_argument("--distinct", type=int, default=300)
    ap.add_argument("-d", type=int, default=3)
    ap.add_argument("-q", type=int, default=2)
    ap.add_argument("--lossy", action="store_true")
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--check", type=int, default=2, help="decoded blocks compared with the chunk text")
    ap.add_argument("--inst", type=int, default=1, help="decoding instances run concurrently (threads, one handle each)")
    ap.add_argument("--stagger", type=float, default=1.5, help="seconds between the starts of the instances")
    a = ap.parse_args()

    h = Handle(3 * a.d, a.q, lossy=a.lossy, quality_offset=33)
    n = a.distinct
    recs = int(n * bench.RECS_PER_BLOCK * 1.02) + 1000
    cap_in = recs * 384
    d_in = h.dev_alloc(cap_in)
    nbytes = h.synth_illumina(1, recs, d_in, cap_in)
    off = bench.record_offsets(1, recs)
    assert off[-1] == nbytes
    starts, sizes = bench.cut_blocks(off, n)
    cap_out = cap_in // 2
    d_blk = h.dev_alloc(cap_out)
    t0 = time.perf_counter()
    o_offs, o_sizes, _, _ = h.compress_batch_device(d_in, starts, sizes, d_blk, cap_out)
    t_comp = time.perf_counter() - t0
    reps = (a.blocks + n - 1) // n
    offs = (list(o_offs) * reps)[:a.blocks]; szs = (list(o_sizes) * reps)[:a.blocks]
    text_bytes = int(sum((list(sizes) * reps)[:a.blocks]) + a.blocks)
    d_txt = h.dev_alloc(text_bytes + 4096)
    print(json.dumps({"compress_s": round(t_comp, 3), "distinct": n, "in_bytes": int(sum(sizes)), "block_bytes": int(sum(o_sizes))}), flush=True)
    for p in range(a.passes):
        t0 = time.perf_counter()
        t_offs, t_sizes = h.decompress_batch_device(d_blk, offs, szs, d_txt, text_bytes + 4096, verify=False)[:2]
        dt = time.perf_counter() - t0
        assert sum(t_sizes) == text_bytes, (sum(t_sizes), text_bytes)
        print(json.dumps({"pass": p, "blocks": a.blocks, "text_bytes": text_bytes, "s": round(dt, 4), "gpu_ms": round(h.last_timing()[0], 1),
                          "MB_per_s": round(text_bytes / dt / 1e6, 1)}), flush=True)
    if a.inst > 1:
        import threading
        hs = [h] + [Handle(3 * a.d, a.q, lossy=a.lossy, quality_offset=33) for _ in range(a.inst - 1)]
        txts = [d_txt] + [x.dev_alloc(text_bytes + 4096) for x in hs[1:]]
        for x, t in zip(hs[1:], txts[1:]):
            x.decompress_batch_device(d_blk, offs, szs, t, text_bytes + 4096, verify=False)       # warm-up: arena, table region

        def work(i):
            for _ in range(a.passes):
                hs[i].decompress_batch_device(d_blk, offs, szs, txts[i], text_bytes + 4096, verify=False)
        ths = [threading.Thread(target=work, args=(i,)) for i in range(a.inst)]
        t0 = time.perf_counter()
        for i, t in enumerate(ths):
            t.start()
            if i + 1 < a.inst:
                time.sleep(a.stagger)
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        print(json.dumps({"instances": a.inst, "passes_each": a.passes, "blocks": a.blocks, "s": round(dt, 3),
                          "MB_per_s": round(text_bytes * a.inst * a.passes / dt / 1e6, 1)}), flush=True)
        for x, t in zip(hs[1:], txts[1:]):
            x.dev_free(t); x.close()
    step = max(1, a.blocks // max(1, a.check))
    for i in list(range(0, a.blocks, step))[:a.check] + [a.blocks - 1]:
        src = h.dev_download(d_in + int(starts[i % n]), int(sizes[i % n]))
        got = h.dev_download(d_txt + int(t_offs[i]), int(t_sizes[i]))
        assert got == src + b"\n", f"decoded text of block {i} differs from the chunk"
    print(json.dumps({"parity": "ok"}), flush=True)
    h.dev_free(d_txt); h.dev_free(d_blk); h.dev_free(d_in)
    h.close()


if __name__ == "__main__":
    main()
