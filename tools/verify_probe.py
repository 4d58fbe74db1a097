#!/usr/bin/env python3
"""-c (verify_after_compress) alone: `inst` instances, each compressing + verifying `blocks` resident chunks per call; prints the
wall time of every call and the compression part of it (dsrcgpu_last_timing), so that the verifying pass's share and the way the
instances overlap can be read off.

    python tools/verify_probe.py --blocks 300 --inst 4 --passes 2
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
from dsrc_amd._lib import Handle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=300)
    ap.add_argument("--inst", type=int, default=4)
    ap.add_argument("--passes", type=int, default=2)
    a = ap.parse_args()
    hs = [Handle(9, 2, crc=True, quality_offset=33, verify=True) for _ in range(a.inst)]
    n = a.blocks
    recs = int(n * bench.RECS_PER_BLOCK * 1.02) + 1000
    cap_in = recs * 384
    d_in = hs[0].dev_alloc(cap_in)
    nbytes = hs[0].synth_illumina(1, recs, d_in, cap_in)
    off = bench.record_offsets(1, recs)
    starts, sizes = bench.cut_blocks(off, n)
    cap_out = cap_in // 2
    outs = [h.dev_alloc(cap_out) for h in hs]
    log = []

    def work(i, passes):
        for p in range(passes):
            t0 = time.perf_counter()
            hs[i].compress_batch_device(d_in, starts, sizes, outs[i], cap_out)
            t1 = time.perf_counter()
            log.append((i, p, round(t0 - T0, 3), round(t1 - t0, 3), round(hs[i].last_timing()[0], 1)))
    T0 = time.perf_counter()
    for i in range(a.inst):
        work(i, 1)
    print("warm-up", log, flush=True); log.clear()
    ths = [threading.Thread(target=work, args=(i, a.passes)) for i in range(a.inst)]
    T0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - T0
    for e in sorted(log, key=lambda e: e[2]):
        print(json.dumps({"inst": e[0], "pass": e[1], "start_s": e[2], "call_s": e[3], "compress_gpu_ms": e[4]}))
    print(json.dumps({"inst": a.inst, "blocks": n, "MB_per_s": round(sum(sizes) * a.inst * a.passes / dt / 1e6, 1), "s": round(dt, 3)}), flush=True)


if __name__ == "__main__":
    main()
