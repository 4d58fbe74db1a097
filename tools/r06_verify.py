#!/usr/bin/env python3
"""-c on the device, round 6: one handle, one call of N chunks (scheduler lanes inside the handle, ONE verifying pass over the call's
blocks) against lanes off (one lane, one pass) -- tools/r06_verify.py [blocks per call ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import bench  # noqa: E402
from dsrc_amd._lib import Handle  # noqa: E402


def main():
    for nb in [int(x) for x in sys.argv[1:]] or [450, 1800]:
        h0 = Handle(9, 2, crc=True, verify=True)
        recs = int(nb * bench.RECS_PER_BLOCK * 1.02) + 1000
        d_in = h0.dev_alloc(recs * 384)
        nbytes = h0.synth_illumina(1, recs, d_in, recs * 384)
        off = bench.record_offsets(1, recs)
        assert off[-1] == nbytes
        starts, sizes = bench.cut_blocks(off, nb)
        cap = recs * 384 // 2
        d_out = h0.dev_alloc(cap)
        for lanes in ((1, 0), (4, 0)):
            h0.set_lanes(*lanes)
            h0.compress_batch_device(d_in, starts, sizes, d_out, cap)          # warm-up: arenas, table region
            t0 = time.perf_counter()
            for _ in range(2):
                h0.compress_batch_device(d_in, starts, sizes, d_out, cap)
            dt = (time.perf_counter() - t0) / 2
            print(f"{nb} blocks per call, lanes {lanes[0]}: {sum(sizes) / dt / 1e6:.1f} MB/s ({dt * 1e3:.0f} ms per call)", flush=True)
            h0.release_memory()
        h0.dev_free(d_in); h0.dev_free(d_out); h0.close()


if __name__ == "__main__":
    main()
