#!/usr/bin/env python3
"""-c on the device: dsrcgpu_compress_batch_device with verify_after_compress on the bench workload, one and four instances,
with the DNA and quality chains of the verifying pass at once (default) and one after the other (DSRC_GPU_VERIFY_SERIAL=1).
Usage: tools/verify_bench.py [blocks per call, default 450]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
from dsrc_amd._lib import Handle  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 450
    h0 = Handle(9, 2, crc=True)
    recs = nb * 22400
    cap = recs * 400
    d_in = h0.dev_alloc(cap)
    n = h0.synth_illumina(1, recs, d_in, cap)
    text = h0.dev_download(d_in, n)
    # chunk boundaries: 8 MiB chunks cut at record starts
    starts = []; sizes = []; pos = 0
    while pos < n and len(starts) < nb:
        end = min(n, pos + (8 << 20))
        if end < n:
            end = text.rfind(b"\n@SRRSYN", pos, end) + 1
        starts.append(pos); sizes.append(end - 1 - pos); pos = end
    h0.close()
    for n_inst in (1, 4):
        hs = [Handle(9, 2, crc=True, verify=True) for _ in range(n_inst)]
        outs = [h.dev_alloc(n // 2) for h in hs]

        def work(i, passes):
            for _ in range(passes):
                hs[i].compress_batch_device(d_in, starts, sizes, outs[i], n // 2)
        for i in range(n_inst):
            work(i, 1)
        ths = [threading.Thread(target=work, args=(i, 2)) for i in range(n_inst)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        print(f"{n_inst} instance(s) x {len(starts)} blocks: {2 * n_inst * sum(sizes) / dt / 1e6:.1f} MB/s, verify pass {hs[0].last_verify_ms() if hasattr(hs[0], 'last_verify_ms') else 0:.0f} ms")
        for h, o in zip(hs, outs):
            h.dev_free(o); h.close()


if __name__ == "__main__":
    main()
