#!/usr/bin/env python3
"""The queue form of the C ABI (dsrcgpu_submit / flush / collect / release) with host-resident chunks, one and two handles, with
the second scheduler lane of a handle on (default) and off (DSRC_GPU_QUEUE_LANES=1).  Usage: tools/queue_bench.py [batches=6] [per_batch=192]"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from dsrc_amd._lib import Handle  # noqa: E402
from tests._oracle import Config  # noqa: E402


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    per_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    cfg = Config.from_levels(3, 2)
    h = Handle(cfg.dna_order, cfg.quality_order)
    recs = 64 * 22400
    cap = recs * 400
    d = h.dev_alloc(cap); n = h.synth_illumina(1, recs, d, cap); text = h.dev_download(d, n); h.dev_free(d); h.close()
    chunks = []; pos = 0
    while pos < n and len(chunks) < 64:
        end = min(n, pos + (8 << 20))
        if end < n:
            end = text.rfind(b"\n@SRRSYN", pos, end) + 1
        chunks.append(text[pos:end - 1]); pos = end
    for nh in (1, 2):
        r = bench.measure_queue_form(cfg, 0, chunks, nh, batches=batches, per_batch=per_batch)
        print(f"{nh} handle(s), {per_batch} chunks per flush: {r}", flush=True)
    for pb in (per_batch, 96):
        r = bench.measure_queue_form(cfg, 0, chunks, 1, batches=batches * 2, per_batch=pb, pinned=True)
        print(f"1 handle, dsrcgpu_submit_pinned, {pb} chunks per flush: {r}", flush=True)


if __name__ == "__main__":
    main()
