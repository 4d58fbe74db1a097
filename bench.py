#!/usr/bin/env python3
"""bench.py -- raw FASTQ MB/s through the MI355X block-compression path (BASELINE.json metric).

Workload (BASELINE.json configs[2] shape): synthetic Illumina-like 150 bp FASTQ of the 100 M-read data
set, -d3 -q2, default -b8 (8 MiB chunks cut where IFastqStreamReader::ReadNextChunk would cut them).
A *step* = `--blocks` consecutive chunks per GPU, generated in HBM by the counter-based generator before
the timed region and pushed through dsrcgpu_compress_batch_device; compressed blocks stay in HBM (PCIe is
not in `value`).  Inside a step the chunks go through `--pipeline` independent scheduler instances
(own HIP stream + arena each) a fraction of a period apart, so that the serial range-coder kernel of one sub-batch
overlaps the data-parallel front end of the next; the timed region covers all K steps end to end.

N > 1: one process per GPU (torch.distributed, nccl = RCCL).  Ranks take disjoint record ranges (weak
scaling, no collective in the data path); after each step the compressed block stream is gathered to
rank 0 in archive order (dsrc_amd/dist.py), inside the timed region and overlapped with the next step's compression
(two output buffers per scheduler instance).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


BUF = 8 << 20                     # -b8 (reference default, src/Common.h:156)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
RECS_PER_BLOCK = 22300
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_final.json")   # written by tools/r06_pmc.sh from rocprofv3 --pmc passes of the library it names
MAX_RESIDENT = 3                  # distinct input shards kept in HBM per scheduler instance (2 when N > 1: rank 0 also holds the gathered streams)
MAX_RESIDENT_BOX = [MAX_RESIDENT]   # ... lowered by main() when the device has less free HBM than the run would take


def csrc_sha() -> str:
    """What the counter file is valid for: the kernel sources (dsrc_amd/csrc/*.h, *.hip) it was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dsrc_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_pmc():
    """L2 <-> fabric traffic per block from the PMC passes (FETCH_SIZE doubled for the 16 B/lane streaming reads, as MI355X_MICROARCH.md
    prescribes, + WRITE_SIZE; separate passes).  A file measured on other kernel sources is refused: every traffic figure is then null."""
    try:
        with open(PMC_FILE) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return None, f"{os.path.relpath(PMC_FILE, ROOT)} missing"
    if pmc.get("csrc_sha") != csrc_sha():
        return None, f"{os.path.relpath(PMC_FILE, ROOT)} is stale (measured on kernel sources {pmc.get('csrc_sha')}, commit {pmc.get('commit')}; these are {csrc_sha()}): run tools/r06_pmc.sh"
    return pmc, f"{os.path.relpath(PMC_FILE, ROOT)} (commit {pmc.get('commit')})"


def title_len(i: np.ndarray) -> np.ndarray:
    def digits(v):
        d = np.ones(v.shape, dtype=np.int64)
        for k in range(1, 19):
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
    """Chunk boundaries as IFastqStreamReader::ReadNextChunk places them (reference src/FastqStream.cpp:18-98):
    a chunk ends just before the first record that starts after byte (start + buf - 8192); its size excludes
    the final newline."""
    starts, sizes = [], []
    r = 0
    for _ in range(nblocks):
        start = off[r]
        nxt = int(np.searchsorted(off, start + BUF - 8192, side="right"))
        if nxt >= len(off):
            break
        starts.append(int(start)); sizes.append(int(off[nxt] - start - 1))
        r = nxt
    return starts, sizes


def cpu_baseline(write_sample, d: int, q: int):
    """The unmodified reference (oracle/_ref: DsrcCompressorMT, then DsrcDecompressorMT on its own archive) on the host
    cores, or our C port on one core when _ref is absent.  Reported beside the GPU numbers; it is not the target.
    write_sample(file) writes the sample FASTQ and returns its size."""
    import tempfile
    from tests._oracle import Oracle, Ref, have_ref
    # the reference's queues use a 64-bit completion mask: thread counts >= 64 are undefined
    cores = min(os.cpu_count() or 1, 60)
    dec = None; crc = None
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        src = os.path.join(td, "s.fastq"); dst = os.path.join(td, "s.dsrc"); back = os.path.join(td, "b.fastq")
        with open(src, "wb") as f:
            size = write_sample(f)
        if have_ref():
            r = Ref()
            t = time.time(); rc = r.compress_file(src, dst, d, q, False, False, 33, 8, cores); dt = time.time() - t
            kind = "reference"
            t = time.time(); rc3 = r.compress_file(src, dst + "c", d, q, False, True, 33, 8, cores); dt3 = time.time() - t
            if rc3 == 0:
                crc = {"value": round(size / dt3 / 1e6, 2), "unit": "MB/s", "cores": cores, "kind": "reference",
                       "sample": f"`dsrc c -d{d} -q{q} -c -t{cores}` on the same {size} bytes, tmpfs to tmpfs, wall {dt3:.2f} s"}
            os.unlink(src)
            t = time.time(); rc2 = r.decompress_file(dst, back, cores); dt2 = time.time() - t
            if rc2 == 0 and os.path.getsize(back) == size:
                dec = {"value": round(size / dt2 / 1e6, 2), "unit": "MB/s", "cores": cores, "kind": "reference",
                       "sample": f"`dsrc d -t{cores}` of the archive of the same {size} bytes, tmpfs to tmpfs, wall {dt2:.2f} s"}
        else:
            o = Oracle(); cores = 1
            t = time.time(); rc = o.compress_file(src, dst, d, q, False, False, 33, 8); dt = time.time() - t
            kind = "port"
        assert rc == 0
    return {"value": round(size / dt / 1e6, 2), "unit": "MB/s", "cores": cores, "kind": kind,
            "sample": f"{size} bytes of the same synthetic FASTQ ({size // BUF + 1} blocks), -d{d} -q{q} -b8, "
                      f"input and output in tmpfs, {cores} worker threads, wall {dt:.2f} s"}, dec, crc


class Lane:
    """One scheduler instance: own handle (HIP stream + arena) and its sub-batches."""

    def __init__(self, cfg, device, sub_blocks, n_sub, rank, lane_id, n_lanes, alloc_out, n_out=1, binned=False, lanes=(1, 0)):
        from dsrc_amd._lib import Handle
        self.h = Handle(cfg.dna_order, cfg.quality_order, quality_offset=33, device=device)
        self.h.set_lanes(*lanes)          # scheduler lanes inside the handle (1 = the call runs on the handle's own lane)
        self.sub = []
        recs = int(sub_blocks * RECS_PER_BLOCK * 1.02) + 1000
        cap_in = recs * 384
        self.cap_out = cap_in // 2
        # all inputs of the timed region stay resident in HBM; beyond MAX_RESIDENT distinct shards per lane they are
        # reused cyclically (a shard is ~3.4 GB, far beyond any cache)
        self.n_res = min(n_sub, MAX_RESIDENT_BOX[0] if n_out == 1 else min(2, MAX_RESIDENT_BOX[0]))
        for k in range(self.n_res):
            gid = (rank * n_lanes + lane_id) * MAX_RESIDENT + k     # disjoint record range per (rank, lane, shard)
            first = 1 + gid * recs
            d_in = self.h.dev_alloc(cap_in)
            nbytes = self.h.synth_illumina(first, recs, d_in, cap_in, binned=binned)
            off = record_offsets(first, recs)
            assert off[-1] == nbytes, (off[-1], nbytes)
            starts, sizes = cut_blocks(off, sub_blocks)
            assert len(starts) == sub_blocks
            self.sub.append((d_in, starts, sizes))
        # two output buffers: with N > 1 the block stream of step s is gathered while step s+1 is being compressed
        self.outs = [alloc_out(self.h, self.cap_out) for _ in range(n_out)]
        self.results = {}
        self.timing = []
        self.trace = []

    def shard(self, k):
        return self.sub[k % self.n_res]

    def free(self):
        """Inputs, outputs and the handle go back to the device (dsrcgpu_destroy frees the handle's own arena, not what dev_alloc gave out)."""
        if getattr(self.h, "h", None):
            for d_in, _, _ in self.sub:
                self.h.dev_free(d_in)
            for ptr, tensor in self.outs:
                if tensor is None:
                    self.h.dev_free(ptr)
            self.sub = []; self.outs = []
            self.h.close()

    def run(self, k):
        d_in, starts, sizes = self.shard(k)
        t0 = time.perf_counter()
        res = self.h.compress_batch_device(d_in, starts, sizes, self.outs[k % len(self.outs)][0], self.cap_out)
        t1 = time.perf_counter()
        self.results[k] = res
        self.timing.append(self.h.last_timing() + self.h.last_stage_timing())
        self.trace.append((k, t0, t1) + self.timing[-1][:2])
        return res


class StepGates:
    """N > 1 only.  Every scheduler instance has a gather thread of its own (its own RCCL group and its own host-side group for the
    footer table, so that the instances' exchanges do not have to agree on an order): the gather of instance i's step s runs as soon
    as i has finished step s, while i already compresses step s+1 into its other output buffer; i may start step s only after the
    gather of its step s-2 (same buffer) is done."""

    def __init__(self, n_lanes):
        self.cv = threading.Condition(); self.done = set(); self.gathered = set(); self.n_lanes = n_lanes

    def lane_may_start(self, idx, s, first):
        with self.cv:
            while s - 2 >= first and (idx, s - 2) not in self.gathered:
                self.cv.wait()

    def lane_done(self, idx, s):
        with self.cv:
            self.done.add((idx, s)); self.cv.notify_all()

    def wait_lane(self, idx, s):
        with self.cv:
            while (idx, s) not in self.done:
                self.cv.wait()

    def lane_gathered(self, idx, s):
        with self.cv:
            self.gathered.add((idx, s)); self.cv.notify_all()


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks ourselves (one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* as torch.distributed.run would set them), rank 0 prints the line.  DSRC_BENCH_SAME_GPU=1 (tests, one-GPU
    boxes): every rank on GPU 0 and the exchanges over gloo through host copies -- RCCL refuses two ranks on one device."""
    import socket
    import subprocess
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    same = bool(os.environ.get("DSRC_BENCH_SAME_GPU"))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if same else str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        while procs and rc == 0:
            for pr in list(procs):
                try:
                    code = pr.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                procs.remove(pr)
                rc = rc or code
    finally:
        for pr in procs:              # a rank failed (or we were interrupted): the others would wait in a collective for ever
            pr.kill()
    return rc


def measure_decode(lanes, cfg, n_blocks, last_step, pmc, n_inst=int(os.environ.get("DSRC_BENCH_DECODE_INST", "2")), passes=int(os.environ.get("DSRC_BENCH_DECODE_PASSES", "3"))):
    """Secondary line: the same blocks back through the GPU decompressor (dsrcgpu_decompress_batch_device), everything in
    HBM.  The blocks are the ones instance 0 wrote in its last sub-batch, taken as many times as needed to make
    `n_blocks` per pass (every copy is decoded into its own text; the decoder's work does not depend on the data being distinct).
    `n_inst` decoding instances run `passes` passes each, concurrently: a pass is a chain of stages (titles, quality, DNA,
    layout) of which the quality stage fills the SIMDs and the DNA stage (one lane per block) hardly uses them, so two passes in
    flight overlap one's DNA stage with the other's quality stage.
    The compression instances are closed first: a decoding pass wants the HBM for model tables (one per block in flight)."""
    from dsrc_amd._lib import Handle
    ln = lanes[0]
    for other in lanes[1:]:
        for d_in, _, _ in other.sub:
            other.h.dev_free(d_in)
        for ptr, tensor in other.outs:
            if tensor is None:
                other.h.dev_free(ptr)
        other.sub = []; other.outs = []
        other.h.close()
    o_offs, o_sizes, _, _ = ln.results[last_step]
    d_blk = ln.outs[last_step % len(ln.outs)][0]
    d_in, starts, sizes = ln.shard(last_step)
    reps = max(1, (n_blocks + len(o_offs) - 1) // len(o_offs))
    offs = (o_offs * reps)[:n_blocks]; szs = (o_sizes * reps)[:n_blocks]
    text_bytes = sum((sizes * reps)[:n_blocks]) + n_blocks
    hs = [ln.h] + [Handle(cfg.dna_order, cfg.quality_order, quality_offset=33, device=ln.h.device) for _ in range(n_inst - 1)]
    txts = [h.dev_alloc(text_bytes + 4096) for h in hs]
    out = [None] * n_inst; gpu_ms = [0.0] * n_inst
    try:
        for h, d_txt in zip(hs, txts):                  # warm-up: sizes the arenas and the table regions
            h.decompress_batch_device(d_blk, offs, szs, d_txt, text_bytes + 4096, verify=True)

        def work(i):
            for _ in range(passes):
                out[i] = hs[i].decompress_batch_device(d_blk, offs, szs, txts[i], text_bytes + 4096, verify=True)
                gpu_ms[i] += hs[i].last_timing()[0]
        ths = [threading.Thread(target=work, args=(i,)) for i in range(n_inst)]
        t0 = time.perf_counter()
        for i, t in enumerate(ths):
            t.start()
            if i + 1 < n_inst:
                time.sleep(float(os.environ.get("DSRC_BENCH_DECODE_STAGGER", "1.5")))     # half a pass apart: the stages interleave
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        for i in range(n_inst):
            t_offs, t_sizes, ok = out[i]
            assert sum(t_sizes) == text_bytes and all(ok)
            # parity: the text of the first and the last copy is the chunk that was compressed
            for k in (0, n_blocks - 1):
                src = ln.h.dev_download(d_in + starts[k % len(starts)], sizes[k % len(starts)])
                assert hs[i].dev_download(txts[i] + t_offs[k], t_sizes[k]) == src + b"\n", f"decode parity check failed on block {k}"
    finally:
        for h, d_txt in zip(hs, txts):
            h.dev_free(d_txt)
        for h in hs[1:]:
            h.close()
    total_text = text_bytes * n_inst * passes
    alg = (text_bytes + sum(szs)) * n_inst * passes
    pass_ms = sum(gpu_ms) / (n_inst * passes)
    return {"metric": f"raw FASTQ MB/s decompressed (text identical to the input) at -d{cfg.dna_order // 3} -q{cfg.quality_order}",
            "value": round(total_text / dt / 1e6, 1), "unit": "MB/s", "blocks": n_blocks, "instances": n_inst, "passes_per_instance": passes, "ms": round(dt * 1e3, 1),
            "data": f"{len(o_offs)} distinct blocks of the timed region x {reps} per pass, compressed blocks and decoded text resident in HBM; "
                    f"{n_inst} decoding instances x {passes} passes of {n_blocks} blocks, concurrent",
            "roofline": {"bound": "hbm", "kernel": "k_dec_qrc (one wavefront per block: range decoding of the quality stream; the DNA stream follows in k_dec_dnarc, one lane per block)",
                         "achieved": round(alg / dt / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / dt / 1e9 / HBM_PEAK_GBS, 6), "pass_ms": round(pass_ms, 1), "launch_bytes": int(text_bytes + sum(szs)),
                         "traffic": int(pmc["decode"]["bytes_per_block"] * n_blocks) if pmc and pmc.get("decode") and cfg.dna_order == 9 and cfg.quality_order == 2 else None,
                         "note": "algorithmic bytes = block bytes in + text bytes out of a pass; a decoded stream is a chain of dependent model-row accesses, one 64-byte row fetched and written back per symbol (traffic: rocprofv3 FETCH_SIZE + WRITE_SIZE of every decoding kernel per block, profiles/r06_pmc_final.json), and the kernels are bound by latency x blocks in flight and instruction issue, not by bandwidth (DESIGN section 7)"}}


def measure_queue_form(cfg, device, chunks, n_handles, batches=4, per_batch=192, pinned=False):
    """The form INTEGRATION.md section 1 binds in place of DsrcCompressor::Process: host-resident chunks go in through
    dsrcgpu_submit / flush, blocks come back through dsrcgpu_collect / release (reference: src/DsrcWorker.cpp:30-73).  PCIe and the
    copy into the page-locked ring are inside the figure.  One submitting and one collecting thread per handle, as the header allows."""
    import ctypes as C
    from dsrc_amd._lib import Handle
    hs = [Handle(cfg.dna_order, cfg.quality_order, quality_offset=33, device=device) for _ in range(n_handles)]
    L = hs[0].L
    errs = []
    pin = None; pin_at = []
    if pinned:          # the chunks in page-locked memory of the caller's (what a host pipeline's reader threads fill): dsrcgpu_submit_pinned
        from dsrc_amd._lib import host_alloc
        pin = host_alloc(sum(len(c) + 256 for c in chunks))
        at = 0
        for c in chunks:
            C.memmove(pin + at, c, len(c)); pin_at.append(at); at += (len(c) + 255) & ~255

    def submitter(h, nb):
        try:
            k = 0
            for _ in range(nb):
                for _ in range(per_batch):
                    c = chunks[k % len(chunks)]; k += 1
                    while True:
                        if pinned:
                            rc = L.dsrcgpu_submit_pinned(h.h, C.c_int64(k), C.c_void_p(pin + pin_at[(k - 1) % len(chunks)]), C.c_uint64(len(c)))
                        else:
                            rc = L.dsrcgpu_submit(h.h, C.c_int64(k), c, C.c_uint64(len(c)))
                        if rc != -8:
                            break
                        time.sleep(0.0005)          # ring full: the collector is behind
                    if rc:
                        raise RuntimeError(L.dsrcgpu_last_error(h.h).decode())
                if L.dsrcgpu_flush(h.h):
                    raise RuntimeError(L.dsrcgpu_last_error(h.h).decode())
        except Exception as e:          # noqa: BLE001
            errs.append(e)

    def collector(h, want):
        pid = C.c_int64(); blk = C.POINTER(C.c_uint8)(); sz = C.c_uint64()
        got = 0
        try:
            while got < want and not errs:
                rc = L.dsrcgpu_collect(h.h, C.byref(pid), C.byref(blk), C.byref(sz), None, None)
                if rc < 0:
                    raise RuntimeError(L.dsrcgpu_last_error(h.h).decode())
                if rc == 0:
                    time.sleep(0.0005); continue
                L.dsrcgpu_release(h.h, blk); got += 1
        except Exception as e:          # noqa: BLE001
            errs.append(e)

    def run(nb):
        ths = []
        for h in hs:
            ths += [threading.Thread(target=submitter, args=(h, nb)), threading.Thread(target=collector, args=(h, nb * per_batch))]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return time.perf_counter() - t0
    try:
        run(6)                                   # sizes the page-locked ring (the lanes + 2 batches) and the lanes' arenas
        dt = run(batches)
    finally:
        for h in hs:
            h.close()
        if pin:
            from dsrc_amd._lib import host_free
            host_free(pin)
    if errs:
        raise errs[0]
    nbytes = sum(len(chunks[k % len(chunks)]) for k in range(batches * per_batch)) * n_handles
    return round(nbytes / dt / 1e6, 1), nbytes, dt


def measure_verify(cfg, ln, n_inst, blocks=None):
    """-c: the compressing call decodes on the device what it has just written and compares the checksums
    (dsrcgpu_settings::verify_after_compress; reference: DsrcCompressor::Process, src/DsrcWorker.cpp:53-62).  `n_inst` scheduler
    instances run concurrently, each on one of instance 0's resident shards -- or, with `blocks`, on a shard of that many chunks of
    its own: a verifying pass is a chain of dependent reads per block, it takes about as long for 900 blocks as
    for 450, so its rate is the number of blocks in flight over the chain's length."""
    from dsrc_amd._lib import Handle
    hs = [Handle(cfg.dna_order, cfg.quality_order, crc=True, quality_offset=33, device=ln.h.device, verify=True) for _ in range(n_inst)]
    if n_inst > 1:
        for h in hs:
            h.set_lanes(1)            # several handles side by side are the lanes (as in rounds 3-5); one handle alone cuts its calls itself
    own = []
    if blocks:
        recs = int(blocks * RECS_PER_BLOCK * 1.02) + 1000
        for i, h in enumerate(hs):
            first = 200_000_001 + i * recs                   # records no other shard of the run holds; numbers stay below 10^9 (larger
                                                             # numeric title fields are undefined in the reference's decoder and refused by ours)
            d_in = h.dev_alloc(recs * 384)
            nbytes = h.synth_illumina(first, recs, d_in, recs * 384)
            off = record_offsets(first, recs)
            assert off[-1] == nbytes
            starts, sizes = cut_blocks(off, blocks)
            own.append((d_in, starts, sizes))
    cap_out = (int(blocks * RECS_PER_BLOCK * 1.02) + 1000) * 384 // 2 if blocks else ln.cap_out
    outs = [h.dev_alloc(cap_out) for h in hs]
    done = [0] * n_inst

    def work(i, passes):
        d_in, starts, sizes = own[i] if blocks else ln.shard(i)
        for _ in range(passes):
            hs[i].compress_batch_device(d_in, starts, sizes, outs[i], cap_out)
            done[i] += sum(sizes)
    try:
        for i in range(n_inst):
            work(i, 1)                               # warm-up: arena, table region
        done = [0] * n_inst
        ths = [threading.Thread(target=work, args=(i, 2)) for i in range(n_inst)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
    finally:
        for h, o in zip(hs, outs):
            h.dev_free(o)
        for h, (d_in, _, _) in zip(hs, own):
            h.dev_free(d_in)
        for h in hs:
            h.close()
    return round(sum(done) / dt / 1e6, 1), blocks or len(ln.shard(0)[1])


def measure_one_handle(cfg, device, blocks, lanes=4, steps=3):
    """What ONE handle does by itself (task 2): the step of the headline -- `blocks` consecutive chunks, device-resident
    -- as one dsrcgpu_compress_batch_device call on one handle, the scheduler lanes inside it (dsrcgpu_set_lanes: the library's default
    sub-batches of about 1.9 GB).  HBM: what the device has less free while the call's lanes hold their arenas (inputs and output
    buffer not counted).  The last block of the last call is compared with the oracle."""
    import ctypes as C
    from dsrc_amd._lib import load

    def alloc_out(h, cap):
        return h.dev_alloc(cap), None
    L = load()

    def free_hbm():
        fr = C.c_uint64(); tot = C.c_uint64()
        return fr.value if L.dsrcgpu_device_memory(device, C.byref(fr), C.byref(tot)) == 0 else 0
    ln = Lane(cfg, device, blocks, steps + 1, 0, 0, 1, alloc_out, lanes=(lanes, 0))
    try:
        before = free_hbm()
        ln.run(0)                                          # arenas of the lanes
        held = before - free_hbm()
        t0 = time.perf_counter()
        for s_ in range(1, steps + 1):
            ln.run(s_)
        wall = time.perf_counter() - t0
        in_bytes = sum(sum(ln.shard(s_)[2]) + len(ln.shard(s_)[2]) for s_ in range(1, steps + 1))
        from tests._oracle import Oracle
        d_in, starts, sizes = ln.shard(steps); o_offs, o_sizes, _, _ = ln.results[steps]
        i = len(starts) - 1
        assert ln.h.dev_download(ln.outs[0][0] + o_offs[i], o_sizes[i]) == Oracle().compress_block(cfg, ln.h.dev_download(d_in + starts[i], sizes[i]))[0], "bench parity check failed (one handle, lanes inside)"
        return {"value": round(in_bytes / wall / 1e6, 1), "unit": "MB/s", "blocks_per_call": blocks, "lanes": lanes, "calls": steps, "ms_per_call": round(wall / steps * 1e3, 1),
                "hbm_held_GB": round(held / 1e9, 1), "k_rc_ms": round(sum(t[1] for t in ln.timing[1:]) / max(1, len(ln.timing) - 1), 1), "parity_checked_blocks": 1,
                "what": "ONE handle, one dsrcgpu_compress_batch_device call per step of the headline's size, inputs and outputs in HBM: the call is cut into sub-batches (~1.9 GB of chunks) that run on scheduler lanes inside the handle, the block-to-block state handed from one to the next (include/dsrc_gpu.h dsrcgpu_set_lanes)"}
    finally:
        ln.free()


def measure_binned(cfg, device, sub_blocks, P, steps=3):
    """Second line (task 5): the same workload with the qualities quantised to four levels, as current instruments
    write them (flavour 1 of dsrcgpu_synth_fastq): a third of a quality stream then lies in one context.  Same scheduler instances,
    same step; one block of the timed region is compared with the oracle."""
    def alloc_out(h, cap):
        return h.dev_alloc(cap), None
    lanes = [Lane(cfg, device, sub_blocks, steps + 1, 0, i, P, alloc_out, binned=True) for i in range(P)]
    try:
        for ln in lanes:
            ln.run(0)
        t_sub = lanes[0].timing[-1][0] / 1e3
        errors = []

        def worker(idx):
            try:
                if idx:
                    time.sleep(t_sub * idx / P)
                for s_ in range(1, steps + 1):
                    lanes[idx].run(s_)
            except Exception as e:      # noqa: BLE001
                errors.append(e)
        ths = [threading.Thread(target=worker, args=(i,)) for i in range(P)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        wall = time.perf_counter() - t0
        if errors:
            raise errors[0]
        in_bytes = sum(sum(ln.shard(s_)[2]) + len(ln.shard(s_)[2]) for ln in lanes for s_ in range(1, steps + 1))
        out_bytes = sum(sum(ln.results[s_][1]) for ln in lanes for s_ in range(1, steps + 1))
        from tests._oracle import Oracle
        ln = lanes[-1]; d_in, starts, sizes = ln.shard(steps); o_offs, o_sizes, _, _ = ln.results[steps]
        i = len(starts) - 1
        chunk = ln.h.dev_download(d_in + starts[i], sizes[i])
        got = ln.h.dev_download(ln.outs[0][0] + o_offs[i], o_sizes[i])
        assert got == Oracle().compress_block(cfg, chunk)[0], "bench parity check failed on the four-level data"
        return {"metric": f"raw FASTQ MB/s compressed (bit-identical .dsrc) at -d{cfg.dna_order // 3} -q{cfg.quality_order}, four-level qualities",
                "value": round(in_bytes / wall / 1e6, 1), "unit": "MB/s", "steps": steps, "blocks_per_step": sub_blocks * P, "pipeline": P,
                "ms_per_step": round(wall / steps * 1e3, 2), "ratio_out_in": round(out_bytes / in_bytes, 4), "parity_checked_blocks": 1,
                "data": "the synthetic Illumina records with Phred quantised to 2 / 12 / 23 / 37 (dsrc_amd/synth.py illumina_fastq(binned=True)), in HBM"}
    finally:
        for ln in lanes:
            ln.free()


def write_config3_file(path, device, reads=100_000_000):
    """BASELINE configs[2]'s own file: records 1 .. 100 M of the counter-based generator (37.7 GB), generated in HBM 4 M reads at a time."""
    from dsrc_amd._lib import Handle
    h = Handle(device=device)
    total = 0; first = 1; piece = 4_000_000
    try:
        with open(path, "wb") as f:
            while first <= reads:
                n_rec = min(piece, reads - first + 1)
                cap = n_rec * 400
                d = h.dev_alloc(cap)
                n = h.synth_illumina(first, n_rec, d, cap)
                f.write(h.dev_download(d, n)); h.dev_free(d)
                total += n; first += n_rec
    finally:
        h.close()
    return total


def measure_host_e2e(src, size, td, inst=4, runs=3, gap=15.0, first_gap=15.0):
    """`dsrc-amd c` and `dsrc-amd d` (C++ host over the C ABI), file in tmpfs -> archive in tmpfs -> file in tmpfs, separated runs.
    The gaps are for the driver, not for the tool: HBM that a process has released is wiped at ~35 GB/s and an allocation that lands
    on memory still waiting for that is held until it is clean (profiles/r05_alloc_probe2.txt) -- this process has just released
    ~200 GB, a `dsrc-amd c` run ~70 GB, a `dsrc-amd d` run ~150 GB (a run that follows another within 6 s: 8.5 instead of 7.4 s,
    profiles/r05_e2e_7.txt)."""
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dsrc_amd", "csrc", "dsrc-amd")
    arc = os.path.join(td, "e2e.dsrc"); back = os.path.join(td, "e2e_back.fastq")
    res = {"c": [], "d": []}
    for k in range(runs):
        if os.path.exists(arc):
            os.unlink(arc)
        time.sleep(first_gap if k == 0 else gap)
        t = time.time(); subprocess.check_call([cli, "c", "-d3", "-q2", f"-t{inst}", src, arc]); res["c"].append(size / (time.time() - t) / 1e6)
    for _ in range(runs):
        if os.path.exists(back):
            os.unlink(back)
        time.sleep(gap)
        t = time.time(); subprocess.check_call([cli, "d", f"-t{inst}", arc, back]); res["d"].append(size / (time.time() - t) / 1e6)
    ok = os.path.getsize(back) == size
    with open(src, "rb") as fa, open(back, "rb") as fb:          # spot comparison: 64 MiB pieces across the file
        for o in range(0, size, max(1, size // 8)):
            fa.seek(o); fb.seek(o); ok = ok and fa.read(64 << 20) == fb.read(64 << 20)
    os.unlink(back); os.unlink(arc)
    med = lambda v: sorted(v)[len(v) // 2]
    return {"unit": "MB/s", "bytes": size, "instances": inst, "runs": runs,
            "compress": {"min": round(min(res["c"]), 1), "median": round(med(res["c"]), 1), "all": [round(x, 1) for x in res["c"]]},
            "decompress": {"min": round(min(res["d"]), 1), "median": round(med(res["d"]), 1), "all": [round(x, 1) for x in res["d"]]},
            "round_trip_identical": bool(ok),
            "spread_compress": round((max(res["c"]) - min(res["c"])) / med(res["c"]), 3), "spread_decompress": round((max(res["d"]) - min(res["d"])) / med(res["d"]), 3),
            "note": "dsrc-amd c -d3 -q2 / dsrc-amd d on BASELINE configs[2]'s own file (100 M reads), file in tmpfs to file in tmpfs, process start to exit, PCIe and file I/O included; runs separated by %.0f s (%.0f s before the first)" % (gap, first_gap)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)  # 10 x 1800 blocks of 8 MiB = 4 passes over the 100 M-read set (~4500 blocks); inputs stay resident in HBM
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=int(os.environ.get("DSRC_BENCH_BLOCKS", "1800")), help="8 MiB chunks per step per GPU")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("DSRC_BENCH_PIPELINE", "4")), help="scheduler instances per GPU (round 6, batches of 450 blocks: three 63.7, four 68.1 GB/s holding 204 GB, five 70.1 - 71.3 holding 255 GB -- which leaves rank 0 of an 8-GPU run no room for the streams it gathers --, six 67.0; five batches of 360: 64.0)")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("DSRC_BENCH_LANES", "1")), help="scheduler lanes INSIDE every handle (dsrcgpu_set_lanes; 0 = the library's default, 1 = none)")
    ap.add_argument("--sub-blocks", type=int, default=int(os.environ.get("DSRC_BENCH_SUB_BLOCKS", "0")), help="chunks per sub-batch of a handle's lanes (0 = the library's default, about 1 GiB)")
    ap.add_argument("--buf-mb", type=int, default=8, help="chunk size (the reference's -b; 8 = BASELINE's configurations; -m1 / -m2 of the reference's command line are 64 / 256)")
    ap.add_argument("--dna", type=int, default=3)
    ap.add_argument("--qua", type=int, default=2)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-blocks", type=int, default=int(os.environ.get("DSRC_BENCH_CPU_BLOCKS", "480")), help="8 MiB chunks of the CPU baseline sample (480 = 4 GB)")
    ap.add_argument("--decode-blocks", type=int, default=int(os.environ.get("DSRC_BENCH_DECODE_BLOCKS", "3600")),
                    help="blocks of the secondary decompression measurement (0 = skip)")
    ap.add_argument("--check", type=int, default=2, help="blocks of the first sub-batch to verify against the oracle")
    ap.add_argument("--dump-step", default=None, help="N > 1 code path only (tests): rank 0 writes the gathered block stream of the last step as an archive (gathered.dsrc) and the FASTQ text of that step (step.fastq) into this directory")
    args = ap.parse_args()
    global BUF, RECS_PER_BLOCK
    if args.buf_mb != 8:
        BUF = args.buf_mb << 20; RECS_PER_BLOCK = RECS_PER_BLOCK * args.buf_mb // 8

    # Every scheduler instance drives two HIP streams (front end + range coder).  The HIP runtime multiplexes streams onto
    # 4 hardware queues by default, which serialises unrelated instances behind each other's 0.25 s range-coder kernel
    # (and behind the host framework's own streams when N > 1); must be set before the runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(2 * max(1, args.pipeline) * (1 if args.lanes == 1 else (args.lanes or 4)) + 6))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} was started inside a group of WORLD_SIZE={world} ranks: the line would not be what was asked for")
    dist = None; torch = None
    host_payload = bool(os.environ.get("DSRC_BENCH_SAME_GPU"))       # (tests) ranks share GPU 0: gloo, payloads through host copies
    if world > 1 or os.environ.get("DSRC_BENCH_FORCE_DIST"):      # FORCE_DIST: exercise the N > 1 code path with one rank
        import torch as torch_
        import torch.distributed as dist_
        torch = torch_
        torch.cuda.set_device(local)
        if host_payload:
            dist_.init_process_group("gloo")
        else:
            dist_.init_process_group("nccl", device_id=torch.device("cuda", local))
        dist = dist_
        assert dist.get_world_size() == args.gpus, f"{dist.get_world_size()} ranks in the group, --gpus {args.gpus}"
    group_name = "gloo group, one GPU shared: a test" if host_payload else "RCCL group"
    xdev = "cpu" if host_payload else "cuda"          # where the tensors of the small exchanges live

    from dsrc_amd.config import Config
    cfg = Config.from_levels(args.dna, args.qua)
    P = max(1, args.pipeline)
    sub_blocks = max(1, args.blocks // P)
    total_steps = args.steps + args.warmup

    def alloc_out(h, cap):
        if torch is not None:                      # a torch tensor so that the gather can send it over RCCL
            t = torch.empty(cap, dtype=torch.uint8, device="cuda")
            return t.data_ptr(), t
        return h.dev_alloc(cap), None

    # What the scheduler instances of this process will take of the device, against what is free: batch arena (~9 x the chunks of a
    # sub-batch + the element slice), the resident input shards, the output buffers -- and on rank 0 of a multi-GPU run one receive
    # buffer per peer.  Too much: fewer resident shards first, then smaller sub-batches (a step stays `--blocks` chunks: more of them).
    def hbm_need(sb, n_res):
        chunks = sb * RECS_PER_BLOCK * 1.02 * 384
        if args.lanes == 1:
            arena = chunks * 7.5 + (7.6e9 if chunks >= 2.5e9 else 1.9e9) + sb * 1.05e6      # dsrc_gpu.hip estimate_arena: 15/2 x the chunks + the element slice + 1 MB per chunk
        else:                       # lanes inside the handle: an arena per lane, sized for a sub-batch (about 1 GiB of chunks unless told otherwise)
            sub = min(chunks, (args.sub_blocks * BUF * 1.0) if args.sub_blocks else 1.9e9)
            arena = (args.lanes or 4) * (sub * 7.5 + 1.9e9 + sub / BUF * 1.05e6)
        per_lane = arena + n_res * chunks + (2 if dist is not None else 1) * chunks / 2
        return P * per_lane + (P * (world - 1) * chunks / 2 if dist is not None and rank == 0 and not host_payload else 0)
    try:
        from dsrc_amd._lib import load as _load
        import ctypes as _C
        fr = _C.c_uint64(); tot_ = _C.c_uint64()
        if _load().dsrcgpu_device_memory(local, _C.byref(fr), _C.byref(tot_)) == 0:
            while MAX_RESIDENT_BOX[0] > 1 and hbm_need(sub_blocks, min(MAX_RESIDENT_BOX[0], 2 if dist is not None else 3)) > 0.92 * fr.value:
                MAX_RESIDENT_BOX[0] -= 1
            while sub_blocks > 64 and hbm_need(sub_blocks, MAX_RESIDENT_BOX[0]) > 0.92 * fr.value:
                sub_blocks = sub_blocks * 3 // 4
            args.blocks = sub_blocks * P
    except OSError:
        pass
    lanes = [Lane(cfg, local, sub_blocks, total_steps, rank, i, P, alloc_out, n_out=2 if dist is not None else 1, lanes=(args.lanes, args.sub_blocks)) for i in range(P)]

    if dist is not None:
        # block-to-block state across ranks (dsrc_amd/dist.py): rank r starts as if ranks 0..r-1 had compressed their chunks
        from dsrc_amd import synth as synth_
        from dsrc_amd._lib import load as load_lib
        from dsrc_amd.dist import exchange_fields_capacity
        t0 = synth_.illumina_title(1 + rank * P * MAX_RESIDENT * (int(sub_blocks * RECS_PER_BLOCK * 1.02) + 1000))
        seed = exchange_fields_capacity([load_lib().dsrcgpu_title_fields(t0, len(t0), 0)], device=torch.device(xdev, local) if xdev == "cuda" else torch.device("cpu"))
        for ln in lanes:
            ln.h.set_fields_capacity(seed)

    def sync_all():
        if torch is not None:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # Every scheduler instance gathers through groups of its own: a device group (RCCL) for the streams, a host group (gloo) for the
    # footer tables -- the instances' gather threads then need no common order.  Rank 0 receives every peer's stream of an instance
    # into buffers allocated once (not inside the timed region): one per peer AND instance, the instances' gathers overlap.
    lane_groups = None; recv_bufs = None
    if dist is not None:
        lane_groups = [(None if host_payload else dist.new_group(), dist.new_group(backend="gloo")) for _ in range(P)]
        if rank == 0 and world > 1:
            recv_bufs = [[None] + [torch.empty(lanes[0].cap_out, dtype=torch.uint8, device="cpu" if host_payload else "cuda") for _ in range(1, world)] for _ in range(P)]

    def gather_lane(li, step, keep=None):
        from dsrc_amd.dist import gather_block_stream
        ln = lanes[li]
        _, o_sizes, _, _ = ln.results[step]
        pay = ln.outs[step % len(ln.outs)][1]
        if host_payload:
            pay = pay[: sum(o_sizes)].cpu()
        res = gather_block_stream(o_sizes, pay, group=lane_groups[li][0], recv_bufs=recv_bufs[li] if recv_bufs else None, size_group=lane_groups[li][1])
        if keep is not None:
            keep(li, ln, res)

    def gather_step(step, keep=None):
        if dist is None:
            return
        for li in range(P):
            gather_lane(li, step, keep)

    # ---- warmup (also sizes the arenas and measures one sub-batch for the stagger) ------------------------
    t_sub = 0.0
    first_chunks = None; first_blob = None
    for s in range(args.warmup):
        for ln in lanes:
            res = ln.run(s); t_sub = ln.timing[-1][0] / 1e3        # GPU time of one sub-batch (HIP events), excludes arena allocation
            if first_chunks is None and args.check and rank == 0:
                d_in, starts, sizes = ln.shard(s)
                n = min(args.check, sub_blocks)
                first_chunks = [ln.h.dev_download(d_in + starts[i], sizes[i]) for i in range(n)]
                first_blob = (res, ln.h.dev_download(ln.outs[s % len(ln.outs)][0], res[0][n - 1] + res[1][n - 1]))
        gather_step(s)
    for ln in lanes:
        ln.timing.clear()
    hbm_held = None
    try:        # what this process holds of the device once every instance has its arena (rank 0 of N > 1: the receive buffers too)
        fr = _C.c_uint64(); tot_ = _C.c_uint64()
        if _load().dsrcgpu_device_memory(local, _C.byref(fr), _C.byref(tot_)) == 0:
            hbm_held = round((tot_.value - fr.value) / 1e9, 1)
    except (OSError, NameError):
        pass

    # ---- timed region: K steps, lanes a fraction of a period apart ------------------------------------------
    first = args.warmup
    gates = StepGates(P)
    errors = []

    def worker(idx):
        try:
            if idx:
                time.sleep(t_sub * idx / P * float(os.environ.get("DSRC_BENCH_STAGGER", "1")))
            for s in range(first, total_steps):
                if dist is not None:
                    gates.lane_may_start(idx, s, first)
                lanes[idx].run(s)
                gates.lane_done(idx, s)
        except Exception as e:      # noqa: BLE001
            errors.append(e)
            for s in range(first, total_steps):      # do not leave the gather thread waiting
                gates.lane_done(idx, s)

    def gatherer(idx):
        try:
            torch.cuda.set_device(local)
            for s in range(first, total_steps):
                gates.wait_lane(idx, s)
                if not errors:
                    gather_lane(idx, s)
                gates.lane_gathered(idx, s)
        except Exception as e:      # noqa: BLE001
            errors.append(e)
            for s in range(first, total_steps):
                gates.lane_gathered(idx, s)

    sync_all()
    t_begin = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(P)]
    if dist is not None:
        threads += [threading.Thread(target=gatherer, args=(i,)) for i in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    sync_all()
    wall = time.perf_counter() - t_begin
    if errors:
        raise errors[0]
    if os.environ.get("DSRC_BENCH_TRACE") and rank == 0:
        for i, ln in enumerate(lanes):
            for k, a, b_, bm, rm in ln.trace:
                if k >= args.warmup:
                    print(f"[trace] lane {i} sub {k}: start {1e3 * (a - t_begin):8.1f} ms  end {1e3 * (b_ - t_begin):8.1f} ms  batch {bm:7.1f} ms  k_rc {rm:6.1f} ms", file=sys.stderr)

    in_bytes = 0; out_bytes = 0
    for ln in lanes:
        for s in range(args.warmup, total_steps):
            in_bytes += sum(ln.shard(s)[2]) + len(ln.shard(s)[2])
            out_bytes += sum(ln.results[s][1])

    # parity spot-check against the oracle (outside the timed region)
    checked = 0
    if rank == 0 and first_chunks:
        from tests._oracle import Oracle
        o = Oracle()
        res, blob = first_blob
        for i, ch in enumerate(first_chunks):
            want = o.compress_block(cfg, ch)[0]
            assert blob[res[0][i]: res[0][i] + res[1][i]] == want, f"bench parity check failed on block {i}"
            checked += 1
    # ... and of what the timed region itself produced while all scheduler instances were running concurrently:
    # the last block of every instance's final sub-batch (still in its output buffer)
    if rank == 0 and args.check:
        from tests._oracle import Oracle
        o = Oracle()
        last = total_steps - 1
        for li, ln in enumerate(lanes):
            d_in, starts, sizes = ln.shard(last)
            o_offs, o_sizes, _, _ = ln.results[last]
            i = len(starts) - 1
            chunk = ln.h.dev_download(d_in + starts[i], sizes[i])
            got = ln.h.dev_download(ln.outs[last % len(ln.outs)][0] + o_offs[i], o_sizes[i])
            assert got == o.compress_block(cfg, chunk)[0], f"bench parity check failed: instance {li}, last block of the timed region"
            checked += 1

    per_rank = None; gather_verified = None
    if dist is not None:
        # ---- one more gather of the last step, checked (outside the timed region): the footer table rank 0 assembled is every rank's
        # own list of block sizes, and the bytes rank 0 holds for rank r have the checksum rank r computed of its own buffer
        last = total_steps - 1

        def digest(t):
            """Order-sensitive checksum of a uint8 device tensor, computed on the device (a stream is ~1 GB per instance and rank)."""
            a = 0; b = 0; piece = 32 << 20
            for o in range(0, t.numel(), piece):
                x = t[o: o + piece].to(torch.int64)
                w = (torch.arange(o, o + x.numel(), device=x.device, dtype=torch.int64) % 65521) + 1
                a = (a + int(x.sum())) % (1 << 61); b = (b + int((x * w).sum())) % (1 << 61)
            return (t.numel(), a, b)
        mine = []
        for ln in lanes:
            _, o_sizes, _, _ = ln.results[last]
            mine.append((list(o_sizes), digest(ln.outs[last % len(ln.outs)][1][: sum(o_sizes)])))
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        verdicts = []; dumped = []

        def keep(li, ln, res):
            if rank != 0:
                return
            sizes, bufs = res
            want_sizes = [x for r in range(world) for x in everyone[r][li][0]]
            ok = sizes == want_sizes
            for r in range(world):
                ok = ok and digest(bufs[r]) == everyone[r][li][1]
            verdicts.append(bool(ok))
            if args.dump_step:
                dumped.append((sizes, [b.cpu().numpy().tobytes() for b in bufs]))
        gather_step(last, keep)
        if rank == 0:
            gather_verified = bool(verdicts) and all(verdicts)
            if args.dump_step:
                from dsrc_amd.dist import archive_bytes
                os.makedirs(args.dump_step, exist_ok=True)
                sizes = [x for sz, _ in dumped for x in sz]
                with open(os.path.join(args.dump_step, "gathered.dsrc"), "wb") as f:
                    f.write(archive_bytes(sizes, [p for _, bl in dumped for p in bl], dna_order=cfg.dna_order, quality_order=cfg.quality_order, lossy=False,
                                          crc=False, tag_flags=0, quality_offset=33, plus_repetition=False, color_space=False))
                with open(os.path.join(args.dump_step, "step.fastq"), "wb") as f:          # this rank's records of that step, instance by instance
                    for ln in lanes:
                        d_in, starts, szs = ln.shard(last)
                        f.write(ln.h.dev_download(d_in + starts[0], starts[-1] + szs[-1] + 1 - starts[0]))
        t = torch.tensor([wall, float(in_bytes), float(out_bytes)], device=xdev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [round(float(x[1]) / float(x[0]) / 1e6, 1) for x in allt]
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        wall = float(tmax[0]); in_bytes = float(tsum[1]); out_bytes = float(tsum[2])

    out_line = None
    if rank == 0:
        value = in_bytes / wall / 1e6
        tm = [x for ln in lanes for x in ln.timing]
        batch_ms = sum(x[0] for x in tm) / max(1, len(tm))
        rc_ms = sum(x[1] for x in tm) / max(1, len(tm))
        n_sub_total = world * P * args.steps
        alg = (in_bytes + out_bytes) / n_sub_total               # chunk read once + block written once, per sub-batch launch
        achieved = alg / (rc_ms / 1e3) / 1e9 if rc_ms > 0 else 0.0
        pmc, pmc_note = load_pmc()
        pc = pmc["compress"] if pmc and args.dna == 3 and args.qua == 2 else None
        step_alg = (in_bytes + out_bytes) / (world * args.steps)               # algorithmic bytes of one step of one GPU
        step_frac = step_alg / (wall / args.steps) / 1e9 / HBM_PEAK_GBS
        line = {
            "metric": f"raw FASTQ MB/s compressed (bit-identical .dsrc) at -d{args.dna} -q{args.qua}", "value": round(value, 1), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u16/u32 integer",
            "data": f"synthetic (counter-based generator, in HBM; {lanes[0].n_res} distinct ~{sub_blocks * 8.4 / 1e3:.1f} GB shards per scheduler instance = {P * lanes[0].n_res * sub_blocks * 8.4 / 1e3:.1f} GB of distinct records per GPU, cycled)",
            "config": {"workload": f"Synthetic Illumina 150 bp FASTQ, 100M-read data set shape (BASELINE configs[2]), -d{args.dna} -q{args.qua} -b{args.buf_mb}; "
                                   f"step = {args.blocks} consecutive {args.buf_mb} MiB chunks per GPU, device-resident, {P} scheduler instances per GPU",
                       "blocks_per_step": args.blocks, "pipeline": P, "hbm_held_GB": hbm_held,
                       "parallelism": (f"{world} process(es), one per GPU ({dist.get_world_size()} ranks in the {group_name}; a gather thread and a group per scheduler instance): contiguous partId ranges, no data-path collective; per step the block sizes are all-gathered and "
                                       f"every rank's block stream goes to rank 0 by point-to-point send (RCCL), overlapped with the next step") if dist is not None else "1 GPU",
                       **({"per_rank_MB_per_s": per_rank, "gather_verified": gather_verified} if dist is not None else {}),
                       "ratio_out_in": round(out_bytes / in_bytes, 4), "parity_checked_blocks": checked},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         # L2<->fabric bytes of one k_rc launch: FETCH_SIZE x 2 (gfx950 counts 16 B/lane streaming reads at half) + WRITE_SIZE,
                         # separate rocprofv3 --pmc passes of a 512-block batch (profiles/r06_pmc_final.json; null when that file is not of this library)
                         "traffic": int(pc["k_rc_bytes_per_block"] * sub_blocks) if pc else None,
                         "kernel": "k_rcs (range coder: one lane per stream, the range recurrence and the low recurrence on two waves of a workgroup; loader waves turn records into the waves' rows and the per-symbol codes into the stream bytes)",
                         "kernel_ms": round(rc_ms, 2), "launch_bytes": int(alg), "batch_ms": round(batch_ms, 2),
                         # what bounds the RUN, not the longest launch: algorithmic bytes of a step over the step's time, and what the kernels really move
                         "step_frac": round(step_frac, 5), "step_achieved": round(step_frac * HBM_PEAK_GBS, 2),
                         "traffic_ratio": round(pc["all_bytes_per_block"] * sub_blocks / alg, 2) if pc else None,
                         "all_kernels_traffic": int(pc["all_bytes_per_block"] * sub_blocks) if pc else None,
                         "traffic_source": pmc_note,
                         "note": "algorithmic bytes = chunk bytes in + block bytes out (SURVEY 8d), of one sub-batch launch for achieved / frac (k_rc, HIP events on the range-coder stream, measured while the other scheduler instances share the GPU) and of one step for step_frac (wall time of the step: all instances, all kernels); traffic_ratio = counter traffic of every compression kernel of a batch / its algorithmic bytes"},
        }
        sort_ms = sum(x[3] for x in tm) / max(1, len(tm)); replay_ms = sum(x[4] for x in tm) / max(1, len(tm))
        if sort_ms > 0:
            # the kernels that bound the THROUGHPUT (k_rc above is the longest launch, but it is hidden behind the other
            # instances' front ends): the front end of the order models.  Same algorithmic bytes per launch group, its own summed HIP-event time
            line["roofline_frontend"] = {
                "bound": "hbm", "kernel": "k_part (the (context key, symbol, t) elements of every 8192-symbol tile of a stream, grouped by <= 1024 buckets; all launches of one sub-batch)",
                "achieved": round(alg / (sort_ms / 1e3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(alg / (sort_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5), "kernel_ms": round(sort_ms, 2), "model_ms": round(replay_ms, 2),
                "launch_bytes": int(alg), "traffic": int(pc["k_part_bytes_per_block"] * sub_blocks) if pc else None,
                "model_traffic": int(pc["model_bytes_per_block"] * sub_blocks) if pc else None,
                "note": "traffic = FETCH_SIZE x 2 + WRITE_SIZE of the k_part launches of a 512-block batch / 512; model_ms / model_traffic = k_binoff + k_model (adaptive counter rows in LDS, one wave per bucket) + k_place (time bins into stream order) of the same sub-batch"}
        decode_line = None
        if args.decode_blocks > 0 and world == 1:
            decode_line = measure_decode(lanes, cfg, args.decode_blocks, total_steps - 1, pmc)
            if decode_line:
                line["decompress"] = decode_line
        if not args.no_cpu and world == 1:
            ln = lanes[0]
            need = max(1, args.cpu_blocks)

            def write_sample(f):
                left = need; k = 0; total = 0
                while left > 0:
                    d_in, starts, sizes = ln.sub[k % ln.n_res]
                    m = min(left, len(starts))
                    end = starts[m - 1] + sizes[m - 1] + 1
                    step = 256 << 20
                    for o in range(0, end, step):
                        f.write(ln.h.dev_download(d_in + o, min(step, end - o)))
                    total += end; left -= m; k += 1
                    if k >= ln.n_res and left > 0:
                        k = 0          # fewer distinct shards than asked for: the sample repeats them
                return total
            line["cpu_baseline"], dec_cpu, crc_cpu = cpu_baseline(write_sample, args.dna, args.qua)
            if decode_line and dec_cpu:
                line["decompress"]["cpu_baseline"] = dec_cpu
            # ---- the forms a user calls (task 6): verify, queue form, the CLI end to end -------------------
            if not os.environ.get("DSRC_BENCH_NO_FORMS"):
                try:
                    for l2 in lanes:
                        if getattr(l2.h, "h", None):
                            l2.h.release_memory()            # the headline's arenas (4 x 37 GB), the decoding passes' arena and model tables
                    v1, nb = measure_verify(cfg, ln, 1)
                    v4, _ = measure_verify(cfg, ln, 4)
                    # round 6: one handle, one call of the step's size: the lanes inside the handle compress, then ONE verifying pass decodes all
                    # the call's blocks (a pass is a chain per block: about as long for 1800 blocks as for 450)
                    vb, nbb = measure_verify(cfg, ln, 1, blocks=P * sub_blocks)
                    big = {"value": vb, "unit": "MB/s", "blocks_per_call": nbb, "instances": 1}
                    line["verify"] = {"value": v4, "unit": "MB/s", "blocks_per_call": nb, "instances": 4, "one_instance": v1, "larger_calls": big,
                                      "what": "-d3 -q2 -c: dsrcgpu_compress_batch_device with calculate_crc32 + verify_after_compress (the blocks are decoded on the device and the three CRC-32 compared), inputs and outputs in HBM",
                                      "cpu_baseline": crc_cpu}
                    d_in, starts, sizes = ln.shard(0)
                    n_host = min(len(starts), 192)
                    chunks = [ln.h.dev_download(d_in + starts[i], sizes[i]) for i in range(n_host)]
                    q = {}
                    for nh in (1, 2):
                        mbs, nbytes, dt = measure_queue_form(cfg, ln.h.device, chunks, nh, batches=12)
                        q[f"handles_{nh}"] = {"value": mbs, "unit": "MB/s", "bytes": nbytes, "s": round(dt, 2)}
                    mbs, nbytes, dt = measure_queue_form(cfg, ln.h.device, chunks, 1, batches=24, pinned=True)
                    q["handles_1_pinned"] = {"value": mbs, "unit": "MB/s", "bytes": nbytes, "s": round(dt, 2),
                                             "what": "the same with dsrcgpu_submit_pinned: the chunks lie in page-locked memory of the caller's and are not copied into the ring"}
                    q["what"] = "dsrcgpu_submit / flush / collect / release with host-resident 8 MiB chunks, 192 chunks per flush, one submitting and one collecting thread per handle; host copy into the page-locked ring, PCIe both ways and the compression inside; a handle runs consecutive batches on three scheduler lanes of its own (DSRC_GPU_QUEUE_LANES)"
                    line["queue_form"] = q
                    del chunks
                    # every GPU resource of the headline run is released; the same step on four-level qualities
                    dev0 = ln.h.device
                    for l2 in lanes:
                        l2.free()
                    line["one_handle"] = measure_one_handle(cfg, dev0, P * sub_blocks)
                    line["binned"] = measure_binned(cfg, dev0, sub_blocks, P)
                    # the CLI on configs[2]'s own 37.7 GB file in tmpfs
                    import tempfile
                    e2e_reads = int(os.environ.get("DSRC_BENCH_E2E_READS", "100000000"))
                    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
                        src = os.path.join(td, "e2e.fastq")
                        size = write_config3_file(src, dev0, e2e_reads)
                        line["host_e2e"] = measure_host_e2e(src, size, td)
                except Exception as e:          # noqa: BLE001  (secondary measurements must not take the headline line with them)
                    line["forms_error"] = repr(e)
        out_line = json.dumps(line)
    for ln in lanes:
        ln.h.close()          # idempotent
    if dist is not None:
        dist.destroy_process_group()
    if out_line is not None:
        # RCCL's version banner sits in C stdio's buffer until exit: push it out first so that the JSON line is the last
        # line of stdout
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(out_line, flush=True)


if __name__ == "__main__":
    main()
