"""Parity at BASELINE block size through size-independent properties (the oracle is too slow for whole data sets):
a batch of full 8 MiB chunks generated in HBM is compressed in one scheduler pass, then
  * every block's header must carry the exact record count / chunk size of its chunk,
  * the reference's own decoder (oracle/_ref, BlockCompressor::Read) must reproduce sampled chunks byte for byte,
  * sampled blocks must equal the oracle's bytes (with the compressor state carried in chunk order),
  * a block's bytes must not depend on which other chunks share the batch (apart from that documented state)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from tests._oracle import Config, _orc_cfg, have_ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    os.environ.pop("DSRC_GPU_LIB", None)
    from dsrc_amd import _lib
    _lib._lib = None
    return _lib


def _bench_helpers():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("d,q,binned", [(3, 2, False), (0, 0, False), (3, 2, True), (0, 0, True)])
def test_full_size_blocks(gpu, oracle, d, q, binned):
    """24 full 8 MiB chunks of BASELINE's generator through the device entry point: header words of every block, sampled blocks
    against the oracle (with the carried state) and the reference's decoder.  binned: the same records with four-level qualities
    (flavour 1 of dsrcgpu_synth_fastq, bench.py's second line): a third of a quality stream in one context at -q2 (hot buckets,
    Rescale() inside k_model), the RLE scheme at -q0."""
    bench = _bench_helpers()
    nblocks = 24
    cfg = Config.from_levels(d, q)
    h = gpu.Handle(cfg.dna_order, cfg.quality_order)
    recs = int(nblocks * bench.RECS_PER_BLOCK * 1.02) + 1000
    cap = recs * 384
    d_in = h.dev_alloc(cap); d_out = h.dev_alloc(cap // 2)
    first = 123456789
    nbytes = h.synth_illumina(first, recs, d_in, cap, binned=binned)
    off = bench.record_offsets(first, recs)
    assert off[-1] == nbytes
    starts, sizes = bench.cut_blocks(off, nblocks)
    if binned:          # the device generator's flavour 1 is synth.illumina_fastq(binned=True)
        from dsrc_amd import synth
        assert h.dev_download(d_in, 200000) == synth.illumina_fastq(1200, first=first, binned=True)[:200000]
    o_offs, o_sizes, raw, comp = h.compress_batch_device(d_in, starts, sizes, d_out, cap // 2)
    blob = h.dev_download(d_out, o_offs[-1] + o_sizes[-1])
    # header words: recordsCount, maxQuaLength, flags, chunkSize (StoreMetaData, reference src/BlockCompressor.cpp:403-443)
    rec_starts = np.searchsorted(off, np.array(starts + [starts[-1] + sizes[-1] + 1]))
    for i in range(nblocks):
        n_recs, max_len, flags, chunk_size = struct.unpack(">IIII", blob[o_offs[i]: o_offs[i] + 16])
        assert n_recs == rec_starts[i + 1] - rec_starts[i]
        assert (max_len, flags, chunk_size) == (150, 0, sizes[i])
        assert sum(comp[4 * i: 4 * i + 4]) == o_sizes[i] and raw[4 * i + 2] == raw[4 * i + 3] == 150 * n_recs
    assert all(o_offs[i] + o_sizes[i] == o_offs[i + 1] for i in range(nblocks - 1))
    sample = [0, nblocks // 2, nblocks - 1]
    chunks = {i: h.dev_download(d_in + starts[i], sizes[i]) for i in sample}
    if have_ref():
        from tests._oracle import Ref
        r = Ref()
        for i in sample:
            back = r.decompress_block(cfg, blob[o_offs[i]: o_offs[i] + o_sizes[i]], sizes[i] + 64)
            assert back == chunks[i] + b"\n"
    # oracle bytes, carrying the reference's block-to-block compressor state in chunk order
    capst = C.c_uint32(0); c = _orc_cfg(cfg)
    for i in range(nblocks):
        if i not in sample and i > 1:
            continue
        ch = chunks[i] if i in chunks else h.dev_download(d_in + starts[i], sizes[i])
        out = (C.c_uint8 * (len(ch) + 65536))(); osz = C.c_uint64(); r4 = (C.c_uint64 * 4)(); c4 = (C.c_uint64 * 4)()
        assert oracle.lib.orc_compress_block_state(C.byref(c), C.byref(capst), ch, C.c_uint64(len(ch)), out, C.c_uint64(len(out)), C.byref(osz), r4, c4) == 0
        assert blob[o_offs[i]: o_offs[i] + o_sizes[i]] == bytes(out[:osz.value]), f"block {i}"
    # batch independence: the same chunk in a different batch position (state already warmed up) gives the same bytes
    h2 = gpu.Handle(cfg.dna_order, cfg.quality_order)
    o2 = h2.compress_batch_device(d_in, [starts[0], starts[5], starts[3]], [sizes[0], sizes[5], sizes[3]], d_out, cap // 2)
    blob2 = h2.dev_download(d_out, o2[0][-1] + o2[1][-1])
    assert blob2[o2[0][1]: o2[0][1] + o2[1][1]] == blob[o_offs[5]: o_offs[5] + o_sizes[5]]
    assert blob2[o2[0][2]: o2[0][2] + o2[1][2]] == blob[o_offs[3]: o_offs[3] + o_sizes[3]]
    h2.close()
    h.dev_free(d_in); h.dev_free(d_out); h.close()


@pytest.mark.parametrize("binned", [False, True])
def test_full_size_blocks_b64(gpu, oracle, binned):
    """Two 64 MiB chunks (`-b64`, the reference's -m1 preset, src/main.cpp:195-219): 27 M symbols per stream, 3260 tiles -- k_model reads a
    bucket's tile table from the count table 64 tiles at a time (round 6; until then such streams went through k_sort / k_replay).
    With four-level qualities the hot buckets outgrow what one wave may walk: those streams are handed back on the device and coded by
    the redo list's kernels.  Both blocks against the oracle, the state carried."""
    bench = _bench_helpers()
    cfg = Config.from_levels(3, 2)
    h = gpu.Handle(cfg.dna_order, cfg.quality_order)
    per = bench.RECS_PER_BLOCK * 8
    recs = int(2 * per * 1.02) + 1000
    cap = recs * 384
    d_in = h.dev_alloc(cap); d_out = h.dev_alloc(cap // 2)
    first = 4242
    nbytes = h.synth_illumina(first, recs, d_in, cap, binned=binned)
    off = bench.record_offsets(first, recs)
    assert off[-1] == nbytes
    cuts = [0, per, 2 * per]
    starts = [int(off[a]) for a in cuts[:-1]]; sizes = [int(off[b] - off[a] - 1) for a, b in zip(cuts[:-1], cuts[1:])]
    assert min(sizes) > 60 << 20
    o_offs, o_sizes, raw, comp = h.compress_batch_device(d_in, starts, sizes, d_out, cap // 2)
    blob = h.dev_download(d_out, o_offs[-1] + o_sizes[-1])
    capst = C.c_uint32(0); c = _orc_cfg(cfg)
    for i in range(2):
        ch = h.dev_download(d_in + starts[i], sizes[i])
        out = (C.c_uint8 * (len(ch) + 65536))(); osz = C.c_uint64(); r4 = (C.c_uint64 * 4)(); c4 = (C.c_uint64 * 4)()
        assert oracle.lib.orc_compress_block_state(C.byref(c), C.byref(capst), ch, C.c_uint64(len(ch)), out, C.c_uint64(len(out)), C.byref(osz), r4, c4) == 0
        assert blob[o_offs[i]: o_offs[i] + o_sizes[i]] == bytes(out[:osz.value]), f"chunk {i} ({sizes[i]} bytes)"
    h.dev_free(d_in); h.dev_free(d_out); h.close()


def test_large_chunks_and_mixed_sizes(gpu, oracle):
    """Chunk sizes other than the default 8 MiB in one batch (a 40 MB chunk = `-b40`, a 1 MB one, a 300-byte one):
    the range-coder stage lays the streams of a wave out with the longest stream's pitch."""
    bench = _bench_helpers()
    cfg = Config.from_levels(3, 2)
    h = gpu.Handle(cfg.dna_order, cfg.quality_order)
    recs = 125000
    cap = recs * 384
    d_in = h.dev_alloc(cap); d_out = h.dev_alloc(cap)
    first = 7
    nbytes = h.synth_illumina(first, recs, d_in, cap)
    off = bench.record_offsets(first, recs)
    assert off[-1] == nbytes
    cuts = [0, 105000, 107800, 107801, 125000]                  # records per chunk: 105000 (~40 MB), 2800, 1, 17199
    starts = [int(off[a]) for a in cuts[:-1]]; sizes = [int(off[b] - off[a] - 1) for a, b in zip(cuts[:-1], cuts[1:])]
    o_offs, o_sizes, raw, comp = h.compress_batch_device(d_in, starts, sizes, d_out, cap)
    blob = h.dev_download(d_out, o_offs[-1] + o_sizes[-1])
    capst = C.c_uint32(0); c = _orc_cfg(cfg)
    for i in range(len(starts)):
        ch = h.dev_download(d_in + starts[i], sizes[i])
        out = (C.c_uint8 * (len(ch) + 65536))(); osz = C.c_uint64(); r4 = (C.c_uint64 * 4)(); c4 = (C.c_uint64 * 4)()
        assert oracle.lib.orc_compress_block_state(C.byref(c), C.byref(capst), ch, C.c_uint64(len(ch)), out, C.c_uint64(len(out)), C.byref(osz), r4, c4) == 0
        assert blob[o_offs[i]: o_offs[i] + o_sizes[i]] == bytes(out[:osz.value]), f"chunk {i} ({sizes[i]} bytes)"
    h.dev_free(d_in); h.dev_free(d_out); h.close()
