"""GPU decompression parity: dsrcgpu_decompress_batch[_device] against the oracle's decoder, which is pinned
to the reference's BlockCompressor::Read, and against the committed decode golden vectors of the reference itself.
Bit-exact bar: bytes of the decoded chunk text."""
import dataclasses
import hashlib
import os

import numpy as np
import pytest

from dsrc_amd import synth
from tests._oracle import Config
from tests.cases import LEVELS, TINY, fuzz_fastq, fuzz_solid, rle_chunks

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    os.environ.pop("DSRC_GPU_LIB", None)
    from dsrc_amd import _lib
    _lib._lib = None
    return _lib


def sha(b):
    return hashlib.sha256(b).hexdigest()


def handle(gpu, cfg):
    return gpu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset,
                      plus_repetition=cfg.plus_repetition, color_space=cfg.color_space, tag_flags=cfg.tag_flags)


def check(gpu, oracle, cfg, chunks, what=None):
    """Blocks written by the (reference-pinned) oracle encoder for `chunks` -> GPU text == oracle text, verdicts equal."""
    try:
        blocks = [b for b, _, _ in oracle.compress_blocks_state(cfg, chunks)]
    except RuntimeError as e:
        assert "rc=-2" in str(e)
        return 0
    want = []
    for b, c in zip(blocks, chunks):
        try:
            want.append(oracle.decompress_block(cfg, b, 2 * len(c) + 4096))
        except RuntimeError:
            want.append(None)
    h = handle(gpu, cfg)
    try:
        if any(w is None for w in want):
            for b, w in zip(blocks, want):
                if w is None:
                    with pytest.raises(gpu.DsrcGpuError):
                        h.decompress_batch([b])
                else:
                    assert h.decompress_batch([b]) == [w], what
        else:
            got, ok = h.decompress_batch(blocks, verify=True)
            assert got == want, what
            assert ok == [oracle.verify_block(cfg, b, 2 * len(c) + 4096) if cfg.crc else 1 for b, c in zip(blocks, chunks)], what
    finally:
        h.close()
    return len(blocks)


@pytest.mark.parametrize("d,q,lossy,crc", LEVELS)
def test_tiny_and_illumina(gpu, oracle, d, q, lossy, crc):
    cfg = Config.from_levels(d, q, lossy, crc)
    check(gpu, oracle, cfg, [TINY])
    check(gpu, oracle, cfg, [synth.illumina_fastq(20000)[:-1], synth.illumina_fastq(3000, first=777)[:-1]])


def test_release_memory_between_phases(gpu, oracle):
    """dsrcgpu_release_memory: arena and table region go back to the device, the next call allocates again; the block-to-block
    state of the compressor survives."""
    chunks = [synth.illumina_fastq(3000, first=1 + 3000 * k)[:-1] for k in range(3)]
    cfg = Config.from_levels(3, 2, False, True)
    want = [b[0] for b in oracle.compress_blocks_state(cfg, chunks + chunks)]
    h = gpu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset)
    try:
        got = [g[0] for g in h.compress_batch(chunks)]
        h.release_memory()
        got += [g[0] for g in h.compress_batch(chunks)]
        assert got == want
        texts = h.decompress_batch(got[:3])
        h.release_memory(); h.release_memory()
        assert texts == h.decompress_batch(got[:3]) == [c + b"\n" for c in chunks]
    finally:
        h.close()


@pytest.mark.parametrize("d,q,lossy,crc", [(2, 1, True, False), (0, 0, False, False), (0, 2, False, False), (0, 1, False, True)])
def test_iontorrent(gpu, oracle, d, q, lossy, crc):
    check(gpu, oracle, Config.from_levels(d, q, lossy, crc), [synth.iontorrent_fastq(5000)[:-1]])


@pytest.mark.parametrize("seed", range(60))
def test_fuzz(gpu, oracle, seed):
    data, desc = fuzz_fastq(seed)
    for d, q, lossy, crc in LEVELS:
        check(gpu, oracle, Config.from_levels(d, q, lossy, crc), [data], (seed, desc, d, q, lossy, crc))


@pytest.mark.parametrize("seed", range(24))
def test_color_space(gpu, oracle, seed):
    data, desc = fuzz_solid(seed)
    for d, q, lossy, crc in LEVELS:
        cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), color_space=True)
        check(gpu, oracle, cfg, [data], (seed, desc, d, q, lossy, crc))


@pytest.mark.parametrize("seed", range(400, 420))
def test_field_filter(gpu, oracle, seed):
    data, desc = fuzz_fastq(seed, [None, 300, 3000][seed % 3])
    for flags in (0b10, 0b1010, 0x7FFFFFFE):
        for d, q, lossy, crc in [(0, 0, False, True), (3, 2, False, False), (2, 1, True, True)]:
            cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), tag_flags=flags)
            check(gpu, oracle, cfg, [data], (seed, desc, bin(flags), d, q, lossy, crc))


def test_plus_repetition_and_rle(gpu, oracle):
    recs = [b"@id.%d x:%d\nACGTNACGT\n+id.%d x:%d\nIIII#IIII" % (i, i * 3, i, i * 3) for i in range(5000)]
    for d, q, lossy, crc in LEVELS[:5]:
        check(gpu, oracle, dataclasses.replace(Config.from_levels(d, q, lossy, crc), plus_repetition=True), [b"\n".join(recs)])
    for d, q in ((0, 0), (2, 0)):
        check(gpu, oracle, Config.from_levels(d, q), rle_chunks())


def test_reference_golden_vectors(gpu, oracle):
    """What the unmodified reference's BlockCompressor::Read returned for the golden blocks (tests/golden/decode_golden.json)."""
    from tests.decode_cases import cases
    n = refused = 0
    handles = {}
    for solid in (False, True):
        for label, cfg, blk, cap, x in cases(oracle, solid):
            key = dataclasses.astuple(cfg)
            if key not in handles:
                handles[key] = handle(gpu, cfg)
            h = handles[key]
            if x.get("refused"):
                with pytest.raises(gpu.DsrcGpuError):
                    h.decompress_batch([blk])
                refused += 1
                continue
            text = h.decompress_batch([blk])[0]
            assert (len(text), sha(text)) == (x["text_size"], x["text_sha256"]), label
            n += 1
    for h in handles.values():
        h.close()
    assert n > 700 and refused > 0


def test_corrupt_and_mismatched_blocks(gpu, oracle):
    data = synth.illumina_fastq(2000)[:-1]
    cfg = Config.from_levels(0, 0, False, True)
    blk = bytearray(oracle.compress_block(cfg, data)[0]); blk[20] ^= 1        # a stored CRC word
    h = handle(gpu, cfg)
    texts, ok = h.decompress_batch([bytes(blk)], verify=True)
    assert ok == [0] and texts[0] == data + b"\n"
    with pytest.raises(gpu.DsrcGpuError):
        h.decompress_batch([bytes(blk[: len(blk) // 2])])                      # truncated
    h.close()
    h = handle(gpu, Config.from_levels(3, 2))
    with pytest.raises(gpu.DsrcGpuError):
        h.decompress_batch([bytes(blk)])                                      # settings of another archive
    h.close()


def _bench_helpers():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("d,q,crc", [(3, 2, True), (0, 0, False)])
def test_full_size_round_trip_on_device(gpu, d, q, crc):
    """Encode -> decode at BASELINE block size, everything resident in HBM: 24 chunks of 8 MiB generated on the device,
    compressed by the HIP path, decompressed by it, and the text must be the input (lossless) with every checksum
    verdict 1 -- the property the reference's `-c` worker checks (src/DsrcWorker.cpp:53-62)."""
    bench = _bench_helpers()
    nblocks = 24
    cfg = Config.from_levels(d, q, False, crc)
    h = handle(gpu, cfg)
    recs = int(nblocks * bench.RECS_PER_BLOCK * 1.02) + 1000
    cap = recs * 384
    d_in = h.dev_alloc(cap); d_blk = h.dev_alloc(cap // 2); d_txt = h.dev_alloc(cap)
    first = 424242
    nbytes = h.synth_illumina(first, recs, d_in, cap)
    off = bench.record_offsets(first, recs)
    assert off[-1] == nbytes
    starts, sizes = bench.cut_blocks(off, nblocks)
    o_offs, o_sizes, raw, comp = h.compress_batch_device(d_in, starts, sizes, d_blk, cap // 2)
    t_offs, t_sizes, ok = h.decompress_batch_device(d_blk, o_offs, o_sizes, d_txt, cap, verify=True)
    assert ok == [1] * nblocks
    assert t_sizes == [s + 1 for s in sizes]
    for i in range(nblocks):
        a = np.frombuffer(h.dev_download(d_in + starts[i], sizes[i]), dtype=np.uint8)
        b = np.frombuffer(h.dev_download(d_txt + t_offs[i], t_sizes[i]), dtype=np.uint8)
        assert b[-1] == 10 and np.array_equal(a, b[:-1]), f"block {i}"
    print(f"decode -d{d} -q{q}: {h.last_timing()[0]:.1f} ms for {nblocks} blocks")
    for p in (d_in, d_blk, d_txt):
        h.dev_free(p)
    h.close()
