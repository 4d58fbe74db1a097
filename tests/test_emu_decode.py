"""Decoder kernel logic on the CPU: dsrc_amd/csrc/k_dec.h compiled against the HIP emulator (tests/emu) and driven
through dsrcgpu_decompress_batch, against the oracle's decoder (pinned to the reference's BlockCompressor::Read).
Test harness for a GPU-less container; the product loads only libdsrc_gpu.so."""
import dataclasses
import os

import pytest

from dsrc_amd import synth
from tests._oracle import Config
from tests.cases import LEVELS, TINY, fuzz_fastq, fuzz_solid
from tests.test_emu_kernels import emu  # noqa: F401  (fixture)


@pytest.fixture(autouse=True)
def one_lane_quality_decoder(request, monkeypatch):
    """The wave-cooperative range decoder costs the emulator 64 context switches per symbol; most tests here use the
    one-lane form of the same decoder (DSRC_GPU_DEC_SERIAL) and test_wave_decoder covers the cooperative one."""
    if "wave" not in request.node.name and not os.environ.get("DSRC_TEST_EMU_FAST_DECODER"):
        monkeypatch.setenv("DSRC_GPU_DEC_SERIAL", "1")


def handle(emu, cfg):
    return emu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset,
                      plus_repetition=cfg.plus_repetition, color_space=cfg.color_space, tag_flags=cfg.tag_flags)


def check(emu, oracle, cfg, chunks, what=None):
    blocks = []
    for c in chunks:
        try:
            blocks.append(oracle.compress_block(cfg, c)[0])
        except RuntimeError as e:
            assert "rc=-2" in str(e)
            return
    want = []
    for b, c in zip(blocks, chunks):
        try:
            want.append(oracle.decompress_block(cfg, b, 2 * len(c) + 4096))
        except RuntimeError:
            want.append(None)                    # undefined in the reference's decoder: must be refused
    h = handle(emu, cfg)
    try:
        if any(w is None for w in want):
            for b, w in zip(blocks, want):
                if w is None:
                    with pytest.raises(emu.DsrcGpuError):
                        h.decompress_batch([b])
                else:
                    assert h.decompress_batch([b]) == [w], what
        else:
            got, ok = h.decompress_batch(blocks, verify=True)
            assert got == want, what
            # the verdict of VerifyChecksum (0 where the reference's own round trip is not the identity, e.g. DNA Huffman
            # over a non-prefix-closed alphabet)
            assert ok == [oracle.verify_block(cfg, b, 2 * len(c) + 4096) if cfg.crc else 1 for b, c in zip(blocks, chunks)], what
    finally:
        h.close()


@pytest.mark.parametrize("d,q,lossy,crc", LEVELS)
def test_tiny(emu, oracle, d, q, lossy, crc):
    check(emu, oracle, Config.from_levels(d, q, lossy, crc), [TINY])


@pytest.mark.parametrize("d,q,lossy,crc", [(3, 2, False, True), (0, 0, False, False), (2, 1, True, False), (0, 1, False, False)])
def test_batches(emu, oracle, d, q, lossy, crc):
    chunks = [synth.illumina_fastq(90, first=1 + 90 * k)[:-1] for k in range(3)] + [synth.iontorrent_fastq(50)[:-1]]
    if not lossy and d > 0:
        chunks = chunks[:3]                      # lossless order-k DNA on IUPAC data is undefined in the reference
    check(emu, oracle, Config.from_levels(d, q, lossy, crc), chunks)


@pytest.mark.parametrize("seed", range(36))
def test_fuzz(emu, oracle, seed):
    data, desc = fuzz_fastq(seed, nrec=[2, 3, 10, 60, 150][seed % 5])
    for d, q, lossy, crc in [(0, 0, False, True), (3, 2, False, False), (2, 1, True, True), (1, 1, False, False), (0, 0, True, False)]:
        check(emu, oracle, Config.from_levels(d, q, lossy, crc), [data], (seed, desc, d, q, lossy, crc))


@pytest.mark.parametrize("seed", range(12))
def test_color_space(emu, oracle, seed):
    data, desc = fuzz_solid(seed, nrec=[2, 5, 40, 120][seed % 4])
    for d, q, lossy, crc in LEVELS[:5]:
        cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), color_space=True)
        check(emu, oracle, cfg, [data], (seed, desc, d, q, lossy, crc))


def test_plus_repetition_and_filter(emu, oracle):
    recs = [b"@id.%d x:%d\nACGTNACGT\n+id.%d x:%d\nIIII#IIII" % (i, i * 3, i, i * 3) for i in range(70)]
    data = b"\n".join(recs)
    for d, q, lossy, crc in LEVELS[:4]:
        check(emu, oracle, dataclasses.replace(Config.from_levels(d, q, lossy, crc), plus_repetition=True), [data])
    data2, _ = fuzz_fastq(301, nrec=80)
    for flags in (0b10, 0b1010):
        check(emu, oracle, dataclasses.replace(Config.from_levels(0, 0, False, True), tag_flags=flags), [data2])


def test_crc_mismatch_is_reported(emu, oracle):
    data = synth.illumina_fastq(40)[:-1]
    cfg = Config.from_levels(0, 0, False, True)
    blk = bytearray(oracle.compress_block(cfg, data)[0]); blk[20] ^= 1
    h = handle(emu, cfg)
    texts, ok = h.decompress_batch([bytes(blk)], verify=True)
    h.close()
    assert ok == [0] and texts[0] == data + b"\n"


def test_wrong_settings_are_refused(emu, oracle):
    data = synth.illumina_fastq(40)[:-1]
    blk = oracle.compress_block(Config.from_levels(3, 2), data)[0]
    h = handle(emu, Config.from_levels(0, 0))
    with pytest.raises(emu.DsrcGpuError):
        h.decompress_batch([blk])
    h.close()


def test_verify_after_compress(emu, oracle):
    """calculate_crc32 + verify_after_compress: the compress call decodes what it wrote (reference: DsrcWorker.cpp:53-62).
    Passes on data that round-trips; fails with DSRCGPU_E_CRC where the reference's own verification fails (a DNA
    Huffman alphabet that is not prefix-closed)."""
    good = synth.illumina_fastq(60)[:-1]
    for d, q, lossy in [(0, 0, False), (3, 2, False), (2, 1, True)]:
        cfg = Config.from_levels(d, q, lossy, True)
        h = emu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, True, verify=True)
        assert h.compress_batch([good, good])[0] == oracle.compress_block(cfg, good)
        h.close()
    bad = b"@a\nWRWSG\n+\n#:#$,\n@b\nRCTAR\n+\n:.,#F"            # W, R, S present without all lower indices
    cfg = Config.from_levels(0, 0, False, True)
    blk = oracle.compress_block(cfg, bad)[0]
    assert oracle.verify_block(cfg, blk, 4096) != 1
    h = emu.Handle(0, 0, False, True, verify=True)
    with pytest.raises(emu.DsrcGpuError) as ei:
        h.compress_batch([bad])
    assert ei.value.code == -7 and "CRC32 checksums mismatch." in str(ei.value)
    h.close()


@pytest.mark.parametrize("d,q,lossy", [(0, 2, False), (1, 1, False), (2, 1, True), (0, 2, True)])
def test_wave_decoder(emu, oracle, d, q, lossy):
    """qua_order_decode_wave (lane i = counter i; scan, ballot): 16-, 32- and 64-symbol alphabets, lossy 8, variable
    lengths, and a context hot enough to rescale."""
    import random
    rng = random.Random(d * 10 + q)
    few = b"\n".join(b"@h.%d\nACGT\n+\n%s" % (i, bytes(rng.choice(b"IIIIIIIH") for _ in range(4))) for i in range(2200))     # > 8k symbols in few contexts
    wide = b"\n".join(b"@w.%d\n%s\n+\n%s" % (i, b"A" * (20 + i % 7), bytes(33 + rng.randrange(2, 62) for _ in range(20 + i % 7))) for i in range(60))
    chunks = [synth.illumina_fastq(40)[:-1], few] + ([] if lossy else [wide])
    check(emu, oracle, Config.from_levels(d, q, lossy), chunks)


@pytest.mark.parametrize("d,q,wide", [(3, 2, False), (1, 2, True), (2, 1, True)])
def test_wave_decoder_rescale(emu, oracle, d, q, wide):
    """Rows hot enough for TSymbolCoderRC::Rescale (src/SymbolCoderRC.h:69-90; > 32 k visits of one row) in the cooperative quality
    decoder -- which applies it when the row is written, not at the next visit -- with one counter per lane and, for the
    128-symbol alphabet (`wide`), two; and in the lane-per-block DNA decoder (one homopolymer context)."""
    import random
    rng = random.Random(7 * d + q)
    recs = [b"@r\nA\n+\n%c" % (73 if rng.random() < 0.97 else 72) for _ in range(42000)]        # reads of length 1: one position context
    if wide:
        recs += [b"@w\n%s\n+\n%s" % (b"C" * 70, bytes(range(33, 103)))]                          # 70 distinct qualities: 128-symbol alphabet
    recs += [b"@h\n%s\n+\n%s" % (b"A" * 200, b"I" * 200) for _ in range(180)]                  # 36 k bases in one DNA context
    check(emu, oracle, Config.from_levels(d, q, False), [b"\n".join(recs)])


@pytest.mark.parametrize("d,q,lossy,crc", LEVELS)
def test_wave_kernels_every_level(emu, oracle, d, q, lossy, crc):
    """The product's decode kernels (not the one-lane forms most tests here use for speed): k_dec_tags_wave, k_dec_qpos / k_dec_qhuff,
    k_dec_qrc, k_dec_dnarc, k_dec_dna0 over every level set, on blocks with constant and variable read lengths."""
    chunks = [TINY, synth.illumina_fastq(60)[:-1]]
    if lossy or d == 0:
        chunks.append(synth.iontorrent_fastq(40)[:-1])          # lossless order-k DNA on IUPAC data is undefined in the reference
    check(emu, oracle, Config.from_levels(d, q, lossy, crc), chunks)


@pytest.mark.parametrize("d,q,lossy", [(3, 2, False), (2, 1, True), (1, 1, False)])
def test_wave_kernels_mixed_lengths_and_hot_rows(emu, oracle, d, q, lossy):
    """k_dec_qrc / k_dec_dnarc on a batch of blocks of very different lengths (the lanes of k_dec_dnarc end at different times),
    and rows hot enough for Rescale()."""
    import random
    rng = random.Random(5 * d + q)
    hot = b"\n".join([b"@r\nA\n+\n%c" % (73 if rng.random() < 0.97 else 60) for _ in range(42000)] + [b"@h\n%s\n+\n%s" % (b"A" * 100, b"I" * 100) for _ in range(40)])
    chunks = [synth.illumina_fastq(40)[:-1], hot, synth.illumina_fastq(90, first=500)[:-1], TINY, synth.illumina_fastq(10, first=7)[:-1]]
    check(emu, oracle, Config.from_levels(d, q, lossy), chunks)


def test_wave_rle_scheme(emu, oracle):
    """The RLE scheme of -q0 (QualityRLEModeler::Decode) through k_dec_qpos's tables: 4-, 20- and 45-symbol alphabets, runs longer
    than 255 that cross record boundaries, and runs the records do not consume."""
    from tests.cases import rle_chunks
    chunks = [c[: c.index(b"\n@r.400 ")] for c in rle_chunks()]          # the first 400 records of each
    for lossy in (False, True):
        check(emu, oracle, Config.from_levels(0, 0, lossy, True), chunks)
