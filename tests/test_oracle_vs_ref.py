"""The oracle against the live reference build (oracle/_ref), when it is present.  Skipped on machines
where the reference was never built; the committed golden vectors cover that case."""
import pytest

from dsrc_amd import synth
from tests._oracle import Config
from tests.cases import LEVELS, fuzz_fastq


@pytest.mark.parametrize("seed", range(100, 130))
def test_fuzz_blocks(oracle, ref, seed):
    data, desc = fuzz_fastq(seed)
    for d, q, lossy, crc in LEVELS[:6]:
        cfg = Config.from_levels(d, q, lossy, crc)
        try:
            a = oracle.compress_block(cfg, data)
        except RuntimeError as e:
            assert "rc=-2" in str(e)          # reference UB: never handed to the reference (it may corrupt its heap)
            continue
        assert a == ref.compress_block(cfg, data), (seed, desc, d, q, lossy, crc)


def test_medium_illumina(oracle, ref):
    data = synth.illumina_fastq(6000)[:-1]
    for d, q, lossy, crc in [(3, 2, False, False), (0, 0, False, False), (2, 1, True, False)]:
        cfg = Config.from_levels(d, q, lossy, crc)
        assert oracle.compress_block(cfg, data) == ref.compress_block(cfg, data)


def test_roundtrip_through_reference_decoder(oracle, ref):
    """What the oracle writes decodes with the reference's own BlockCompressor::Read."""
    data = synth.illumina_fastq(500)[:-1]
    cfg = Config.from_levels(3, 2)
    blk = oracle.compress_block(cfg, data)[0]
    assert ref.decompress_block(cfg, blk, len(data) + 16) == data + b"\n"


def test_analyze(oracle, ref):
    for data in (synth.illumina_fastq(50), synth.iontorrent_fastq(50)):
        assert oracle.analyze(data[:-1]) == ref.analyze(data[:-1])


@pytest.mark.parametrize("seed", range(300, 320))
def test_field_filter_blocks(oracle, ref, seed):
    """`-f` (FastqParserExt, reference src/FastqParser.cpp:167-251): the oracle restates it -- including the kept last
    field swallowing the line terminator and the tokenizer then seeing the first, already index-transformed base --
    (SURVEY 8a-2)."""
    import dataclasses
    data, desc = fuzz_fastq(seed)
    for flags in (0b10, 0b1010, 0b11110, 0x7FFFFFFE):
        for d, q, lossy, crc in [(0, 0, False, True), (1, 1, False, False), (2, 1, True, True)]:
            cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), tag_flags=flags)
            try:
                a = oracle.compress_block(cfg, data)
            except RuntimeError as e:
                assert "rc=-2" in str(e)
                continue
            assert a == ref.compress_block(cfg, data), (seed, desc, bin(flags), d, q, lossy, crc)


@pytest.mark.parametrize("seed", range(40))
def test_color_space_blocks(oracle, ref, seed):
    """SOLiD colour space (src/RecordsProcessor.cpp:25-58, src/BlockCompressor.cpp:184-199,380-393,415-422): colours to
    bases before the index transform; with a constant primer the records lose their first base / quality AFTER the
    statistics were taken, and the meta stream carries csSeqBegin / csQuaBegin."""
    import dataclasses
    from tests.cases import fuzz_solid
    data, desc = fuzz_solid(seed)
    for d, q, lossy, crc in LEVELS:
        cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), color_space=True)
        try:
            a = oracle.compress_block(cfg, data)
        except RuntimeError as e:
            assert "rc=-2" in str(e)
            continue
        assert a == ref.compress_block(cfg, data), (seed, desc, d, q, lossy, crc)
