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
