"""The decoding half of the oracle (oracle/dsrc_oracle_dec.c) against the live reference build: whatever the reference's
BlockCompressor::Store writes, orc_decompress_block must turn into the same text as BlockCompressor::Read
(reference src/BlockCompressor.cpp:262-297).  Skipped where oracle/_ref was never built; tests/test_oracle_golden.py
covers that case with committed vectors."""
import dataclasses

import pytest

from dsrc_amd import synth
from tests._oracle import Config
from tests.cases import LEVELS, TINY, fuzz_fastq, fuzz_solid, rle_chunks


def canon(data: bytes) -> bytes:
    """What a lossless round trip gives back: LF line ends, final newline."""
    return data.replace(b"\r\n", b"\n").replace(b"\r", b"\n") + b"\n"


def _roundtrip(oracle, ref, cfg, data, what, cap=None):
    try:
        blk = oracle.compress_block(cfg, data)[0]
    except RuntimeError as e:
        assert "rc=-2" in str(e)              # reference UB on the encoding side
        return None
    cap = cap or len(data) + 64
    want = ref.decompress_block(cfg, blk, cap)
    try:
        got = oracle.decompress_block(cfg, blk, cap)
    except RuntimeError as e:
        # blocks the reference's own decoder cannot read back: DnaModelerHuffman codes a non-prefix-closed alphabet with
        # the wrong frequencies / codes, so decoding runs off the end of the block into stale
        # memory.  Refused by the oracle; the reference's text is then not the input either.
        assert "rc=-3" in str(e) and cfg.dna_order == 0 and want != canon(data), what
        return None
    assert got == want, what
    return got


@pytest.mark.parametrize("seed", range(100, 160))
def test_fuzz_blocks_decode(oracle, ref, seed):
    data, desc = fuzz_fastq(seed)
    for d, q, lossy, crc in LEVELS:
        cfg = Config.from_levels(d, q, lossy, crc)
        got = _roundtrip(oracle, ref, cfg, data, (seed, desc, d, q, lossy, crc))
        if got is not None and not lossy:
            assert got == canon(data), (seed, desc, d, q, crc)


def test_tiny_and_synthetic(oracle, ref):
    for data in (TINY, synth.illumina_fastq(4000)[:-1], synth.iontorrent_fastq(3000)[:-1]):
        for d, q, lossy, crc in LEVELS:
            cfg = Config.from_levels(d, q, lossy, crc)
            got = _roundtrip(oracle, ref, cfg, data, (d, q, lossy, crc))
            if got is not None and not lossy:
                assert got == canon(data)


def test_plus_repetition(oracle, ref):
    recs = [b"@id.%d x:%d\nACGTNACGT\n+id.%d x:%d\nIIII#IIII" % (i, i * 3, i, i * 3) for i in range(50)]
    data = b"\n".join(recs)
    for d, q, lossy, crc in LEVELS[:5]:
        cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), plus_repetition=True)
        got = _roundtrip(oracle, ref, cfg, data, (d, q, lossy, crc))
        if not lossy:
            assert got == data + b"\n"


def test_rle_alphabets_decode(oracle, ref):
    for data in rle_chunks():
        data = b"\n".join(data.split(b"\n")[: 4 * 400])
        cfg = Config.from_levels(0, 0)
        assert _roundtrip(oracle, ref, cfg, data, "rle") == data + b"\n"


@pytest.mark.parametrize("seed", range(300, 312))
def test_field_filter_decode(oracle, ref, seed):
    data, desc = fuzz_fastq(seed)
    for flags in (0b10, 0b1010, 0x7FFFFFFE):
        for d, q, lossy, crc in [(0, 0, False, True), (1, 1, False, False), (2, 1, True, True)]:
            cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), tag_flags=flags)
            _roundtrip(oracle, ref, cfg, data, (seed, desc, bin(flags), d, q, lossy, crc), cap=2 * len(data) + 4096)


@pytest.mark.parametrize("seed", range(40))
def test_color_space_decode(oracle, ref, seed):
    """Constant-primer colour-space blocks round-trip; blocks whose records start with different primers are undefined
    in the reference's decoder (src/RecordsProcessor.cpp:297-313 looks a character up in the index table) and refused."""
    data, desc = fuzz_solid(seed)
    for d, q, lossy, crc in LEVELS:
        cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), color_space=True)
        try:
            blk = oracle.compress_block(cfg, data)[0]
        except RuntimeError as e:
            assert "rc=-2" in str(e)
            continue
        cap = len(data) + 64
        if not desc[0]:                               # variable primer
            with pytest.raises(RuntimeError, match="rc=-2"):
                oracle.decompress_block(cfg, blk, cap)
            continue
        want = ref.decompress_block(cfg, blk, cap)
        assert oracle.decompress_block(cfg, blk, cap) == want, (seed, desc, d, q, lossy, crc)
        if not lossy:
            # the decoder gives EVERY record the primer quality of record 0 (ChunkHeader::csQuaBegin,
            # src/RecordsProcessor.cpp:297-301); otherwise the text comes back
            lines = data.split(b"\n")
            q0 = lines[3][:1]
            for i in range(3, len(lines), 4):
                lines[i] = q0 + lines[i][1:]
            assert want == b"\n".join(lines) + b"\n"


def test_verify_checksum(oracle):
    data = synth.illumina_fastq(300)[:-1]
    for d, q, lossy in [(0, 0, False), (3, 2, False), (2, 1, True)]:
        cfg = Config.from_levels(d, q, lossy, True)
        blk = bytearray(oracle.compress_block(cfg, data)[0])
        assert oracle.verify_block(cfg, bytes(blk), len(data) + 64) == 1
        text, st, ac = oracle.decompress_block(cfg, bytes(blk), len(data) + 64, with_crc=True)
        assert st[1] == ac[1] and (lossy or st[2] == ac[2]) and st[0] == ac[0]
        blk[20] ^= 1                                  # inside the stored CRC words (meta stream: 16 bytes, then tag/seq/qual CRCs)
        assert oracle.verify_block(cfg, bytes(blk), len(data) + 64) == 0
