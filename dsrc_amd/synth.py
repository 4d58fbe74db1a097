"""Counter-based synthetic FASTQ generators (BASELINE.json configs).

Every byte of record ``i`` is a pure function of ``(seed, i, position)`` so any shard of a
data set can be produced independently (per block, per GPU).  All arithmetic is integer
(the Gaussian term is an 8-byte Irwin-Hall sum) so that the host generator here and the
device generator in ``csrc/k_synth`` agree bit for bit.

This is bench/test input only -- it is not part of the compression path.
"""
from __future__ import annotations

import numpy as np

SEED = 0xD5C0FFEE
MASK = (1 << 64) - 1


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays."""
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _byte_sum(h: np.ndarray) -> np.ndarray:
    s = np.zeros(h.shape, dtype=np.int64)
    for k in range(8):
        s += ((h >> np.uint64(8 * k)) & np.uint64(0xFF)).astype(np.int64)
    return s


def illumina_title(i: int) -> bytes:
    lane = 1 + ((i - 1) // 250000) % 8
    tile = 1101 + ((i - 1) // 5000) % 96
    x = 1000 + (7 * i) % 20000
    y = 2000 + (13 * i) % 90000
    return b"@SRRSYN.%d HWI-ST1234:100:C0ABCACXX:%d:%d:%d:%d 1:N:0:ATCACG" % (i, lane, tile, x, y)


def illumina_bases_quals(first: int, count: int, read_len: int = 150, seed: int = SEED, binned: bool = False):
    """(count, read_len) uint8 arrays of bases and Phred+33 qualities for records first..first+count-1.
    binned: the qualities quantised to four levels (2 / 12 / 23 / 37), as current instruments write them
    (flavour 1 of dsrcgpu_synth_fastq, csrc/k_synth.h)."""
    with np.errstate(over="ignore"):
        i = np.arange(first, first + count, dtype=np.uint64)[:, None]
        p = np.arange(read_len, dtype=np.uint64)[None, :]
        h = _mix64((i << np.uint64(10) | p) ^ np.uint64(seed))
        h2 = _mix64(h)
    base = np.frombuffer(b"ACGT", dtype=np.uint8)[(h & np.uint64(3)).astype(np.int64)]
    is_n = ((h >> np.uint64(2)) % np.uint64(500)) == 0
    s = _byte_sum(h2)
    z4 = np.floor_divide((s - 1020) * 4 + 104, 209)
    q = 38 - (6 * p.astype(np.int64)) // 100 + z4
    q = np.clip(q, 2, 40)
    if binned:
        q = np.where(q < 3, 2, np.where(q < 18, 12, np.where(q < 30, 23, 37)))
    base = np.where(is_n, np.uint8(ord("N")), base)
    q = np.where(is_n, 2, q)
    return base.astype(np.uint8), (q + 33).astype(np.uint8)


def illumina_fastq(n_reads: int, first: int = 1, read_len: int = 150, seed: int = SEED, crlf: bool = False, binned: bool = False) -> bytes:
    """Config 1-4 shape: Illumina-like fixed-length reads, Phred+33, '+' line bare, LF newlines."""
    nl = b"\r\n" if crlf else b"\n"
    out = []
    step = 65536
    for lo in range(first, first + n_reads, step):
        cnt = min(step, first + n_reads - lo)
        b, q = illumina_bases_quals(lo, cnt, read_len, seed, binned)
        for k in range(cnt):
            out.append(illumina_title(lo + k)); out.append(nl)
            out.append(b[k].tobytes()); out.append(nl + b"+" + nl)
            out.append(q[k].tobytes()); out.append(nl)
    return b"".join(out)


_IUPAC = np.frombuffer(b"NRYKMSWBDHV", dtype=np.uint8)


def iontorrent_fastq(n_reads: int, first: int = 1, seed: int = SEED ^ 0x454) -> bytes:
    """Config 5 shape: 454/Ion-Torrent-like variable-length reads (40..500) with 1 % IUPAC codes."""
    out = []
    with np.errstate(over="ignore"):
        for i in range(first, first + n_reads):
            hl = int(_mix64(np.array([(i << 10) ^ seed ^ 0xABCDEF], dtype=np.uint64))[0])
            L = 40 + hl % 461
            p = np.arange(L, dtype=np.uint64)
            h = _mix64((np.uint64(i) << np.uint64(10) | p) ^ np.uint64(seed))
            h2 = _mix64(h)
            base = np.frombuffer(b"ACGT", dtype=np.uint8)[(h & np.uint64(3)).astype(np.int64)]
            amb = ((h >> np.uint64(2)) % np.uint64(100)) == 0
            code = _IUPAC[((h >> np.uint64(16)) % np.uint64(11)).astype(np.int64)]
            s = _byte_sum(h2)
            z4 = np.floor_divide((s - 1020) * 4 + 104, 209)
            q = np.clip(28 + 2 * z4, 0, 40)
            zero = amb & (((h >> np.uint64(24)) % np.uint64(10)) < 7)
            q = np.where(zero, 0, q)
            base = np.where(amb, code, base)
            a = (i * 7) % 10000
            bb = (i * 13) % 10000
            out.append(b"@GXYZ1234.%d length=%d xy=%04d_%04d region=%d\n" % (i, L, a, bb, 1 + i % 4))
            out.append(base.astype(np.uint8).tobytes()); out.append(b"\n+\n")
            out.append((q + 33).astype(np.uint8).tobytes()); out.append(b"\n")
    return b"".join(out)
