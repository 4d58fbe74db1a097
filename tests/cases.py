"""Shared parity cases: (name, d, q, lossy, crc, generator) used by the emulator tests (CPU) and the GPU tests."""
from __future__ import annotations

import random

from dsrc_amd import synth

TINY = (b"@SEQ.1 lane:1:10:100\nACGTACGTAC\n+\nIIIIHHGG##\n@SEQ.2 lane:1:12:205\nTTGCANNGTA\n+\nIIFF##!!CC\n"
        b"@SEQ.3 lane:1:15:317\nGGGGCCCCAA\n+\nABCDEFGHII\n@SEQ.4 lane:2:11:90\nACACACACGT\n+\nIIIIIIII##")

LEVELS = [(0, 0, False, False), (3, 2, False, False), (2, 1, True, False), (0, 0, False, True), (1, 1, False, False),
          (0, 0, True, False), (3, 2, True, False), (2, 2, False, True), (0, 2, False, False), (3, 0, False, False)]


def fuzz_fastq(seed: int, nrec: int | None = None):
    """Diverse small FASTQ chunks: every quality/DNA/tag scheme of the reference is reachable.
    Returns (chunk_bytes_without_final_newline, description)."""
    rng = random.Random(seed)
    if nrec is None:
        nrec = rng.choice([2, 3, 10, 300, 1200])
    style = rng.choice(['illumina', 'casava', 'sra', 'weird', 'mixed', 'strlen'])
    varlen = rng.random() < 0.4
    L0 = rng.choice([1, 5, 36, 76, 100, 151, 250])
    qmode = rng.choice(['wide', 'binned', 'tails', 'runs', 'const', 'few'])
    nmode = rng.choice(['none', 'lowq', 'highq', 'iupac'])
    crlf = rng.random() < 0.15
    nl = b'\r\n' if crlf else b'\n'
    out = []
    x = rng.randrange(1000); lane = 1
    for i in range(nrec):
        L = rng.randrange(max(1, L0 // 2), L0 + 1) if varlen else L0
        if style == 'illumina':
            if rng.random() < 0.3: x = rng.randrange(20000)
            if rng.random() < 0.01: lane += 1
            t = b"@HWI-ST%d:%d:FC:%d:%d:%d:%d 1:N:0:%s" % (700, 33, lane, 1101 + i // 50, x, (i * 37) % 5000, rng.choice([b'ACGT', b'ACGT', b'TTAG']))
        elif style == 'casava':
            t = b"@M00123_%d/%d" % (i * 3 + 7, 1 + (i & 1))
        elif style == 'sra':
            t = b"@SRR%d.%d %d length=%d" % (1234, i + 1, i + 1, L)
        elif style == 'weird':
            t = b"@r%04d#%s=%d,%d" % (i, rng.choice([b'a', b'bb', b'ccc']), rng.randrange(3), 1000000 + rng.randrange(600))
        elif style == 'strlen':
            t = b"@id_%s %d %s" % (bytes(rng.choice(b'abcdefgh') for _ in range(rng.randrange(1, 9))), rng.randrange(40, 60), rng.choice([b'x:y', b'x:yy', b'xq:z']))
        else:
            t = rng.choice([b"@a.%d b" % i, b"@a.%d" % i, b"@a_%d b:c" % i]) if i > 3 else b"@a.%d b" % i
        alpha = b'ACGTNRWS' if nmode == 'iupac' else b'ACGT'
        seq = bytearray(rng.choice(alpha) for _ in range(L))
        if qmode == 'wide': q = [rng.randrange(2, 41) for _ in range(L)]
        elif qmode == 'binned': q = [rng.choice([2, 11, 25, 37]) for _ in range(L)]
        elif qmode == 'few': q = [rng.choice([30, 31]) for _ in range(L)]
        elif qmode == 'const': q = [40] * L
        elif qmode == 'runs':
            q = []
            while len(q) < L: q += [rng.choice([2, 15, 30, 40])] * rng.randrange(1, 400)
            q = q[:L]
        else:
            k = rng.randrange(0, L + 1) if rng.random() < 0.7 else L
            q = [rng.randrange(20, 41) for _ in range(k)] + [2] * (L - k)
        if nmode in ('lowq', 'highq', 'iupac'):
            for j in range(L):
                if rng.random() < 0.03:
                    if nmode != 'iupac': seq[j] = ord('N')
                    q[j] = rng.randrange(0, 7) if nmode == 'lowq' else rng.randrange(0, 41)
                elif nmode == 'iupac' and seq[j] in b'NRWS' and rng.random() < 0.5:
                    q[j] = rng.randrange(0, 7)
        out += [t, nl, bytes(seq), nl, b'+', nl, bytes(v + 33 for v in q), nl]
    data = b''.join(out)
    return data[:-len(nl)], (style, varlen, L0, qmode, nmode, crlf)


def fuzz_solid(seed: int, nrec: int | None = None):
    """SOLiD colour-space chunks (primer base + colours 0-3 and '.', as many qualities as sequence characters).
    Returns (chunk_bytes_without_final_newline, description)."""
    rng = random.Random(0x501D ^ seed)
    if nrec is None:
        nrec = rng.choice([2, 5, 40, 400, 1500])
    const_begin = rng.random() < 0.7
    varlen = rng.random() < 0.3
    L0 = rng.choice([2, 3, 26, 36, 51, 76])
    dots = rng.choice([0.0, 0.0, 0.01, 0.1])
    qmode = rng.choice(['wide', 'binned', 'tails', 'runs', 'few'])
    q0 = rng.choice([0, 0, 2, 30, None])                # quality of the primer position (None: like the others)
    out = []
    for i in range(nrec):
        L = rng.randrange(max(2, L0 // 2), L0 + 1) if varlen else L0
        first = b'T' if const_begin else bytes([rng.choice(b'TTTGAC')])
        seq = bytearray(first) + bytearray(rng.choice(b'0123') for _ in range(L - 1))
        if qmode == 'wide': q = [rng.randrange(2, 41) for _ in range(L)]
        elif qmode == 'binned': q = [rng.choice([2, 11, 25, 37]) for _ in range(L)]
        elif qmode == 'few': q = [rng.choice([30, 31]) for _ in range(L)]
        elif qmode == 'runs':
            q = []
            while len(q) < L: q += [rng.choice([2, 15, 30, 40])] * rng.randrange(1, 400)
            q = q[:L]
        else:
            k = rng.randrange(0, L + 1) if rng.random() < 0.7 else L
            q = [rng.randrange(20, 41) for _ in range(k)] + [2] * (L - k)
        for j in range(1, L):
            if rng.random() < dots:
                seq[j] = ord('.')
                q[j] = rng.choice([0, 1, 2, 5, 8, 20])
        if q0 is not None:
            q[0] = q0
        t = b"@SRR%d.%d solid_%d_%d_%d" % (2000 + seed, i + 1, 1 + i // 300, (i * 7) % 2048, (i * 13) % 2048)
        out += [t, b'\n', bytes(seq), b'\n+\n', bytes(v + 33 for v in q), b'\n']
    return b''.join(out)[:-1], (const_begin, varlen, L0, dots, qmode, q0)


def rle_chunks():
    """Quality strings made of long runs: the RLE scheme (QualityRLEModeler) with 4, 20 and 45 distinct values -- the small
    alphabets keep code tables and histograms in LDS, the large ones take the global-memory fallbacks -- and runs longer
    than 255 (split) that cross record boundaries."""
    import random
    out = []
    for nsym, nrec, L in ((4, 6000, 150), (20, 3000, 200), (45, 1500, 250)):
        rng = random.Random(nsym)
        vals = rng.sample(range(2, 60), nsym)
        recs = []; cur = vals[0]; left = 0
        for i in range(nrec):
            q = bytearray()
            for _ in range(L):
                if left == 0:
                    cur = rng.choice(vals); left = rng.choice([1, 2, 3, 7, 30, 90, 400, 700])
                q.append(33 + cur); left -= 1
            seq = bytes(rng.choice(b"ACGT") for _ in range(L))
            recs.append(b"@r.%d x\n" % i + seq + b"\n+\n" + bytes(q))
        out.append(b"\n".join(recs))
    return out


def state_dependent_fastq(n_per_region: int = 9000):
    """A file whose archive depends on the state DSRC carries from block to block (the capacity of TagStats::fields,
    DESIGN.md section 1): regions of ~1 MB whose titles have 5, 9, 17, 9, 3 and 17 fields, all numeric fields coded
    ValueVar + Huffman.  With -b1 every region becomes one or two blocks."""
    rng = random.Random(2024)
    out = []
    first = 1
    for nf in (5, 9, 17, 9, 3, 17):
        for i in range(n_per_region):
            title = b"@r.%d" % (first + i) + b"".join(b":%d" % ((7 * i + k) % 90 + 10) for k in range(nf - 2))
            seq = bytes(rng.choice(b"ACGT") for _ in range(36))
            qua = bytes(33 + rng.randint(20, 40) for _ in range(36))
            out += [title, b"\n", seq, b"\n+\n", qua, b"\n"]
        first += n_per_region
    return b"".join(out)


def alphabet_fastq(n_sym: int, n_rec: int = 400, L: int = 100, seed: int = 1, iupac: bool = False, spread: bool = False, q_max: int = 93):
    """Reads whose qualities use exactly `n_sym` distinct values (the order models take the 16-, 32-, 64- or 128-symbol alphabet
    that holds them).  spread = False: values drawn around a slowly moving mean (few contexts, as instruments write them);
    True: independent uniform values (as many contexts as symbols allow: the bucketed path runs out of counter rows).
    iupac: N/R/W/S with high quality stay in the DNA stream (the 8-symbol DNA alphabet at -d >= 1)."""
    rng = random.Random(1000 * n_sym + seed)
    vals = sorted(rng.sample(range(0, q_max), n_sym))      # q_max <= 64 for the lossy modes (their bin table has 64 entries)
    recs = []
    for i in range(n_rec):
        if spread:
            q = [rng.choice(vals) for _ in range(L)]
        else:
            c = rng.randrange(n_sym)
            q = []
            for _ in range(L):
                c = min(n_sym - 1, max(0, c + rng.choice([-1, 0, 0, 0, 1])))
                q.append(vals[c])
        if i == 0:
            q[:n_sym] = vals[:L]                        # every value at least once
        alpha = b"ACGTACGTACGTNRWS" if iupac else b"ACGT"
        seq = bytes(rng.choice(alpha) for _ in range(L))
        if iupac:
            q = [max(v, 10) if seq[k] in b"NRWS" else v for k, v in enumerate(q)]
        recs.append(b"@a.%d %d\n" % (i, L) + seq + b"\n+\n" + bytes(33 + v for v in q))
    return b"\n".join(recs)
