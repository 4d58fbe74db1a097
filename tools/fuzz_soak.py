#!/usr/bin/env python3
"""One-off soak: many more fuzz seeds than the test-suite runs, GPU path vs oracle, for a bounded time.
Usage: tools/fuzz_soak.py [first_seed] [seconds] [batch|solid|records|decode]"""
import dataclasses
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsrc_amd import _lib  # noqa: E402
from tests._oracle import Config, Oracle  # noqa: E402
from tests.cases import fuzz_fastq, fuzz_solid  # noqa: E402


def batch_mode(seed, limit):
    """Batches of 40-70 heterogeneous chunks per scheduler pass: range-coder waves with chains of very different
    lengths (full and partial waves), block-to-block state carried in chunk order."""
    import ctypes as C
    import random
    from tests._oracle import _orc_cfg
    o = Oracle()
    t0 = time.time(); n = 0; nb = 0
    cfgs = [(3, 2, False, False), (2, 1, True, False), (1, 1, False, True), (0, 2, False, False), (3, 0, False, False)]
    while time.time() - t0 < limit:
        rng = random.Random(seed)
        d, q, lossy, crc = cfgs[seed % len(cfgs)]
        cfg = Config.from_levels(d, q, lossy, crc)
        chunks = []
        want_n = rng.randrange(40, 71)
        s2 = seed * 1000
        while len(chunks) < want_n:
            data, _ = fuzz_fastq(s2, rng.choice([None, None, 2000, 6000])); s2 += 1
            try:
                o.compress_block(cfg, data)      # reference-UB inputs cannot be part of a batch
            except RuntimeError:
                continue
            chunks.append(data)
        h = _lib.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc)
        got = h.compress_batch(chunks); h.close()
        cap = C.c_uint32(0); c = _orc_cfg(cfg)
        for i, ch in enumerate(chunks):
            out = (C.c_uint8 * (len(ch) + 65536))(); osz = C.c_uint64(); raw = (C.c_uint64 * 4)(); comp = (C.c_uint64 * 4)()
            assert o.lib.orc_compress_block_state(C.byref(c), C.byref(cap), ch, C.c_uint64(len(ch)), out, C.c_uint64(len(out)), C.byref(osz), raw, comp) == 0
            assert got[i][0] == bytes(out[:osz.value]), f"batch seed {seed} chunk {i} -d{d} -q{q} lossy={lossy}: GPU block differs from the oracle"
            n += 1
        nb += 1; seed += 1
    print(f"fuzz soak (batches): {nb} batches, {n} blocks identical, {time.time() - t0:.0f} s")


def solid_mode(seed, limit):
    """SOLiD colour space: single blocks and, every fourth seed, a batch of several chunks."""
    o = Oracle()
    t0 = time.time(); n = 0; refused = 0
    cfgs = [(0, 0, False, False), (3, 2, False, True), (2, 1, True, False), (1, 1, False, False), (2, 2, False, False), (3, 2, True, False), (0, 2, False, False), (3, 0, False, True)]
    while time.time() - t0 < limit:
        chunks = [fuzz_solid(seed * 10 + k, [None, 3000, 9000][(seed + k) % 3]) for k in range(4 if seed % 4 == 0 else 1)]
        for d, q, lossy, crc in cfgs:
            cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), color_space=True)
            ok = []
            for data, desc in chunks:
                try:
                    ok.append((data, desc, o.compress_block(cfg, data)))
                except RuntimeError as e:
                    assert "rc=-2" in str(e), e
                    refused += 1
            if not ok:
                continue
            h = _lib.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, color_space=True)
            got = h.compress_batch([x[0] for x in ok]); h.close()
            for (data, desc, want), g in zip(ok, got):
                assert g == want, f"solid seed {seed} {desc} -d{d} -q{q} lossy={lossy} crc={crc}: GPU block differs from the oracle"
                n += 1
        seed += 1
    print(f"fuzz soak (colour space): {n} blocks identical, {refused} reference-UB inputs skipped, seeds up to {seed - 1}, {time.time() - t0:.0f} s")


def records_mode(seed, limit):
    """Record layout (dsrcgpu_set_record_layout): batches of LF-only chunks with running chunkSize words."""
    o = Oracle()
    t0 = time.time(); n = 0
    cfgs = [(0, 0, False), (3, 0, False), (2, 1, True), (3, 2, True), (1, 0, False)]
    while time.time() - t0 < limit:
        d, q, lossy = cfgs[seed % len(cfgs)]
        cfg = Config(dna_order=3 * d, quality_order=3 * q, lossy=lossy)
        chunks = []; s2 = seed * 100
        while len(chunks) < 6:
            data = fuzz_fastq(s2, [None, 2500, 7000][s2 % 3])[0].replace(b"\r\n", b"\n"); s2 += 1
            try:
                o.compress_records_block(cfg, data, 1)
            except RuntimeError:
                continue
            chunks.append(data)
        sizes = []; tot = 0xFFFFF000 if seed % 2 else 0
        for c in chunks:
            tot += len(c) + 1; sizes.append(tot & 0xFFFFFFFF)
        h = _lib.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, False)
        h.set_record_layout(sizes)
        got = h.compress_batch(chunks); h.close()
        cap = 0
        for c, sz, g in zip(chunks, sizes, got):
            want, cap = o.compress_records_block(cfg, c, sz, cap)
            assert g[0] == want, f"records seed {seed} -d{d} -q{q} lossy={lossy}: GPU block differs from the oracle"
            n += 1
        seed += 1
    print(f"fuzz soak (record layout): {n} blocks identical, seeds up to {seed - 1}, {time.time() - t0:.0f} s")


def decode_mode(seed, limit):
    """Round trip on the GPU: batches of 20-40 heterogeneous chunks compressed (state carried in chunk order), the blocks
    decompressed by the GPU decoder and compared with the oracle's decoder (pinned to the reference's Read) on the same
    blocks and with the stored checksums.  (The fuzz inputs carry CR LF, repeated titles on the + line ...: the reference does
    not give those back byte for byte, so the input itself is not the yardstick here; tests/test_gpu_decode.py has the round trips.)"""
    import random
    o = Oracle()
    t0 = time.time(); n = 0; nb = 0; broken = 0
    cfgs = [(3, 2, False, True), (0, 0, False, False), (2, 1, True, True), (1, 2, False, False), (3, 0, False, True), (0, 2, True, False)]
    while time.time() - t0 < limit:
        rng = random.Random(seed)
        d, q, lossy, crc = cfgs[seed % len(cfgs)]
        flags = [0, 0, 0b110, 0][seed % 4]
        cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), tag_flags=flags)
        chunks = []; s2 = seed * 1000
        want_n = rng.randrange(20, 41)
        while len(chunks) < want_n:
            data, _ = fuzz_fastq(s2, rng.choice([None, None, 2000, 6000])); s2 += 1
            try:
                o.compress_block(cfg, data)
            except RuntimeError:
                continue
            chunks.append(data)
        h = _lib.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, tag_flags=flags)
        blocks = [b[0] for b in h.compress_batch(chunks)]
        keep = []; want = []
        for blk, ch in zip(blocks, chunks):
            try:
                want.append(o.decompress_block(cfg, blk, len(ch) + 4096)); keep.append((blk, ch))
            except RuntimeError:
                broken += 1                                      # the reference's own decoder runs off this block
        texts, ok = h.decompress_batch([b for b, _ in keep], text_caps=[len(c) + 4096 for _, c in keep], verify=True)
        h.close()
        for i, (blk, ch) in enumerate(keep):
            assert texts[i] == want[i], f"decode seed {seed} chunk {i} -d{d} -q{q} lossy={lossy} flags={flags:#x}: GPU text differs from the oracle's"
            if crc: assert ok[i] == o.verify_block(cfg, blk, len(ch) + 4096), f"decode seed {seed} chunk {i}: checksum verdict {ok[i]} differs from the reference's VerifyChecksum"
            n += 1
        nb += 1; seed += 1
    print(f"fuzz soak (decode round trips): {nb} batches, {n} blocks decoded identically, {broken} blocks the reference cannot decode skipped, {time.time() - t0:.0f} s")


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "decode":
        return decode_mode(int(sys.argv[1]), float(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "batch":
        return batch_mode(int(sys.argv[1]), float(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "solid":
        return solid_mode(int(sys.argv[1]), float(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "records":
        return records_mode(int(sys.argv[1]), float(sys.argv[2]))
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    limit = float(sys.argv[2]) if len(sys.argv) > 2 else 180.0
    o = Oracle()
    t0 = time.time(); n = 0; refused = 0
    cfgs = [(0, 0, False, False), (3, 2, False, True), (2, 1, True, False), (1, 1, False, False), (2, 2, False, False), (3, 2, True, False), (0, 2, False, False), (3, 0, False, False)]
    while time.time() - t0 < limit:
        nrec = [None, 3000, 7000][seed % 3]
        data, desc = fuzz_fastq(seed, nrec)
        for ci, (d, q, lossy, crc) in enumerate(cfgs):
            flags = [0, 0, 0b110, 0x7FFFFFFE, 0b101000, 0, 0b10, 0][(seed + ci) % 8]      # -f masks on some
            cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), tag_flags=flags)
            h = _lib.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, tag_flags=flags)
            try:
                want = o.compress_block(cfg, data)
            except RuntimeError as e:
                assert "rc=-2" in str(e), e
                try:
                    h.compress_block(data)
                    raise AssertionError(f"seed {seed} {desc} -d{d} -q{q}: reference-UB input was not refused")
                except _lib.DsrcGpuError:
                    refused += 1
                h.close()
                continue
            got = h.compress_block(data)
            h.close()
            assert got == want, f"seed {seed} {desc} -d{d} -q{q} lossy={lossy} crc={crc}: GPU block differs from the oracle"
            n += 1
        seed += 1
    print(f"fuzz soak: {n} blocks identical, {refused} reference-UB inputs refused, seeds up to {seed - 1}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
