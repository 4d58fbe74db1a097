"""GPU parity: blocks from the HIP path (through the C ABI) must be bit-identical to the oracle's.

Bit-exact bar: integer/byte work only, no tolerance.  Sizes are what the oracle finishes in seconds;
full-size properties are in test_gpu_properties.py."""
import os

import pytest

from dsrc_amd import synth
from tests.cases import LEVELS, TINY, fuzz_fastq
from tests._oracle import Config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not os.environ.get("DSRC_TEST_KEEP_GPU_LIB"):          # (set to run this suite on a variant build: tools/variant_bench.sh)
        os.environ.pop("DSRC_GPU_LIB", None)
    from dsrc_amd import _lib
    _lib._lib = None
    return _lib


def _check(gpu, oracle, cfg, chunks):
    h = gpu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset)
    got = h.compress_batch(chunks)
    h.close()
    want = oracle.compress_blocks_state(cfg, chunks)          # EVERY chunk, compressor state carried like `dsrc c -t1`
    assert len(got) == len(want)
    for i in range(len(chunks)):
        assert got[i][0] == want[i][0], f"chunk {i}: block bytes differ"
        assert got[i][1] == want[i][1] and got[i][2] == want[i][2], f"chunk {i}: stream sizes differ"
    return got


@pytest.mark.parametrize("d,q,lossy,crc", LEVELS)
def test_tiny(gpu, oracle, d, q, lossy, crc):
    _check(gpu, oracle, Config.from_levels(d, q, lossy, crc), [TINY])


@pytest.mark.parametrize("d,q,lossy,crc", LEVELS)
def test_illumina_20k(gpu, oracle, d, q, lossy, crc):
    data = synth.illumina_fastq(20000)[:-1]
    _check(gpu, oracle, Config.from_levels(d, q, lossy, crc), [data])


@pytest.mark.parametrize("d,q,lossy,crc", [(2, 1, True, False), (0, 0, False, False), (0, 2, False, False), (0, 1, False, True)])
def test_iontorrent_5k(gpu, oracle, d, q, lossy, crc):
    data = synth.iontorrent_fastq(5000)[:-1]
    _check(gpu, oracle, Config.from_levels(d, q, lossy, crc), [data])


def test_batch_of_blocks_with_state(gpu, oracle):
    """Several chunks in one scheduler pass; compressor state advances in chunk order like `dsrc c -t1`."""
    import ctypes as C
    from tests._oracle import _orc_cfg
    chunks = [synth.illumina_fastq(3000, first=1 + 3000 * k)[:-1] for k in range(5)] + [synth.iontorrent_fastq(800)[:-1]]
    for d, q in ((0, 2), (0, 0), (0, 1)):          # lossless order-k DNA on IUPAC data is undefined in the reference
        cfg = Config.from_levels(d, q)
        h = gpu.Handle(cfg.dna_order, cfg.quality_order)
        got = h.compress_batch(chunks)
        h.close()
        cap = C.c_uint32(0)
        c = _orc_cfg(cfg)
        for i, ch in enumerate(chunks):
            out = (C.c_uint8 * (len(ch) + 65536))(); osz = C.c_uint64(); raw = (C.c_uint64 * 4)(); comp = (C.c_uint64 * 4)()
            rc = oracle.lib.orc_compress_block_state(C.byref(c), C.byref(cap), ch, C.c_uint64(len(ch)), out, C.c_uint64(len(out)), C.byref(osz), raw, comp)
            assert rc == 0
            assert got[i][0] == bytes(out[:osz.value]), f"chunk {i} differs at -d{d} -q{q}"


@pytest.mark.parametrize("seed", range(60))
def test_fuzz(gpu, oracle, seed):
    data, desc = fuzz_fastq(seed)
    for d, q, lossy, crc in [(0, 0, False, False), (3, 2, False, True), (2, 1, True, False), (1, 1, False, False), (0, 0, True, False)]:
        cfg = Config.from_levels(d, q, lossy, crc)
        try:
            want = oracle.compress_block(cfg, data)
        except RuntimeError as e:
            if "rc=-2" in str(e):      # input is undefined behaviour in the reference: the GPU path must refuse it too
                h = gpu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc)
                with pytest.raises(gpu.DsrcGpuError):
                    h.compress_block(data)
                h.close()
                continue
            raise
        h = gpu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc)
        got = h.compress_block(data)
        h.close()
        assert got == want, f"seed {seed} {desc} -d{d} -q{q} lossy={lossy} crc={crc}"


def test_hot_contexts_rescale(gpu, oracle):
    """Contexts with > 32k symbols: the adaptive rows rescale several times (SymbolCoderRC::Rescale)."""
    import random
    rng = random.Random(5)
    recs = []
    for i in range(4000):
        seq = ''.join(rng.choice('AAAAAAAC') for _ in range(250))
        q = ''.join('I' if rng.random() < 0.97 else 'H' for _ in range(250))
        recs.append(f"@r.{i}\n{seq}\n+\n{q}")
    data = '\n'.join(recs).encode()
    for d, q, lossy in [(1, 2, False), (3, 1, False), (2, 2, True), (3, 2, False)]:
        _check(gpu, oracle, Config.from_levels(d, q, lossy), [data])


def test_rle_quality_alphabets(gpu, oracle):
    """RLE quality scheme at -q0 with small and large alphabets (LDS tables / global fallbacks in k_qrle_hist, k_qrle_emit)."""
    from tests.cases import rle_chunks
    chunks = rle_chunks()
    for c in chunks:
        blk, _, comp = oracle.compress_block(Config.from_levels(0, 0), c)
        assert blk[comp[0] + comp[1]] == 2          # scheme byte of the quality stream: RLE
    for d, q, lossy in [(0, 0, False), (2, 0, False)]:
        _check(gpu, oracle, Config.from_levels(d, q, lossy), chunks)


def test_range_coder_reference_loop_path(gpu_hooks, oracle, monkeypatch):
    """The carry-clamp fallback of k_rc (reference loop + byte re-dealing) must give the same stream as the fast
    path; DSRC_GPU_FORCE_EXACT_RC sends every 16-symbol group through it."""
    monkeypatch.setenv("DSRC_GPU_FORCE_EXACT_RC", "1")
    chunks = [synth.illumina_fastq(20000, first=1 + 20000 * k)[:-1] for k in range(3)]
    for d, q, lossy in [(3, 2, False), (2, 1, True)]:
        _check(gpu_hooks, oracle, Config.from_levels(d, q, lossy), chunks)


@pytest.mark.parametrize("hook", ["DSRC_GPU_RC_REDO", "DSRC_GPU_RC_RECOVER", "DSRC_GPU_RC_ONE_LANE"])
def test_range_coder_redo_list_and_one_lane_kernel(gpu_hooks, oracle, monkeypatch, hook):
    """Round 6: k_rcs codes a stream on two waves (range, low); a stream in which the carry clamp fires is put on a redo list and coded
    again by k_rc (both recurrences in one lane) after the batch's state read-back -- but first the lane tries to put things right inside
    the kernel (rcs_recover: the chunk walked again with the reference's loop, the next chunk's words and the range handed to wave R).
    DSRC_GPU_RC_REDO makes every stream report a clamp and go to the list (whatever k_rcs wrote for it is overwritten),
    DSRC_GPU_RC_RECOVER forces a recovery every few chunks of every stream (lanes that meet in one period: the list),
    DSRC_GPU_RC_ONE_LANE sends the whole batch through k_rc: same blocks."""
    monkeypatch.setenv(hook, "1")
    chunks = [synth.illumina_fastq(20000, first=1 + 20000 * k)[:-1] for k in range(3)] + [synth.illumina_fastq(9000, first=777777)[:-1], synth.illumina_fastq(40)[:-1]]
    for d, q, lossy in [(3, 2, False), (2, 1, True), (1, 1, False)]:
        _check(gpu_hooks, oracle, Config.from_levels(d, q, lossy), chunks)


def test_staging_overflow_second_pass(gpu_hooks, oracle, monkeypatch):
    """tests/test_emu_kernels.py::test_staging_overflow_second_pass on the GPU: the range coder's staging estimate made too small, the
    batch run again with worst-case staging -- device-resident form with a field filter (the caller's text stays as it was) and with a
    fixed arena."""
    import dataclasses
    monkeypatch.setenv("DSRC_GPU_HOOK_RC_BOUND_SHIFT", "3")
    chunks = [synth.illumina_fastq(9000, first=1 + 9000 * k)[:-1] for k in range(3)]
    for flags, fixed in ((0b1010, 0), (0, 1 << 30), (0b10, 1 << 30)):
        cfg = dataclasses.replace(Config.from_levels(3, 2), tag_flags=flags)
        want = [oracle.compress_block(cfg, c)[0] for c in chunks]
        h = gpu_hooks.Handle(cfg.dna_order, cfg.quality_order, tag_flags=flags, arena_bytes=fixed)
        blob = b"\n".join(chunks)
        offs = []; at = 0
        for c in chunks:
            offs.append(at); at += len(c) + 1
        d_in = h.dev_alloc(len(blob) + 64); h.dev_upload(d_in, blob)
        cap = 16 << 20; d_out = h.dev_alloc(cap)
        o_offs, o_sizes, _, _ = h.compress_batch_device(d_in, offs, [len(c) for c in chunks], d_out, cap)
        out = h.dev_download(d_out, o_offs[-1] + o_sizes[-1])
        assert [out[o_offs[i]: o_offs[i] + o_sizes[i]] for i in range(3)] == want, (flags, fixed)
        assert h.dev_download(d_in, len(blob)) == blob
        h.dev_free(d_in); h.dev_free(d_out); h.close()


def test_split_range_coder_selftest(gpu):
    """dsrcgpu_selftest: the two-wave coder against the reference's loop on states at the carry clamp (k_selftest_rcs), the exact
    divisions, the LDS ordering the front end stands on."""
    h = gpu.Handle()
    assert h.selftest() == 0
    h.close()


def test_concurrent_scheduler_instances(gpu, oracle):
    """Several handles driven from several host threads at once (what bench.py does) give the same blocks as one
    handle alone: no state is shared between instances."""
    import threading
    cfg = Config.from_levels(3, 2)
    batches = [[synth.illumina_fastq(6000, first=1 + 100000 * t + 6000 * k)[:-1] for k in range(3)] for t in range(3)]
    want = []
    for b in batches:
        h = gpu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset)
        want.append(h.compress_batch(b)); h.close()
    assert want[0][0][0] == oracle.compress_block(cfg, batches[0][0])[0]
    got = [None] * 3; errs = []

    def work(t):
        try:
            for _ in range(3):          # a fresh handle each time: block-to-block state starts like `want`'s
                h = gpu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset)
                got[t] = h.compress_batch(batches[t])
                h.close()
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(3)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for t in range(3):
        assert [g[0] for g in got[t]] == [w[0] for w in want[t]]


def test_queue_form(gpu, oracle):
    """dsrcgpu_submit / flush / collect -- the form INTEGRATION.md section 1 binds in place of DsrcCompressor::Process
    (reference src/DsrcWorker.cpp:39-70) -- on the real device: part ids come back with their blocks, blocks equal the
    oracle's with the state carried in submission order, over several flushes of one handle."""
    chunks = [synth.illumina_fastq(2500, first=1 + 2500 * k)[:-1] for k in range(6)] + [synth.iontorrent_fastq(600)[:-1]]
    for d, q, lossy in ((0, 0, False), (0, 2, False), (2, 1, True)):
        cfg = Config.from_levels(d, q, lossy)
        want = oracle.compress_blocks_state(cfg, chunks)
        h = gpu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset)
        got = []
        for lo, hi in ((0, 3), (3, 4), (4, 7)):
            for i in range(lo, hi):
                h.submit(1000 + i, chunks[i])
            h.flush()
            while True:
                r = h.collect()
                if r is None:
                    break
                got.append(r)
        assert h.collect() is None
        h.close()
        assert [g[0] for g in got] == [1000 + i for i in range(len(chunks))]
        for i, g in enumerate(got):
            assert (g[1], g[2], g[3]) == want[i], f"part {i} at -d{d} -q{q}"


def test_full_size_iontorrent_lossy(gpu, oracle):
    """BASELINE config 5 at block size: variable-length reads with IUPAC codes, -d2 -q1 -l, twelve 8 MiB chunks cut like
    the reference's reader cuts them (FLAG_VARIABLE_LENGTH, per-record length bits in the tag stream, the qp_stream
    position contexts of k_sort).  Every block against the oracle, sampled blocks through the (reference-pinned) oracle decoder."""
    data = synth.iontorrent_fastq(178000)
    cuts = oracle.cut_chunks(data, 8 << 20)
    assert len(cuts) >= 12
    chunks = [data[s: s + n] for s, n in cuts[:12]]
    cfg = Config.from_levels(2, 1, True)
    got = _check(gpu, oracle, cfg, chunks)
    for i in (0, 11):
        blk = got[i][0]
        assert blk[8:12] == b"\x00\x00\x00\x02"                      # FLAG_VARIABLE_LENGTH
        text = oracle.decompress_block(cfg, blk, len(chunks[i]) + 64)      # pinned against the reference's Read
        lines = text.split(b"\n"); src = chunks[i].split(b"\n")
        assert lines[0::4][:-1] == src[0::4] and [len(x) for x in lines[1::4]] == [len(x) for x in src[1::4]]


def test_sort_ballot_variant(gpu_hooks, oracle, monkeypatch):
    """k_sort ranks with LDS atomics on a device that passed k_lds_order_test (the self-test below fails if MI355X ever does
    not) and with ballots otherwise; DSRC_GPU_SORT_BALLOT=1 forces the second variant: identical blocks, both equal to the oracle."""
    chunks = [synth.illumina_fastq(3000, first=1 + 3000 * i)[:-1] for i in range(3)] + [fuzz_fastq(77, 3000)[0]]
    for d, q, lossy in [(3, 2, False), (2, 1, True)]:
        cfg = Config.from_levels(d, q, lossy)
        want = [oracle.compress_block(cfg, c) for c in chunks]
        h = gpu_hooks.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset)
        atomic = [h.compress_block(c) for c in chunks]
        h.close()
        monkeypatch.setenv("DSRC_GPU_SORT_BALLOT", "1")
        h = gpu_hooks.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset)
        ballot = [h.compress_block(c) for c in chunks]
        h.close()
        monkeypatch.delenv("DSRC_GPU_SORT_BALLOT")
        assert atomic == want and ballot == want


def test_no_kernel_reads_unwritten_arena_bytes(gpu_hooks, oracle, monkeypatch):
    """DSRC_GPU_DEBUG_FILL=<byte> fills the arena before every batch: the blocks must not depend on the byte (nothing reads
    what nobody wrote) and equal the oracle's."""
    chunks = [synth.illumina_fastq(4000, first=1 + 4000 * i)[:-1] for i in range(3)] + [fuzz_fastq(91, 3000)[0]]
    for d, q, lossy, crc in [(3, 2, False, True), (0, 0, False, False), (2, 1, True, False)]:
        cfg = Config.from_levels(d, q, lossy, crc)
        want0 = oracle.compress_block(cfg, chunks[0])
        ref = None
        for fill in ("0", "255", "90"):
            monkeypatch.setenv("DSRC_GPU_DEBUG_FILL", fill)
            h = gpu_hooks.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset)
            got = h.compress_batch(chunks) + h.compress_batch(chunks[::-1])
            h.close()
            assert got[0] == want0, (d, q, fill)
            assert ref is None or got == ref, (d, q, fill)
            ref = got
        monkeypatch.delenv("DSRC_GPU_DEBUG_FILL")


def test_exact_division_selftest(gpu):
    h = gpu.Handle()
    assert h.selftest() == 0
    h.close()


@pytest.mark.parametrize("seed", range(400, 440))
def test_field_filter(gpu, oracle, seed):
    """-f on the GPU path (k_tag_filter / k_tag_poke) against the oracle's restatement of FastqParserExt."""
    import dataclasses
    data, desc = fuzz_fastq(seed, [None, 300, 3000][seed % 3])
    for flags in (0b10, 0b1010, 0b11110, 0x7FFFFFFE, 0b100100):
        for d, q, lossy, crc in [(0, 0, False, True), (3, 2, False, False), (2, 1, True, True)]:
            cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), tag_flags=flags)
            h = gpu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset, tag_flags=flags)
            try:
                want = oracle.compress_block(cfg, data)
            except RuntimeError as e:
                assert "rc=-2" in str(e)
                with pytest.raises(gpu.DsrcGpuError):
                    h.compress_block(data)
                h.close()
                continue
            got = h.compress_block(data)
            h.close()
            assert got == want, (seed, desc, bin(flags), d, q, lossy, crc)


def test_device_synth_matches_host(gpu):
    h = gpu.Handle()
    cap = 2 << 20
    d = h.dev_alloc(cap)
    n = h.synth_illumina(1, 3000, d, cap)
    got = h.dev_download(d, n)
    h.dev_free(d); h.close()
    assert got == synth.illumina_fastq(3000)


@pytest.mark.parametrize("d,q,lossy", [(0, 0, False), (3, 2, False), (0, 2, False), (2, 1, True), (0, 0, True)])
def test_read_lengths_at_wave_boundaries(gpu, oracle, d, q, lossy):
    """Reads of 1, 63, 64, 65, 127, 128, 129, 255, 256, 257 and 65 535 (2 000 at -q0) bases with hoisted (quality < 7) and kept ambiguity codes in
    the first, last and 64th lane of a 64-base step: the per-base transform and the stream compaction of k_prep_stats / k_prep_write
    (LosslessRecordsProcessor::ProcessForward, src/RecordsProcessor.cpp:209-267; lossy :344-408) where a wave's `in the read`
    mask changes.  (Round 2 saw a GPU-only wrong quality stream from k_prep_write with the transform called under that mask,
    NOTES/rounds_1_to_4.md section 10; bit-exact blocks over these lengths pin the streams both kernels write.)"""
    import random
    rng = random.Random(11 * d + q)
    amb = b"N" if (d > 0 and not lossy) else b"NRYKMSW"          # lossless order-k DNA takes at most 8 symbols
    recs = []
    for rep in range(3):
        for n in (1, 63, 64, 65, 127, 128, 129, 255, 256, 257, (65535 if q else 2000) if rep == 0 else 300):          # -q0: one Huffman tree per position
            seq = bytearray(rng.choice(b"ACGT") for _ in range(n))
            qua = bytearray(33 + rng.randint(8, 40) for _ in range(n))
            for pos in (0, 62, 63, 64, 65, 126, 127, 128, n - 2, n - 1):
                if 0 <= pos < n and rng.random() < 0.8:
                    seq[pos] = rng.choice(amb)
                    qua[pos] = 33 + (rng.randint(0, 6) if rng.random() < 0.6 else rng.randint(7, 40))          # hoisted / kept
            recs.append(b"@b.%d.%d len=%d\n%s\n+\n%s" % (rep, n, n, bytes(seq), bytes(qua)))
    data = b"\n".join(recs)
    _check(gpu, oracle, Config.from_levels(d, q, lossy), [data, data[: data.index(b"\n@b.1.")]])


@pytest.mark.timeout(1500)
def test_one_block_of_a_gigabyte(gpu, oracle):
    """`-b1024` at the order levels (the reference accepts buffer sizes of 1 .. 1024 MB, src/main.cpp:300-305): one chunk of 1 GiB --
    ~430 M bases and as many qualities per stream, 3.4 GB of 8-byte records per range-coder chain (the 32-bit byte offsets of k_rc
    reach 4 GiB = 536 M symbols) -- against the oracle, by digest."""
    import hashlib
    n = (1 << 30) - (1 << 20)
    h = gpu.Handle()
    recs = 2_900_000
    d = h.dev_alloc(recs * 400)
    nbytes = h.synth_illumina(1, recs, d, recs * 400)
    assert nbytes > n
    text = h.dev_download(d, n + 4096)
    h.dev_free(d); h.close()
    end = text.rindex(b"\n@SRRSYN.", 0, n)                 # a record boundary below 1 GiB
    chunk = text[:end]
    del text
    assert len(chunk) > (1 << 30) - (2 << 20)
    cfg = Config.from_levels(3, 2)
    h = gpu.Handle(cfg.dna_order, cfg.quality_order)
    got = h.compress_batch([chunk])[0]
    h.close()
    want = oracle.compress_block(cfg, chunk)
    assert len(got[0]) == len(want[0]) and hashlib.sha256(got[0]).digest() == hashlib.sha256(want[0]).digest()
    assert got[1] == want[1] and got[2] == want[2]


# ---- the bucketed context path (k_bucket.h: k_part, k_binoff, k_model, k_place) ----------------------------------------------

@pytest.mark.parametrize("n_sym", [3, 12, 20, 40, 90])
def test_bucketed_path_alphabet_sizes(gpu, oracle, n_sym):
    """16-, 32-, 64- and 128-symbol quality models (k_model<16..128>: one to four counter words per symbol) at -q1 / -q2, with
    the 4-symbol DNA model beside them, on blocks of ~1.3 M symbols (every bucket of the 1024 in use)."""
    from tests.cases import alphabet_fastq
    data = alphabet_fastq(n_sym, n_rec=9000, L=150)
    for d, q in ((2, 2), (1, 1)):
        _check(gpu, oracle, Config.from_levels(d, q), [data])


def test_bucketed_path_hand_backs_and_switches(gpu_hooks, oracle, monkeypatch, capfd):
    """Streams the bucketed path hands back to k_sort / k_replay inside a batch of streams it keeps: independent uniform qualities
    (k_model runs out of counter rows), a context with most of a 4 M-symbol stream (k_model finds a bucket too large for one wave);
    then the same batch with the path off and with k_model scattering to stream order itself."""
    import random
    from tests.cases import alphabet_fastq
    rng = random.Random(3)
    hot = "\n".join("@r.%d\n%s\n+\n%s" % (i, "".join(rng.choice("AAAAAAAAAAAAAAAC") for _ in range(200)),
                                            "".join("I" if rng.random() < 0.98 else "H" for _ in range(200))) for i in range(20000)).encode()
    chunks = [synth.illumina_fastq(6000)[:-1], alphabet_fastq(30, n_rec=6000, L=100, spread=True), hot, synth.illumina_fastq(6000, first=7001)[:-1]]
    cfg = Config.from_levels(3, 2)
    monkeypatch.setenv("DSRC_GPU_DEBUG", "1")
    _check(gpu_hooks, oracle, cfg, chunks)
    err = capfd.readouterr().err
    assert "8 of 8 streams tried, 2 handed back" in err, err          # the spread qualities (rows) and the hot block's bases (a bucket > BK_LIMIT); its qualities
                                                                      # (contexts of ~470 k symbols, fourteen rescales each) stay in their buckets
    for env in ({"DSRC_GPU_BUCKETS": "0"}, {"DSRC_GPU_BUCKETS_BINNED": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        _check(gpu_hooks, oracle, cfg, chunks)
        for k in env:
            monkeypatch.delenv(k)


def test_bucketed_path_lossy_and_eight_symbol_dna(gpu, oracle):
    """21 key bits (eleven of them inside a bucket, rows through the 16-bit map): the lossy quality model of -q2 and the 8-symbol
    DNA model of -d3 on reads with ambiguity codes of high quality."""
    from tests.cases import alphabet_fastq
    _check(gpu, oracle, Config.from_levels(3, 2), [alphabet_fastq(20, n_rec=8000, L=120, iupac=True)])
    _check(gpu, oracle, Config.from_levels(3, 2, True), [alphabet_fastq(20, n_rec=8000, L=120, iupac=True, q_max=42)])


@pytest.mark.parametrize("lanes", ["2", "1"])
def test_queue_form_two_lanes_carry_the_state(gpu, oracle, lanes, monkeypatch):
    """The queue form on the device with batches in flight on both scheduler lanes of a handle (round 4): chunks whose titles have
    5, 9, 17, 9, 3, 17, 5 fields -- the capacity of TagStats::fields changes from batch to batch and every block depends on it --,
    three flushes before anything is collected, at -d3 -q2 (bucketed front end, range coder on each lane's own stream)."""
    import random
    monkeypatch.setenv("DSRC_GPU_QUEUE_LANES", lanes)
    rng = random.Random(7)
    chunks = []
    for k, nf in enumerate((5, 9, 17, 9, 3, 17, 5)):
        recs = []
        for i in range(1500):
            title = b"@r.%d" % (10000 * k + i) + b"".join(b":%d" % ((7 * i + f) % 90 + 10) for f in range(nf - 2))
            recs.append(title + b"\n" + bytes(rng.choice(b"ACGT") for _ in range(60)) + b"\n+\n" + bytes(33 + rng.randint(20, 40) for _ in range(60)))
        chunks.append(b"\n".join(recs))
    cfg = Config.from_levels(3, 2)
    want = oracle.compress_blocks_state(cfg, chunks)
    h = gpu.Handle(cfg.dna_order, cfg.quality_order)
    got = []
    for lo, hi in ((0, 1), (1, 3), (3, 4)):
        for i in range(lo, hi):
            assert h.submit(i, chunks[i])
        h.flush()
    for lo in (4, 5, 6):
        while True:
            r = h.collect()
            if r is None:
                break
            got.append(r)
        assert h.submit(lo, chunks[lo]); h.flush()
    while True:
        r = h.collect()
        if r is None:
            break
        got.append(r)
    assert h.get_fields_capacity() == oracle.last_fields_cap
    h.close()
    assert [g[0] for g in got] == list(range(7))
    assert [(g[1], g[2], g[3]) for g in got] == want
