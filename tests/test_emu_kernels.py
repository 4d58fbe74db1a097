"""Kernel logic on the CPU: the *same* kernel sources (dsrc_amd/csrc/*.h, dsrc_gpu.hip) compiled against
the HIP emulator in tests/emu and driven through the C ABI, compared with the oracle.  This is a test
harness for a GPU-less container, not a product path (the product loads only libdsrc_gpu.so)."""
import os
import subprocess

import pytest

from dsrc_amd import synth
from tests._oracle import Config
from tests.cases import LEVELS, TINY, fuzz_fastq, fuzz_solid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libdsrc_emu.so")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")], stdout=subprocess.DEVNULL)
    old = os.environ.get("DSRC_GPU_LIB")
    os.environ["DSRC_GPU_LIB"] = EMU
    from dsrc_amd import _lib
    _lib._lib = None
    yield _lib
    _lib._lib = None
    if old is None:
        os.environ.pop("DSRC_GPU_LIB", None)
    else:
        os.environ["DSRC_GPU_LIB"] = old


def run(emu, cfg, data):
    h = emu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset)
    try:
        return h.compress_block(data)
    finally:
        h.close()


@pytest.mark.parametrize("d,q,lossy,crc", LEVELS)
def test_tiny(emu, oracle, d, q, lossy, crc):
    cfg = Config.from_levels(d, q, lossy, crc)
    assert run(emu, cfg, TINY) == oracle.compress_block(cfg, TINY)


@pytest.mark.parametrize("d,q,lossy,crc", [(3, 2, False, False), (0, 0, False, True), (2, 1, True, False)])
def test_illumina(emu, oracle, d, q, lossy, crc):
    data = synth.illumina_fastq(150)[:-1]
    cfg = Config.from_levels(d, q, lossy, crc)
    assert run(emu, cfg, data) == oracle.compress_block(cfg, data)


@pytest.mark.parametrize("d,q,lossy", [(2, 1, True), (0, 0, False), (0, 2, False)])
def test_iontorrent(emu, oracle, d, q, lossy):
    data = synth.iontorrent_fastq(120)[:-1]
    cfg = Config.from_levels(d, q, lossy)
    assert run(emu, cfg, data) == oracle.compress_block(cfg, data)


@pytest.mark.parametrize("seed", range(24))
def test_fuzz(emu, oracle, seed):
    data, desc = fuzz_fastq(seed, nrec=[2, 3, 10, 60, 150][seed % 5])
    for d, q, lossy, crc in [(0, 0, False, False), (3, 2, False, True), (2, 1, True, False)]:
        cfg = Config.from_levels(d, q, lossy, crc)
        try:
            want = oracle.compress_block(cfg, data)
        except RuntimeError as e:
            assert "rc=-2" in str(e)
            with pytest.raises(emu.DsrcGpuError):
                run(emu, cfg, data)
            continue
        assert run(emu, cfg, data) == want, (seed, desc, d, q, lossy, crc)


def test_batch_state_and_queue_api(emu, oracle):
    import ctypes as C
    from tests._oracle import _orc_cfg
    chunks = [synth.illumina_fastq(60, first=1 + 60 * k)[:-1] for k in range(3)] + [synth.iontorrent_fastq(40)[:-1]]
    cfg = Config.from_levels(0, 1)
    h = emu.Handle(cfg.dna_order, cfg.quality_order)
    for i, c in enumerate(chunks):
        h.submit(10 + i, c)
    h.flush()
    got = []
    while True:
        r = h.collect()
        if r is None:
            break
        got.append(r)
    h.close()
    assert [g[0] for g in got] == [10, 11, 12, 13]
    cap = C.c_uint32(0); c = _orc_cfg(cfg)
    for i, ch in enumerate(chunks):
        out = (C.c_uint8 * (len(ch) + 65536))(); osz = C.c_uint64(); raw = (C.c_uint64 * 4)(); comp = (C.c_uint64 * 4)()
        assert oracle.lib.orc_compress_block_state(C.byref(c), C.byref(cap), ch, C.c_uint64(len(ch)), out, C.c_uint64(len(out)), C.byref(osz), raw, comp) == 0
        assert got[i][1] == bytes(out[:osz.value])
        assert got[i][2] == list(raw) and got[i][3] == list(comp)


def test_chain_hands_state_between_handles(emu, oracle):
    """Two handles taking consecutive batches of one archive through a chain give the blocks of one handle fed in
    order (the reference's -t1 history of TagStats::fields' capacity)."""
    import ctypes as C
    from tests._oracle import _orc_cfg
    # field counts 5, 9, 9, 17: the capacity grows across batches
    import random
    rng = random.Random(11)

    def fq(nf, n, first):
        recs = []
        for i in range(n):
            title = f"@r.{first + i}" + "".join(f":{(7 * i + k) % 90 + 10}" for k in range(nf - 2))
            seq = "".join(rng.choice("ACGT") for _ in range(40))
            qua = "".join(chr(33 + rng.randint(20, 40)) for _ in range(40))
            recs.append(f"{title}\n{seq}\n+\n{qua}")
        return "\n".join(recs).encode()
    chunks = [fq(5, 30, 1), fq(9, 30, 100), fq(9, 30, 200), fq(17, 30, 300)]
    cfg = Config.from_levels(0, 0)
    want = []
    cap = C.c_uint32(0); c = _orc_cfg(cfg)
    for ch in chunks:
        out = (C.c_uint8 * (len(ch) + 65536))(); osz = C.c_uint64(); raw = (C.c_uint64 * 4)(); comp = (C.c_uint64 * 4)()
        assert oracle.lib.orc_compress_block_state(C.byref(c), C.byref(cap), ch, C.c_uint64(len(ch)), out, C.c_uint64(len(out)), C.byref(osz), raw, comp) == 0
        want.append(bytes(out[:osz.value]))
    chain = emu.Chain()
    ha = emu.Handle(cfg.dna_order, cfg.quality_order); hb = emu.Handle(cfg.dna_order, cfg.quality_order)
    ha.set_chain(chain, 0); got = [r[0] for r in ha.compress_batch(chunks[:2])]
    hb.set_chain(chain, 1); got += [r[0] for r in hb.compress_batch(chunks[2:3])]
    ha.set_chain(chain, 2); got += [r[0] for r in ha.compress_batch(chunks[3:])]
    ha.close(); hb.close(); chain.close()
    assert got == want
    # without the chain the second handle starts from an empty history and (for this input) codes differently
    hc = emu.Handle(cfg.dna_order, cfg.quality_order)
    alone = hc.compress_batch(chunks[2:3])[0][0]; hc.close()
    assert alone != want[2]


def test_short_streams_end_inside_a_chunk(emu, oracle):
    """72 symbols per stream: the second 384-byte chunk of k_rc's record layout holds eight records and must still have its room
    (fuzz seed 420 of tests/test_gpu_parity.py::test_field_filter found the arrays of two streams overlapping)."""
    from tests.cases import fuzz_fastq
    data, _ = fuzz_fastq(420, None)
    for d, q, lossy in [(3, 2, False), (1, 1, False), (2, 2, True)]:
        cfg = Config.from_levels(d, q, lossy)
        assert run(emu, cfg, data) == oracle.compress_block(cfg, data)


def _state_chunks():
    import random
    rng = random.Random(11)

    def fq(nf, n, first):
        recs = []
        for i in range(n):
            title = f"@r.{first + i}" + "".join(f":{(7 * i + k) % 90 + 10}" for k in range(nf - 2))
            seq = "".join(rng.choice("ACGT") for _ in range(40))
            qua = "".join(chr(33 + rng.randint(20, 40)) for _ in range(40))
            recs.append(f"{title}\n{seq}\n+\n{qua}")
        return "\n".join(recs).encode()
    # field counts that make the carried capacity grow across sub-batches
    return [fq(5, 30, 1), fq(9, 30, 100), fq(9, 30, 200), fq(17, 30, 300), fq(5, 25, 400), fq(33, 20, 500), fq(9, 30, 600)]


@pytest.mark.parametrize("lanes,sub", [(3, 1), (2, 2), (5, 3)])
def test_scheduler_lanes_inside_a_handle(emu, oracle, lanes, sub):
    """dsrcgpu_set_lanes: a batch call cut into sub-batches that run on lanes inside the handle gives the blocks -- and leaves the
    state -- of the same call on one lane (the reference's -t1 history), for host-resident and device-resident chunks."""
    chunks = _state_chunks()
    for d, q, lossy, crc in [(0, 0, False, False), (1, 1, False, True)]:
        cfg = Config.from_levels(d, q, lossy, crc)
        want = oracle.compress_blocks_state(cfg, chunks + chunks[:2])
        h = emu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset, verify=crc)
        h.set_lanes(lanes, sub)
        got = h.compress_batch(chunks)
        assert [g[0] for g in got] == [w[0] for w in want[:len(chunks)]]
        assert [g[1:] for g in got] == [w[1:] for w in want[:len(chunks)]]
        # the state left behind is the one the next call starts from (here on the device-resident form)
        blob = b"\n".join(chunks[:2])
        offs = [0, len(chunks[0]) + 1]; sizes = [len(chunks[0]), len(chunks[1])]
        d_in = h.dev_alloc(len(blob) + 64); h.dev_upload(d_in, blob)
        cap = 1 << 20; d_out = h.dev_alloc(cap)
        o_offs, o_sizes, _, _ = h.compress_batch_device(d_in, offs, sizes, d_out, cap)
        out = h.dev_download(d_out, o_offs[-1] + o_sizes[-1])
        assert [out[o_offs[i]: o_offs[i] + o_sizes[i]] for i in range(2)] == [w[0] for w in want[len(chunks):]]
        # an output buffer that is too small: DSRCGPU_E_CAPACITY, and the state is where it was
        before = h.get_fields_capacity()
        with pytest.raises(emu.DsrcGpuError) as e:
            h.compress_batch_device(d_in, offs, sizes, d_out, o_sizes[0] + 5)
        assert e.value.code == -4
        assert h.get_fields_capacity() == before
        h.dev_free(d_in); h.dev_free(d_out)
        h.close()


def test_queue_form_is_asynchronous_and_ordered(emu, oracle):
    """submit / flush / collect / release: several batches flushed before anything is collected (the ring holds the lanes + 2), blocks
    come back in submission order with the state carried across batches, try_collect never blocks, and a ring slot is
    reused only after its blocks were released.  Every other chunk goes in through dsrcgpu_submit_pinned: read where it lies."""
    import ctypes as C
    chunks = [synth.illumina_fastq(30, first=1 + 30 * k)[:-1] for k in range(10)]
    cfg = Config.from_levels(0, 1)
    want = oracle.compress_blocks_state(cfg, chunks)
    h = emu.Handle(cfg.dna_order, cfg.quality_order)
    assert h.collect(wait=False) is None
    pinned = emu.host_alloc(sum(len(c) for c in chunks) + 4096)
    at = 0

    def put(i):
        nonlocal at
        if i % 2:
            return h.submit(100 + i, chunks[i])
        C.memmove(pinned + at, chunks[i], len(chunks[i]))
        ok = h.submit_pinned(100 + i, pinned + at, len(chunks[i]))
        if ok:
            at += len(chunks[i]) + 64
        return ok
    for lo, hi in ((0, 2), (2, 4), (4, 5), (5, 7), (7, 8)):           # five batches in flight: three lanes + 2
        for i in range(lo, hi):
            assert put(i)
        h.flush()
    assert put(8) is False                                      # ring full: reported, not waited for (the caller is the collector)
    got = []
    while True:
        r = h.collect()
        if r is None:
            break
        got.append(r)
    for i in (8, 9):                                            # the ring slots are free again
        assert put(i); h.flush()
    while True:
        r = h.collect()
        if r is None:
            break
        got.append(r)
    h.close()
    emu.host_free(pinned)
    assert [g[0] for g in got] == [100 + i for i in range(10)]
    assert [(g[1], g[2], g[3]) for g in got] == want


def test_shard_seeding_gives_the_t1_blocks(emu, oracle):
    """A handle that starts in the middle of an archive (another GPU / rank) is seeded with the fold of
    dsrcgpu_fields_capacity_after over the first title of every chunk before its shard; its blocks are then those of one
    handle fed everything in order.  Also with a field filter, where the field count is that of the rewritten title."""
    import dataclasses
    import random
    rng = random.Random(12)

    def fq(nf, n, first):
        recs = []
        for i in range(n):
            title = f"@r.{first + i}" + "".join(f":{(7 * i + k) % 90 + 10}" for k in range(nf - 2))
            seq = "".join(rng.choice("ACGT") for _ in range(30))
            qua = "".join(chr(33 + rng.randint(20, 40)) for _ in range(30))
            recs.append(f"{title}\n{seq}\n+\n{qua}")
        return "\n".join(recs).encode()
    chunks = [fq(5, 25, 1), fq(9, 25, 100), fq(9, 25, 200), fq(17, 25, 300), fq(3, 25, 400), fq(17, 25, 500)]
    for flags in (0, 0b1111010):
        cfg = dataclasses.replace(Config.from_levels(0, 0), tag_flags=flags)
        want = [b for b, _, _ in oracle.compress_blocks_state(cfg, chunks)]
        caps = [emu.fields_capacity_fold(chunks[:k], flags) for k in range(len(chunks) + 1)]
        assert caps[-1] == oracle.last_fields_cap
        got = []
        for lo, hi in ((0, 2), (2, 3), (3, 6)):
            h = emu.Handle(cfg.dna_order, cfg.quality_order, tag_flags=flags)
            h.set_fields_capacity(caps[lo])
            got += [r[0] for r in h.compress_batch(chunks[lo:hi])]
            assert h.get_fields_capacity() == caps[hi]
            h.close()
        assert got == want, flags
        if flags == 0:
            h = emu.Handle(cfg.dna_order, cfg.quality_order)        # unseeded: the shard starts from an empty history
            assert h.compress_batch(chunks[2:3])[0][0] != want[2]
            h.close()
            # a chain (several instances per device) that starts mid-archive
            chain = emu.Chain(); chain.seed(caps[2])
            h = emu.Handle(cfg.dna_order, cfg.quality_order); h.set_chain(chain, 0)
            assert [r[0] for r in h.compress_batch(chunks[2:4])] == want[2:4]
            h.close(); chain.close()


def test_chain_survives_a_capacity_retry(emu, oracle):
    """A batch that comes back with DSRCGPU_E_CAPACITY has already taken its turn in the chain; announcing the same
    batch again and retrying with a larger buffer must neither wait for that turn a second time (it deadlocked) nor
    pick up state that later batches have published since."""
    import threading
    chunks = [synth.illumina_fastq(40, first=1 + 40 * k)[:-1] for k in range(3)]
    cfg = Config.from_levels(0, 0)
    hs = emu.Handle(cfg.dna_order, cfg.quality_order)
    want = [r[0] for r in hs.compress_batch(chunks)]; hs.close()
    chain = emu.Chain()
    ha = emu.Handle(cfg.dna_order, cfg.quality_order); hb = emu.Handle(cfg.dna_order, cfg.quality_order)
    ha.set_chain(chain, 0)
    with pytest.raises(emu.DsrcGpuError) as ei:
        ha.compress_batch(chunks[:2], cap=64)
    assert ei.value.code == -4
    hb.set_chain(chain, 1); got_b = [r[0] for r in hb.compress_batch(chunks[2:])]      # the chain has moved on meanwhile
    res = []
    def retry():
        ha.set_chain(chain, 0); res.append([r[0] for r in ha.compress_batch(chunks[:2])])
    t = threading.Thread(target=retry, daemon=True); t.start(); t.join(60)
    assert not t.is_alive(), "the retry waits for a chain turn it has already taken"
    assert res[0] + got_b == want
    ha.close(); hb.close(); chain.close()


def test_hot_contexts_rescale(emu, oracle):
    """One context with ~40k symbols: exercises the epoch/rescale path of k_replay and multi-wave ranges."""
    import random
    rng = random.Random(5)
    recs = []
    for i in range(170):
        seq = ''.join(rng.choice('AAAAAAAC') for _ in range(250))
        q = ''.join('I' if rng.random() < 0.97 else 'H' for _ in range(250))
        recs.append(f"@r.{i}\n{seq}\n+\n{q}")
    data = '\n'.join(recs).encode()
    for d, q, lossy in [(1, 2, False), (2, 1, True)]:
        cfg = Config.from_levels(d, q, lossy)
        assert run(emu, cfg, data) == oracle.compress_block(cfg, data)


def test_range_coder_reference_loop_path(emu, oracle, monkeypatch):
    """The carry-clamp fallback of k_rc (reference loop + byte re-dealing) must give the same stream as the fast
    path; DSRC_GPU_FORCE_EXACT_RC sends every 16-symbol group through it."""
    monkeypatch.setenv("DSRC_GPU_FORCE_EXACT_RC", "1")
    data = synth.illumina_fastq(300)[:-1]
    for d, q, lossy in [(3, 2, False), (2, 1, True)]:
        cfg = Config.from_levels(d, q, lossy)
        assert run(emu, cfg, data) == oracle.compress_block(cfg, data)


@pytest.mark.parametrize("hook", ["DSRC_GPU_RC_REDO", "DSRC_GPU_RC_RECOVER", "DSRC_GPU_RC_ONE_LANE"])
def test_range_coder_redo_list_and_one_lane_kernel(emu, oracle, monkeypatch, hook):
    """k_rcs (two waves: range, low) + the redo list coded by k_rc, and k_rc for the whole batch: same blocks (tests/test_gpu_parity.py)."""
    monkeypatch.setenv(hook, "1")
    data = synth.illumina_fastq(300)[:-1]
    for d, q, lossy in [(3, 2, False), (2, 1, True)]:
        cfg = Config.from_levels(d, q, lossy)
        assert run(emu, cfg, data) == oracle.compress_block(cfg, data)
    cfg = Config.from_levels(1, 1)
    chunks = [synth.illumina_fastq(120, first=1 + 200 * k)[:-1] for k in range(5)] + [synth.illumina_fastq(3)[:-1]]
    h = emu.Handle(cfg.dna_order, cfg.quality_order)
    got = h.compress_batch(chunks)
    h.close()
    assert [g[0] for g in got] == [oracle.compress_block(cfg, c)[0] for c in chunks]


def test_range_coder_wide_workgroups(emu, oracle, monkeypatch):
    """k_rcs<32> (launches of more than 512 streams on the GPU) on the emulator's small batches: DSRC_GPU_RC_WIDE_FROM=0; with the forced
    recoveries as well."""
    monkeypatch.setenv("DSRC_GPU_RC_WIDE_FROM", "0")
    cfg = Config.from_levels(3, 2)
    chunks = [synth.illumina_fastq(150, first=1 + 200 * k)[:-1] for k in range(3)] + [synth.illumina_fastq(3)[:-1]]
    want = [oracle.compress_block(cfg, c)[0] for c in chunks]
    for hook in (None, "DSRC_GPU_RC_RECOVER"):
        if hook:
            monkeypatch.setenv(hook, "1")
        h = emu.Handle(cfg.dna_order, cfg.quality_order)
        got = h.compress_batch(chunks)
        h.close()
        assert [g[0] for g in got] == want


def test_staging_overflow_second_pass(emu, oracle, monkeypatch):
    """A range-coded stream that outgrows the estimate of its staging (1 1/16 bytes per symbol) is detected by the coder, never written
    past, and the batch is run again with the two bytes per symbol that cannot be exceeded.  DSRC_GPU_HOOK_RC_BOUND_SHIFT makes the
    estimate far too small, so that this really happens: with a field filter on the device-resident form (the kernels rewrite the
    text: the second pass must find it as it came, and so must the caller), and with a fixed arena."""
    import dataclasses
    monkeypatch.setenv("DSRC_GPU_HOOK_RC_BOUND_SHIFT", "3")
    chunks = [synth.illumina_fastq(200, first=1 + 300 * k)[:-1] for k in range(3)]
    for flags, fixed in ((0b1010, 0), (0, 64 << 20), (0b10, 96 << 20)):
        cfg = dataclasses.replace(Config.from_levels(2, 1), tag_flags=flags)
        want = [oracle.compress_block(cfg, c)[0] for c in chunks]
        h = emu.Handle(cfg.dna_order, cfg.quality_order, tag_flags=flags, arena_bytes=fixed)
        blob = b"\n".join(chunks)
        offs = []; at = 0
        for c in chunks:
            offs.append(at); at += len(c) + 1
        d_in = h.dev_alloc(len(blob) + 64); h.dev_upload(d_in, blob)
        cap = 1 << 20; d_out = h.dev_alloc(cap)
        o_offs, o_sizes, _, _ = h.compress_batch_device(d_in, offs, [len(c) for c in chunks], d_out, cap)
        out = h.dev_download(d_out, o_offs[-1] + o_sizes[-1])
        assert [out[o_offs[i]: o_offs[i] + o_sizes[i]] for i in range(3)] == want, (flags, fixed)
        assert h.dev_download(d_in, len(blob)) == blob                      # the caller's text is as it was
        # the handle has learnt: the next batch takes one pass (same blocks)
        o_offs, o_sizes, _, _ = h.compress_batch_device(d_in, offs[:1], [len(chunks[0])], d_out, cap)
        assert h.dev_download(d_out, o_sizes[0]) == oracle.compress_block(cfg, chunks[0])[0]
        h.dev_free(d_in); h.dev_free(d_out); h.close()


def test_sort_ballot_variant(emu, oracle, monkeypatch):
    """k_sort has two ranking variants (LDS atomics where the device applies them in lane order, else ballots): same blocks."""
    monkeypatch.setenv("DSRC_GPU_SORT_BALLOT", "1")
    data = synth.illumina_fastq(300)[:-1]
    for d, q, lossy in [(3, 2, False), (2, 1, True), (1, 2, False)]:
        cfg = Config.from_levels(d, q, lossy)
        assert run(emu, cfg, data) == oracle.compress_block(cfg, data)


def test_exact_division_selftest(emu):
    h = emu.Handle()
    assert h.selftest() == 0
    h.close()


def test_device_synth_matches_host(emu):
    h = emu.Handle()
    cap = 1 << 20
    d = h.dev_alloc(cap)
    n = h.synth_illumina(999990, 1200, d, cap)
    got = h.dev_download(d, n)
    h.dev_free(d); h.close()
    assert got == synth.illumina_fastq(1200, first=999990)


@pytest.mark.parametrize("seed", range(400, 405))
def test_field_filter(emu, oracle, seed):
    """-f: titles rewritten in place by k_tag_filter (FastqParserExt), incl. the kept last field that takes the line
    terminator along and the index-transformed base the tokenizer then sees (k_tag_poke)."""
    import dataclasses
    data, desc = fuzz_fastq(seed, [30, 40, 120][seed % 3])
    for flags in (0b10, 0b1010, 0b11110, 0x7FFFFFFE):
        for d, q, lossy, crc in [(0, 0, False, True), (1, 1, False, False), (2, 1, True, True)]:
            cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), tag_flags=flags)
            h = emu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset, tag_flags=flags)
            try:
                want = oracle.compress_block(cfg, data)
            except RuntimeError as e:
                assert "rc=-2" in str(e)
                with pytest.raises(emu.DsrcGpuError):
                    h.compress_block(data)
                h.close()
                continue
            got = h.compress_block(data)
            h.close()
            assert got == want, (seed, desc, bin(flags), d, q, lossy, crc)


@pytest.mark.parametrize("seed", range(500, 506))
def test_record_layout(emu, oracle, seed):
    """dsrcgpu_set_record_layout: chunks assembled from records (BlockCompressorExt) -- chunkSize word given by the
    caller, last title separator = the index-transformed first base; block-to-block state carried as usual."""
    data, desc = fuzz_fastq(seed, [30, 60, 150][seed % 3])
    data = data.replace(b"\r\n", b"\n")                   # record strings hold no line terminators
    other = synth.illumina_fastq(40, first=7)[:-1]
    for d, q, lossy in [(0, 0, False), (2, 0, False), (1, 1, True), (3, 2, True)]:
        # settings mapping of the archive API: qualityOrder = 3 * level also when lossless (src/DsrcArchive.cpp:41-42)
        cfg = Config(dna_order=3 * d, quality_order=3 * q, lossy=lossy)
        h = emu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, False, cfg.quality_offset)
        try:
            w0, cap = oracle.compress_records_block(cfg, data, 123456789)
            w1, cap = oracle.compress_records_block(cfg, other, 0xFFFFFFF0 + 77, cap)
            w2, _ = oracle.compress_records_block(cfg, data, 5, cap)
        except RuntimeError as e:
            assert "rc=-2" in str(e)
            h.close()
            continue
        h.set_record_layout([123456789, 0xFFFFFFF0 + 77])
        got = h.compress_batch([data, other])
        assert got[0][0] == w0 and got[1][0] == w1, (seed, desc, d, q, lossy)
        h.set_record_layout([5])
        assert h.compress_batch([data])[0][0] == w2
        assert h.compress_batch([data])[0][0] != w2            # one-shot: back to the text layout
        h.close()


def test_record_layout_arguments(emu):
    h = emu.Handle(crc=True)
    h.set_record_layout([1])
    with pytest.raises(emu.DsrcGpuError):
        h.compress_batch([TINY])
    h.close()
    h = emu.Handle()
    h.set_record_layout([1, 2])
    with pytest.raises(emu.DsrcGpuError):
        h.compress_batch([TINY])                            # one chunkSize per chunk
    assert h.compress_batch([TINY])[0][0]                   # cleared by the failed call
    h.close()


@pytest.mark.parametrize("seed", range(8))
def test_color_space(emu, oracle, seed):
    """SOLiD: colours decoded in place by k_cs_decode (prefix xor), constant-primer blocks shortened by k_cs_reduce after
    the statistics, meta with FLAG_DELTA_CONSTANT + csSeqBegin/csQuaBegin."""
    import dataclasses
    data, desc = fuzz_solid(seed, [5, 40, 130, 70][seed % 4])
    for d, q, lossy, crc in LEVELS:
        cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), color_space=True)
        h = emu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset, color_space=True)
        try:
            want = oracle.compress_block(cfg, data)
        except RuntimeError as e:
            assert "rc=-2" in str(e)
            h.close()
            continue
        got = h.compress_batch([data, TINY_CS])
        h.close()
        assert got[0] == want, (seed, desc, d, q, lossy, crc)


def test_hot_contexts_cross_ranges(emu, oracle):
    """A context that holds most of a stream (> 32k symbols: several Rescale() calls) spans many replay ranges: the seam
    states of k_replay_seams must let every range start in the middle of the segment."""
    import random
    rng = random.Random(11)
    recs = []
    for i in range(250):
        seq = ''.join(rng.choice('AAAAAAAAAAAAAAAC') for _ in range(250))
        q = ''.join('I' if rng.random() < 0.98 else 'H' for _ in range(250))
        recs.append(f"@r.{i}\n{seq}\n+\n{q}")
    data = '\n'.join(recs).encode()
    for d, q, lossy in [(1, 2, False), (2, 2, True)]:
        cfg = Config.from_levels(d, q, lossy)
        assert run(emu, cfg, data) == oracle.compress_block(cfg, data), (d, q, lossy)


def test_hot_buckets_windows_inside_a_run(emu, oracle, monkeypatch):
    """Four-level qualities, skewed: a few contexts hold most of the stream.  The windows of a large bucket are looked for inside
    one tile's run first (k_model, BK_BIG) and through the head / scatter bookkeeping otherwise; rescales in between.  The quality
    contexts of a read of 250 split by position class, so the size from which a bucket counts as large is lowered to reach that
    path with a block the emulator finishes in seconds (full size: tests/test_gpu_properties.py::test_full_size_blocks[binned])."""
    import random
    monkeypatch.setenv("DSRC_GPU_BUCKET_BIG", "512")
    for seed, weights in [(3, (70, 15, 10, 5)), (4, (94, 3, 2, 1)), (5, (40, 30, 20, 10))]:
        rng = random.Random(seed)
        recs = []
        for i in range(220):
            seq = ''.join(rng.choice('ACGT') for _ in range(250))
            q = ''.join(rng.choices('FA<,', weights=weights, k=250))
            recs.append(f"@r.{i}\n{seq}\n+\n{q}")
        data = '\n'.join(recs).encode()
        for d, q, lossy in [(3, 2, False), (2, 2, True)]:
            cfg = Config.from_levels(d, q, lossy)
            assert run(emu, cfg, data) == oracle.compress_block(cfg, data), (seed, d, q, lossy)


def test_rle_quality_alphabets(emu, oracle):
    """RLE quality scheme with 4 / 20 / 45 distinct values: LDS code tables and histograms, and their global fallbacks."""
    from tests.cases import rle_chunks
    cfg = Config.from_levels(0, 0)
    for c in rle_chunks():
        c = c[:len(c) // 8]; c = c[:c.rfind(b"\n@")]
        want = oracle.compress_block(cfg, c)
        assert want[0][want[2][0] + want[2][1]] == 2
        assert run(emu, cfg, c) == want


def test_color_space_golden_small(emu):
    """The small -q0 blocks of the reference's SOLiD golden set (all three quality schemes; the RLE modeler takes its
    alphabet from the runs of the shortened records, not from the statistics)."""
    import dataclasses, hashlib, json
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "solid_golden.json")))
    n = 0
    for e in G["blocks"]:
        d, q, lossy, crc = e["levels"]
        if q != 0 or e["size"] > 5000:
            continue
        data = fuzz_solid(e["seed"], e["nrec"])[0]
        cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), color_space=True)
        h = emu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset, color_space=True)
        blk = h.compress_block(data)[0]
        h.close()
        assert hashlib.sha256(blk).hexdigest() == e["sha256"], e
        n += 1
    assert n >= 20


TINY_CS = b"@s.1 a_1\nT0120.312\n+\n!IIII#III\n@s.2 a_2\nT3321..01\n+\n!HHHH!!HH"


def test_bad_arguments(emu):
    with pytest.raises(emu.DsrcGpuError):
        emu.Handle(tag_flags=1 << 31)          # field numbers above 30 are undefined in the reference (32-bit BIT())
    with pytest.raises(emu.DsrcGpuError):
        emu.Handle(quality_offset=20)
    h = emu.Handle(color_space=True, tag_flags=0b110)          # colour space + field filter: refused at the first batch
    with pytest.raises(emu.DsrcGpuError):
        h.compress_block(fuzz_solid(1, 5)[0])
    h.close()
    h = emu.Handle()
    with pytest.raises(emu.DsrcGpuError):
        h.compress_block(b"not a fastq chunk")
    h.close()


def test_capacity_failure_leaves_the_block_to_block_state(emu, oracle):
    """A batch that fails with DSRCGPU_E_CAPACITY has already folded its first titles into the capacity of the reference's
    TagStats::fields vector; the handle must forget that, or the grow-and-retry pattern writes blocks
    that differ from a fresh pass (record 0's extra count of a numeric field survives only behind the last reallocation)."""
    recs = lambda first, n: b"\n".join(b"@r.%d a:%d b:%d c:%d d:%d e:%d f:%d g:%d\nACGT\n+\nIIII" % (i, i % 7, i % 5, (i * 3) % 11, i % 3, i % 13, i % 2, i % 17)
                                          for i in range(first, first + n))
    chunks = [recs(1, 60), recs(61, 60)]
    cfg = Config.from_levels(0, 0)
    want = [oracle_block for oracle_block, _, _ in oracle.compress_blocks_state(cfg, chunks)]
    h = emu.Handle(cfg.dna_order, cfg.quality_order)
    with pytest.raises(emu.DsrcGpuError) as ei:
        h.compress_batch(chunks, cap=64)
    assert ei.value.code == -4
    got = [r[0] for r in h.compress_batch(chunks)]
    h.close()
    assert got == want


def test_release_memory_between_phases(emu, oracle):
    """dsrcgpu_release_memory hands the arena and the table region back; the next call allocates again and the block-to-block
    state (TagStats::fields capacity) is what it was."""
    chunks = [synth.illumina_fastq(60, first=1 + 60 * k)[:-1] for k in range(3)]
    cfg = Config.from_levels(2, 2, False, True)
    want = [b[0] for b in oracle.compress_blocks_state(cfg, chunks + chunks)]
    h = emu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset)
    try:
        got = [g[0] for g in h.compress_batch(chunks)]
        h.release_memory()
        got += [g[0] for g in h.compress_batch(chunks)]
        assert got == want
        h.release_memory(); h.release_memory()
        texts = h.decompress_batch(got[:3])
        h.release_memory()
        assert texts == h.decompress_batch(got[:3]) == [c + b"\n" for c in chunks]
        # ... and dsrcgpu_reserve_memory asks for both ahead of a call: larger than the call needs, smaller (nothing shrinks), none
        h.release_memory()
        h.reserve_memory(32 << 20, 8 << 20)
        assert h.decompress_batch(got[:3]) == texts
        h.reserve_memory(1 << 20, 0); h.reserve_memory(0, 0)
        assert len(h.compress_batch(chunks)) == 3                  # (a batch call after it; its blocks' state has moved on)
        assert h.decompress_batch(got[:3]) == texts
    finally:
        h.close()


@pytest.mark.parametrize("lanes", ["2", "1"])
def test_queue_form_two_lanes_carry_the_state(emu, oracle, lanes, monkeypatch):
    """The queue form runs consecutive batches on two scheduler lanes (the handle and its twin: the range coder of one batch next to
    the front end of the next); what DSRC carries from block to block -- the capacity of TagStats::fields -- goes from lane to lane
    through the handle's own chain, in flush order.  Titles with 5, 9, 17, 9, 3, 17 fields, one or two chunks per flush, a handle
    seeded like a shard that starts inside an archive; DSRC_GPU_QUEUE_LANES=1 is the one-lane form."""
    import random
    monkeypatch.setenv("DSRC_GPU_QUEUE_LANES", lanes)
    rng = random.Random(7)
    chunks = []
    for k, nf in enumerate((5, 9, 17, 9, 3, 17, 5)):
        recs = []
        for i in range(25):
            title = b"@r.%d" % (100 * k + i) + b"".join(b":%d" % ((7 * i + f) % 90 + 10) for f in range(nf - 2))
            recs.append(title + b"\n" + bytes(rng.choice(b"ACGT") for _ in range(36)) + b"\n+\n" + bytes(33 + rng.randint(20, 40) for _ in range(36)))
        chunks.append(b"\n".join(recs))
    cfg = Config.from_levels(0, 1)
    for seed in (0, 4):
        want = oracle.compress_blocks_state(cfg, chunks, fields_cap=seed)
        h = emu.Handle(cfg.dna_order, cfg.quality_order)
        if seed:
            h.set_fields_capacity(seed)
        got = []
        for lo, hi in ((0, 1), (1, 3), (3, 4)):                     # three batches in flight, then the rest one by one
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
        assert [(g[1], g[2], g[3]) for g in got] == want, (lanes, seed)


def _field_chunks(fields, n_rec=25, seed=7):
    import random
    rng = random.Random(seed)
    chunks = []
    for k, nf in enumerate(fields):
        recs = []
        for i in range(n_rec):
            title = b"@r.%d" % (100 * k + i) + b"".join(b":%d" % ((7 * i + f) % 90 + 10) for f in range(nf - 2))
            recs.append(title + b"\n" + bytes(rng.choice(b"ACGT") for _ in range(36)) + b"\n+\n" + bytes(33 + rng.randint(20, 40) for _ in range(36)))
        chunks.append(b"\n".join(recs))
    return chunks


def test_queue_form_second_lane_takes_the_first_batches(emu, oracle, monkeypatch):
    """Which lane runs which batch must not matter (advisor, round 4): lane 0 -- the handle itself -- starts late, so the twin takes
    batch 0, publishes the capacity it leaves (17 fields: 32) and stays busy; the next flush falls into that window.  It used to
    take `lane 0 has not run a batch yet` for `the user made batch calls` and wrote the handle's stale capacity over the published one."""
    import time
    monkeypatch.setenv("DSRC_GPU_HOOK_LANE0_DELAY_MS", "2500")
    monkeypatch.setenv("DSRC_GPU_HOOK_BATCH_HOLD_MS", "700")
    chunks = _field_chunks((17, 5, 9, 17))
    cfg = Config.from_levels(0, 1)
    want = oracle.compress_blocks_state(cfg, chunks)
    h = emu.Handle(cfg.dna_order, cfg.quality_order)
    got = []
    assert h.submit(0, chunks[0]); h.flush()
    time.sleep(0.35)                                            # batch 0 has published and is being held
    assert h.submit(1, chunks[1]); h.flush()
    for i in (2, 3):
        assert h.submit(i, chunks[i]); h.flush()
        while True:
            r = h.collect()
            if r is None:
                break
            got.append(r)
    assert h.get_fields_capacity() == oracle.last_fields_cap
    h.close()
    assert [g[0] for g in got] == [0, 1, 2, 3]
    assert [(g[1], g[2], g[3]) for g in got] == want


@pytest.mark.parametrize("lanes", ["2", "1"])
def test_queue_form_record_layout_set_while_a_batch_runs(emu, oracle, lanes, monkeypatch):
    """dsrcgpu_set_record_layout belongs to the batch of the next flush (include/dsrc_gpu.h) -- also when it is set while an earlier
    batch is still running on the handle's own lane (advisor, round 4: the scheduler thread used to clear the user's field when
    that batch ended)."""
    import time
    monkeypatch.setenv("DSRC_GPU_QUEUE_LANES", lanes)
    monkeypatch.setenv("DSRC_GPU_HOOK_BATCH_HOLD_MS", "500")
    a = synth.illumina_fastq(40, first=7)[:-1]
    b = synth.illumina_fastq(30, first=900)[:-1]
    cfg = Config(dna_order=0, quality_order=0, lossy=False)
    w0, cap = oracle.compress_records_block(cfg, a, 111)
    w1, cap = oracle.compress_records_block(cfg, b, 222, cap)
    w2 = oracle.compress_blocks_state(cfg, [a], fields_cap=cap)[0][0]
    h = emu.Handle(cfg.dna_order, cfg.quality_order)
    h.set_record_layout([111])
    assert h.submit(0, a); h.flush()
    time.sleep(0.1)                                             # batch 0 is running (held)
    h.set_record_layout([222])
    assert h.submit(1, b); h.flush()
    assert h.submit(2, a); h.flush()                            # one-shot: text layout again
    got = []
    while True:
        r = h.collect()
        if r is None:
            break
        got.append(r)
    h.close()
    assert [g[1] for g in got] == [w0, w1, w2]


def test_queue_form_then_batch_calls_on_one_handle(emu, oracle):
    """Batch calls on a handle whose queue form runs two lanes: refused while flushed batches are in flight, allowed once the queue
    has drained -- and the block-to-block state goes on from one form to the other and back."""
    import random
    rng = random.Random(11)
    chunks = []
    for k, nf in enumerate((5, 17, 9, 3, 17)):
        recs = [b"@r.%d" % (100 * k + i) + b"".join(b":%d" % ((7 * i + f) % 90 + 10) for f in range(nf - 2)) + b"\n" +
                bytes(rng.choice(b"ACGT") for _ in range(36)) + b"\n+\n" + bytes(33 + rng.randint(20, 40) for _ in range(36)) for i in range(25)]
        chunks.append(b"\n".join(recs))
    cfg = Config.from_levels(0, 1)
    want = oracle.compress_blocks_state(cfg, chunks)
    h = emu.Handle(cfg.dna_order, cfg.quality_order)
    got = []
    assert h.submit(0, chunks[0]); h.flush()
    assert h.submit(1, chunks[1]); h.flush()
    with pytest.raises(emu.DsrcGpuError) as ei:
        h.compress_batch([chunks[2]])
    assert ei.value.code == -6
    for _ in range(2):
        r = h.collect(); got.append((r[1], r[2], r[3]))
    assert h.collect() is None
    got += h.compress_batch([chunks[2]])                              # drained: the batch form, from the lanes' state
    got += h.compress_batch([chunks[3]])
    assert h.submit(4, chunks[4]); h.flush()                          # ... and back
    r = h.collect(); got.append((r[1], r[2], r[3]))
    h.close()
    assert got == want
