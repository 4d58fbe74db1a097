"""The bucketed context path (dsrc_amd/csrc/k_bucket.h: k_part, k_binoff, k_model, k_place) on the CPU emulator: every alphabet size
of the order models, every counter-row layout (rows by key, rows handed out through the byte / the 16-bit map), both hand-backs to
k_sort / k_replay (a bucket too large for one wave, more contexts in a bucket than rows), the scattering form, and the switch that
turns the path off -- always the oracle's block.  Test harness only; the product links libdsrc_gpu.so."""
import random

import pytest

from dsrc_amd import synth
from tests._oracle import Config
from tests.cases import alphabet_fastq, fuzz_fastq
from tests.test_emu_kernels import emu, run  # noqa: F401  (fixture)


def check(emu, oracle, data, levels):
    for d, q, lossy in levels:
        cfg = Config.from_levels(d, q, lossy)
        assert run(emu, cfg, data) == oracle.compress_block(cfg, data), (d, q, lossy)


@pytest.mark.parametrize("n_sym", [3, 12, 20, 40, 90])
def test_alphabet_sizes(emu, oracle, n_sym):
    """3 / 12 / 20 / 40 / 90 quality values: the 16-, 32-, 64- and 128-symbol models (k_model<16..128>: two to four levels of
    counter words), at -q1 (few key bits: rows by key) and -q2 (rows through the map)."""
    data = alphabet_fastq(n_sym, n_rec=420, L=100)
    check(emu, oracle, data, [(2, 2, False), (1, 1, False)])


def test_dna_with_eight_symbols_and_lossy_orders(emu, oracle):
    """Ambiguity codes kept in the DNA stream: the 8-symbol DNA model (k_model<8>, 21 key bits at -d3: eleven of them inside a
    bucket); the lossy quality model of -q2 (8 symbols, order 6: 21 key bits as well)."""
    data = alphabet_fastq(20, n_rec=420, L=100, iupac=True)
    check(emu, oracle, data, [(3, 2, False), (2, 1, False)])
    data = alphabet_fastq(20, n_rec=420, L=100, iupac=True, q_max=42)
    check(emu, oracle, data, [(3, 2, True), (1, 1, True)])


def test_long_streams_read_their_tile_table_by_groups(emu, oracle, monkeypatch):
    """Streams of more tiles than k_model keeps in LDS (512: 4 M symbols; the reference's -m1 / -m2 buffers give 27 M / 107 M per stream)
    read a bucket's tile offsets and counts from the count table 64 tiles at a time.  DSRC_GPU_BUCKET_NARROW_BINS=2 puts every stream
    of more than two tiles on that road: ordinary data, hot contexts (the windows inside one tile's run), several alphabets."""
    monkeypatch.setenv("DSRC_GPU_BUCKET_NARROW_BINS", "2")
    check(emu, oracle, synth.illumina_fastq(600)[:-1], [(3, 2, False), (2, 1, True)])
    check(emu, oracle, alphabet_fastq(40, n_rec=420, L=100), [(2, 2, False), (1, 1, False)])
    monkeypatch.setenv("DSRC_GPU_BUCKET_BIG", "256")
    check(emu, oracle, _hot(260), [(1, 2, False)])


def test_statistics_in_several_workgroups_per_block(emu, oracle, monkeypatch):
    """A batch of few, large blocks gives every block's statistics (k_prep_stats, k_tag_scan) to several workgroups whose sums meet in the
    block's state; DSRC_GPU_HOOK_STATS_PARTS forces that for small blocks: reads of one and of several lengths, ambiguity codes (the rare symbols'
    counters), lossy qualities, more parts than a block has groups of records."""
    for parts in ("3", "16"):
        monkeypatch.setenv("DSRC_GPU_HOOK_STATS_PARTS", parts)
        check(emu, oracle, synth.illumina_fastq(300)[:-1], [(3, 2, False), (2, 1, True)])
        check(emu, oracle, alphabet_fastq(20, n_rec=150, L=100, iupac=True), [(3, 2, False), (0, 0, False)])
    # ... and the titles' (k_tag_scan: minima, maxima and flags per field; titles of mixed formatting)
    monkeypatch.setenv("DSRC_GPU_HOOK_STATS_PARTS", "5")
    for seed in (74, 75, 61, 64, 63, 90, 123):
        check(emu, oracle, fuzz_fastq(seed)[0], [(1, 1, False), (3, 2, True)])


def test_model_runs_out_of_rows(emu, oracle, capfd, monkeypatch):
    """Independent uniform qualities: nearly every symbol of a bucket has a context of its own, k_model runs out of counter rows and
    hands the stream back to k_sort / k_replay (their launches follow k_model's in the same batch)."""
    monkeypatch.setenv("DSRC_GPU_DEBUG", "1")
    data = alphabet_fastq(30, n_rec=500, L=100, spread=True)
    check(emu, oracle, data, [(3, 2, False)])
    err = capfd.readouterr().err
    assert "2 of 2 streams tried, 1 handed back" in err, err


def _hot(n_rec, seed=3):
    rng = random.Random(seed)
    recs = []
    for i in range(n_rec):
        seq = "".join(rng.choice("AAAAAAAAAAAAAAAC") for _ in range(200))
        q = "".join("I" if rng.random() < 0.98 else "H" for _ in range(200))
        recs.append(f"@r.{i}\n{seq}\n+\n{q}")
    return "\n".join(recs).encode()


def test_hot_contexts_rescale_inside_the_bucket(emu, oracle, capfd, monkeypatch):
    """Contexts with 40-60 k symbols (several Rescale() calls each) stay on the bucketed path: in buckets that large k_model codes the
    windows in which a row could reach its rescale point one element at a time, the lane whose row is due halves it first (md_rescale);
    4-, 8- and 16-symbol rows, one and two radix-4 levels, with and without the pair level."""
    monkeypatch.setenv("DSRC_GPU_DEBUG", "1")
    data = _hot(300)
    check(emu, oracle, data, [(1, 1, False), (3, 2, False), (2, 1, True)])
    err = capfd.readouterr().err
    assert err.count("2 of 2 streams tried, 0 handed back") == 3, err
    # 32-, 64- and 128-symbol alphabets whose hot context rescales (rows of two and three radix-4 levels, with and without the pair level)
    for n_vals in (20, 40, 90):
        rng = random.Random(n_vals)
        vals = list(range(2, 2 + n_vals))
        recs = []
        for i in range(260):
            q = bytes(33 + (vals[rng.randrange(n_vals)] if rng.random() < 0.03 else vals[-1]) for _ in range(200))
            recs.append(b"@q.%d\n" % i + bytes(rng.choice(b"ACGT") for _ in range(200)) + b"\n+\n" + q)
        check(emu, oracle, b"\n".join(recs), [(2, 2, False)] if n_vals != 40 else [(1, 1, False)])


def test_bucket_too_large_for_a_wave(emu, oracle, capfd, monkeypatch):
    """A bucket beyond the limit a wave may walk (BK_LIMIT; lowered here): the wave that finds it (k_model adds up its tiles' counts) hands the stream back."""
    monkeypatch.setenv("DSRC_GPU_DEBUG", "1")
    monkeypatch.setenv("DSRC_GPU_BUCKET_LIMIT", "16384")
    check(emu, oracle, _hot(300), [(1, 1, False)])
    assert "2 of 2 streams tried, 1 handed back" in capfd.readouterr().err       # the DNA stream (64 contexts, one of them with most symbols)


@pytest.mark.parametrize("env", [{"DSRC_GPU_BUCKETS": "0"}, {"DSRC_GPU_BUCKETS_BINNED": "0"}, {"DSRC_GPU_BUCKETS_MIN": "0"}])
def test_switches(emu, oracle, env, monkeypatch):
    """The path off; records scattered to stream order by k_model itself (no k_place); and the path on for streams of any length (tiny blocks: most buckets empty)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    # (the library reads the switches per batch)
    data = synth.illumina_fastq(260)[:-1]
    check(emu, oracle, data, [(3, 2, False), (2, 1, True)])
    tiny = synth.illumina_fastq(30)[:-1]
    check(emu, oracle, tiny, [(3, 2, False)])


def test_verify_decodes_both_chains_at_once(emu, oracle, monkeypatch):
    """verify_after_compress with the wave decoders: the compressing pass knows where every block's DNA stream starts and how many
    symbols it holds, so k_dec_dnarc runs next to k_dec_qrc instead of behind it (DecHint); DSRC_GPU_VERIFY_SERIAL=1 keeps the order
    archives need.  Same verdicts either way, 4- and 8-symbol DNA models, blocks without a DNA stream among the others."""
    good = synth.illumina_fastq(50)[:-1]
    iup = alphabet_fastq(20, n_rec=50, L=60, iupac=True)
    no_dna = b"\n".join(b"@n.%d\nNNNNNNNN\n+\n########" % i for i in range(30))
    for serial in (False, True):
        if serial:
            monkeypatch.setenv("DSRC_GPU_VERIFY_SERIAL", "1")
        for d, q in ((3, 2), (1, 1)):
            cfg = Config.from_levels(d, q, False, True)
            h = emu.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, True, verify=True)
            got = h.compress_batch([good, iup, no_dna, good])
            h.close()
            assert got[1][0] == oracle.compress_block(cfg, iup)[0]
