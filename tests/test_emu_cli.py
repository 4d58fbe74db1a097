"""The C++ host pipelines (dsrc_amd/csrc/host: DsrcCompressorGPU, DsrcDecompressorGPU behind the dsrc-amd CLI) on the CPU,
linked against the HIP emulator build of the kernels (tests/emu/dsrc-amd-emu): archive identity with the reference's
`dsrc c -t1` golden md5 on data whose blocks depend on block-to-block state -- with several scheduler instances spread
over a device list -- and decompression back to the input.  Test harness only; the product links libdsrc_gpu.so."""
import hashlib
import json
import os
import subprocess

import pytest

from dsrc_amd import synth
from tests.cases import state_dependent_fastq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_CLI = os.path.join(ROOT, "tests", "emu", "dsrc-amd-emu")
G = json.load(open(os.path.join(ROOT, "tests", "golden", "state_golden.json")))


@pytest.fixture(autouse=True)
def one_lane_quality_decoder(monkeypatch):
    monkeypatch.setenv("DSRC_GPU_DEC_SERIAL", "1")     # the wave-cooperative decoder is slow on the emulator (tests/test_emu_decode.py)


@pytest.fixture(scope="module")
def cli():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return EMU_CLI


def md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def test_sharded_instances_write_the_t1_archive(cli, tmp_path):
    a = G["small"]                       # three 1 MiB chunks whose blocks depend on the carried state
    data = state_dependent_fastq(a["n_per_region"])
    assert hashlib.sha256(data).hexdigest() == a["in_sha256"]
    src = tmp_path / "state.fastq"; src.write_bytes(data)
    arc = tmp_path / "s.dsrc"; back = tmp_path / "back.fastq"
    # two instances on each of two entries of the device list, one chunk per batch: every block on another instance
    subprocess.check_call([cli, "c", *a["flags"], "-b1", "-n1", "-t2", "-g0,0", str(src), str(arc)])
    assert (os.path.getsize(arc), md5(arc)) == (a["size"], a["md5"])
    # (decompression of this archive over a device list: tests/test_gpu_host_cli.py; on the emulator a small one below)


@pytest.mark.parametrize("flags", [["-d1", "-q1", "-c"], ["-d2", "-q1", "-l"]])
def test_round_trip_and_errors(cli, tmp_path, oracle, flags):
    data = synth.illumina_fastq(500)
    src = tmp_path / "a.fastq"; src.write_bytes(data)
    arc = tmp_path / "a.dsrc"; back = tmp_path / "a.out"; ref_arc = tmp_path / "o.dsrc"
    subprocess.check_call([cli, "c", *flags, str(src), str(arc)])
    d = int(flags[0][2:]); q = int(flags[1][2:])
    assert oracle.compress_file(str(src), str(ref_arc), d, q, "-l" in flags, "-c" in flags, 0, 8) == 0
    assert md5(arc) == md5(ref_arc)
    out = subprocess.run([cli, "d", "-s", "-t2", "-g0,0", "-n1", str(arc)], capture_output=True, check=True).stdout
    if "-l" not in flags:
        assert out == data
    else:
        assert out.split(b"\n")[0::4] == data.split(b"\n")[0::4]
    # a truncated archive and a non-archive are refused with the reference's messages, and no output is left behind
    bad = tmp_path / "bad.dsrc"; bad.write_bytes(arc.read_bytes()[:100])
    r = subprocess.run([cli, "d", str(bad), str(back)], capture_output=True)
    assert r.returncode != 0 and b"Corrupted DSRC archive" in r.stderr and not back.exists()
    bad.write_bytes(b"@not an archive\nACGT\n+\nIIII\n" * 4)
    r = subprocess.run([cli, "d", str(bad), str(back)], capture_output=True)
    assert r.returncode != 0 and b"Invalid archive" in r.stderr
    r = subprocess.run([cli, "c", "-d7", str(src), str(arc)], capture_output=True)
    assert r.returncode != 0 and b"invalid DNA compression mode" in r.stderr


def test_direct_io_reader_writes_the_same_archive(cli, tmp_path):
    """DSRC_HOST_DIRECT_IO=1: the batches are read with O_DIRECT (whole sectors around every chunk, the chunk in place behind the
    sector's head); chunks that start and end inside sectors, the end of the file inside a sector.  Where the file system refuses
    O_DIRECT (tmpfs) the switch falls back to ordinary reads -- the archive is the same either way."""
    a = G["small"]
    data = state_dependent_fastq(a["n_per_region"])
    got = {}
    for where in (tmp_path, os.path.join(ROOT, "tests", "emu")):      # pytest's tmp (often tmpfs) and the repository's file system
        src = os.path.join(str(where), "direct_io_state.fastq"); arc = os.path.join(str(where), "direct_io_state.dsrc")
        try:
            with open(src, "wb") as f:
                f.write(data)
            r = subprocess.run([cli, "c", *a["flags"], "-b1", "-n1", "-t2", src, arc], env=dict(os.environ, DSRC_HOST_DIRECT_IO="1", DSRC_HOST_TRACE="1"),
                               capture_output=True, check=True)
            got[str(where)] = b"direct reads: on" in r.stderr
            assert (os.path.getsize(arc), md5(arc)) == (a["size"], a["md5"])
        finally:
            for p in (src, arc):
                if os.path.exists(p):
                    os.unlink(p)
    assert len(got) == 2


def test_decompress_into_a_pipe_and_a_crlf_archive(cli, tmp_path):
    """Outputs that are not regular files (stdout by name -- /dev/stdout, /proc/self/fd/1 --, a named pipe) get their bytes in batch order through the in-turn
    fwrite path -- they can be neither sized nor mapped nor written at positions (the reference fwrites in order).  An archive of
    CRLF text decodes shorter than its blocks declare: the mapped attempt is abandoned and the same handles decode it again
    through the buffered path (one `instance` line per worker in the trace, not two)."""
    data = synth.illumina_fastq(300)
    src = tmp_path / "a.fastq"; src.write_bytes(data)
    arc = tmp_path / "a.dsrc"
    subprocess.check_call([cli, "c", "-d1", "-q1", "-b1", str(src), str(arc)])
    r = subprocess.run("%s d -t2 -n1 %s /proc/self/fd/1 | cat" % (cli, arc), shell=True, capture_output=True, check=True)
    assert r.stdout == data
    fifo = tmp_path / "out.fifo"; os.mkfifo(fifo)
    reader = subprocess.Popen(["cat", str(fifo)], stdout=subprocess.PIPE)
    subprocess.check_call([cli, "d", "-t2", "-n1", str(arc), str(fifo)])
    assert reader.communicate(timeout=60)[0] == data
    # CRLF input: same records, line ends dropped by the block format
    crlf = tmp_path / "c.fastq"; crlf.write_bytes(synth.illumina_fastq(300, crlf=True))
    arc2 = tmp_path / "c.dsrc"; back = tmp_path / "c.out"
    subprocess.check_call([cli, "c", "-d1", "-q1", "-b1", str(crlf), str(arc2)])
    r = subprocess.run([cli, "d", "-t2", "-n1", str(arc2), str(back)], env=dict(os.environ, DSRC_HOST_TRACE="1"), capture_output=True, check=True)
    assert back.read_bytes() == data
    assert r.stderr.count(b"instance") <= 2
