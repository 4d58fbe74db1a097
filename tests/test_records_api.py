"""Record-level archive API: wrap::DsrcArchive::StartCompress / WriteNextRecord / FinishCompress.

CPU part: the oracle's restatement of that path (orc_compress_records_file) against archives written by the unmodified
reference's DsrcArchive (tests/golden/records_golden.json, made by tests/golden/make_records_golden.py, and live against
oracle/_ref/ref_records when it is there); the C++ host's DsrcArchive linked against the HIP emulator against the oracle.
GPU part: dsrc-amd-records and pydsrc.DsrcArchive against the same golden digests."""
import hashlib
import json
import os
import subprocess

import pytest

from tests.conftest import need_built

from dsrc_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = json.load(open(os.path.join(ROOT, "tests", "golden", "records_golden.json")))
REF_RECORDS = os.path.join(ROOT, "oracle", "_ref", "ref_records")
EMU_HOST = os.path.join(ROOT, "tests", "emu", "dsrc-amd-records-emu")
GPU_HOST = os.path.join(ROOT, "dsrc_amd", "csrc", "dsrc-amd-records")

FILES = {
    "illumina9000": lambda: synth.illumina_fastq(9000),
    "illumina3200": lambda: synth.illumina_fastq(3200, first=501),
    "iontorrent6000": lambda: synth.iontorrent_fastq(6000),
}


def sha(b):
    return hashlib.sha256(b).hexdigest()


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("rec")
    paths = {}
    for name, gen in FILES.items():
        data = gen()
        for e in G["archives"]:
            if e["name"] == name:
                assert sha(data) == e["in_sha256"], "generator drifted from the golden input"
        p = d / (name + ".fastq"); p.write_bytes(data)
        paths[name] = str(p)
    return d, paths


def test_oracle_against_reference_archives(oracle, inputs):
    d, paths = inputs
    out = str(d / "orc.dsrc")
    for e in G["archives"]:
        dl, ql, lossy = e["levels"]
        assert oracle.compress_records_file(paths[e["name"]], out, dl, ql, bool(lossy), e["quality_offset"], e["buf_mb"]) == 0
        arc = open(out, "rb").read()
        assert (len(arc), sha(arc)) == (e["size"], e["sha256"]), e


def test_running_chunk_size_and_block_cut():
    """What makes these archives differ from `dsrc c`: blocks are cut by payload bytes and the chunkSize word is a
    running total (src/BlockCompressorExt.cpp:126, src/BlockCompressor.cpp:105-109)."""
    e = next(x for x in G["archives"] if x["name"] == "illumina9000" and x["buf_mb"] == 1 and x["levels"] == [0, 0, 0])
    assert len(e["block_sizes"]) == 4


def test_oracle_against_live_reference(oracle, inputs, tmp_path):
    need_built(REF_RECORDS, "oracle/_ref/ref_records")
    d, paths = inputs
    # plus repetition, offset 64 data and other buffer sizes than the golden set
    data = synth.illumina_fastq(5000, first=90001)
    lines = data.split(b"\n")
    for i in range(2, len(lines), 4):
        lines[i] = b"+" + lines[i - 2][1:]
    rep = tmp_path / "rep.fastq"; rep.write_bytes(b"\n".join(lines))
    cases = [(str(rep), 0, 0, 0, 1, 33, 1), (str(rep), 2, 2, 1, 1, 33, 1), (paths["illumina9000"], 3, 0, 0, 3, 33, 0),
             (paths["iontorrent6000"], 0, 0, 0, 1, 33, 0), (paths["iontorrent6000"], 1, 1, 1, 2, 33, 0)]
    for src, dl, ql, lossy, buf, off, prep in cases:
        a = str(tmp_path / "ref.dsrc"); b = str(tmp_path / "orc.dsrc")
        subprocess.check_call([REF_RECORDS, src, a, str(dl), str(ql), str(lossy), str(buf), str(off), str(prep)], stderr=subprocess.DEVNULL)
        assert oracle.compress_records_file(src, b, dl, ql, bool(lossy), off, buf, bool(prep)) == 0
        assert open(a, "rb").read() == open(b, "rb").read(), (src, dl, ql, lossy, buf, prep)


def test_host_archive_on_emulator(inputs):
    """dsrc_host.cpp's DsrcArchive (chunk assembly, running chunkSize, batches, archive writer) with the kernels on
    the CPU emulator: the reference's archive, two blocks."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")], stdout=subprocess.DEVNULL)
    d, paths = inputs
    e = next(x for x in G["archives"] if x["name"] == "illumina3200" and x["levels"] == [0, 0, 0])
    assert len(e["block_sizes"]) == 2
    out = str(d / "emu.dsrc")
    subprocess.check_call([EMU_HOST, paths["illumina3200"], out, "0", "0", "0", "1", "33"], stderr=subprocess.DEVNULL)
    arc = open(out, "rb").read()
    assert (len(arc), sha(arc)) == (e["size"], e["sha256"])
    # lossless quality level 1 = qualityOrder 3 in this API, as in the reference (src/DsrcArchive.cpp:42): the reference's archive
    e = next(x for x in G["archives"] if x["name"] == "illumina3200" and x["levels"] == [0, 2, 0])
    subprocess.check_call([EMU_HOST, paths["illumina3200"], out, "0", "2", "0", "1", "33"], stderr=subprocess.DEVNULL)
    arc = open(out, "rb").read()
    assert (len(arc), sha(arc)) == (e["size"], e["sha256"])
    # the offset must be given
    assert subprocess.run([EMU_HOST, paths["illumina3200"], out, "0", "0", "0", "1", "0"], capture_output=True).returncode == 1


def test_host_archive_read_side_on_emulator(inputs, oracle):
    """DsrcArchive::StartDecompress / ReadNextRecord / FinishDecompress (reference src/DsrcArchive.cpp:170-215) on the
    emulator: records come back from an archive written by the record-level API (chunkSize words are running totals)
    and from one written by `dsrc c` (own sizes), with the settings taken from the footer."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")], stdout=subprocess.DEVNULL)
    d, paths = inputs
    src = paths["illumina3200"]
    a = str(d / "rd_rec.dsrc"); b = str(d / "rd_file.dsrc"); back = str(d / "rd_back.fastq")
    assert oracle.compress_records_file(src, a, 0, 0, False, 33, 1) == 0
    assert oracle.compress_file(src, b, 1, 1, False, True, 0, 1) == 0
    env = dict(os.environ, DSRC_GPU_DEC_SERIAL="1")     # the wave-cooperative decoder is slow on the emulator (tests/test_emu_decode.py)
    for arc in (a, b):
        r = subprocess.run([EMU_HOST, "-x", arc, back], capture_output=True, env=env)
        assert r.returncode == 0, r.stderr
        assert b"records: 3200" in r.stderr
        assert open(back, "rb").read() == open(src, "rb").read()


@pytest.mark.gpu
def test_gpu_archive_read_side(inputs, oracle):
    """Every golden record-level archive (written by the reference's DsrcArchive) read back record by record on the GPU."""
    assert os.path.exists(GPU_HOST), "dsrc-amd-records not built"
    d, paths = inputs
    arc = str(d / "rr.dsrc"); back = str(d / "rr.fastq")
    for e in G["archives"]:
        dl, ql, lossy = e["levels"]
        assert oracle.compress_records_file(paths[e["name"]], arc, dl, ql, bool(lossy), e["quality_offset"], e["buf_mb"]) == 0
        assert sha(open(arc, "rb").read()) == e["sha256"]
        subprocess.check_call([GPU_HOST, "-x", arc, back], stderr=subprocess.DEVNULL)
        got = open(back, "rb").read(); src = open(paths[e["name"]], "rb").read()
        if not lossy:
            assert got == src, e
        else:
            assert got.split(b"\n")[0::4] == src.split(b"\n")[0::4] and len(got) == len(src)


@pytest.mark.gpu
def test_gpu_host_archives(inputs):
    assert os.path.exists(GPU_HOST), "dsrc-amd-records not built"
    d, paths = inputs
    out = str(d / "gpu.dsrc")
    for e in G["archives"]:
        dl, ql, lossy = e["levels"]
        subprocess.check_call([GPU_HOST, paths[e["name"]], out, str(dl), str(ql), str(lossy), str(e["buf_mb"]), str(e["quality_offset"])],
                              stderr=subprocess.DEVNULL)
        arc = open(out, "rb").read()
        assert (len(arc), sha(arc)) == (e["size"], e["sha256"]), e


@pytest.mark.gpu
def test_gpu_pydsrc_archive(inputs):
    """The reference's Python names (py/Interface.cpp:59-94): FastqFile -> DsrcArchive, as in examples/py."""
    from dsrc_amd import pydsrc
    d, paths = inputs
    e = next(x for x in G["archives"] if x["name"] == "illumina9000" and x["buf_mb"] == 1 and x["levels"] == [3, 2, 1])
    out = str(d / "py.dsrc")
    f = pydsrc.FastqFile(); f.Open(paths["illumina9000"])
    a = pydsrc.DsrcArchive()
    a.DNACompressionLevel = 3; a.QualityCompressionLevel = 2; a.LossyCompression = True
    a.FastqBufferSizeMB = 1; a.QualityOffset = 33
    a.StartCompress(out)
    rec = pydsrc.FastqRecord(); n = 0
    while f.ReadNextRecord(rec):
        a.WriteNextRecord(rec); n += 1
    a.FinishCompress(); f.Close()
    assert n == 9000
    arc = open(out, "rb").read()
    assert (len(arc), sha(arc)) == (e["size"], e["sha256"])
