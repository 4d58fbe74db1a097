"""The drop-in boundary proven against the REAL reference (test-only): oracle/_ref/dsrc_ref_gpu is the unmodified reference's
chunk reader, queues, pools, ordered writer / archive reader and FASTQ writer with its CPU workers replaced by the one
worker INTEGRATION.md section 1 describes (oracle/ref_gpu_main.cpp), bound to the C ABI through dsrcgpu_submit / flush /
try_collect / collect / release and dsrcgpu_decompress_batch.  Its archives must be the ones the reference itself writes
with `dsrc c -t1` (reference src/DsrcWorker.cpp:30-104, src/DsrcOperator.cpp:230-521).
CPU: the same binary linked against the HIP-emulator build of the kernels.  GPU: linked against libdsrc_gpu.so."""
import hashlib
import os
import subprocess

import pytest

from tests.conftest import need_built

from dsrc_amd import synth
from tests._oracle import REF_BIN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_GPU = os.path.join(ROOT, "oracle", "_ref", "dsrc_ref_gpu")
REF_GPU_EMU = os.path.join(ROOT, "oracle", "_ref", "dsrc_ref_gpu_emu")


def md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def _check(binary, tmp_path, data, levels, batch, env=None):
    src = tmp_path / "in.fastq"; src.write_bytes(data)
    for flags in levels:
        ref = tmp_path / "ref.dsrc"; ours = tmp_path / "ours.dsrc"; back = tmp_path / "back.fastq"; refback = tmp_path / "refback.fastq"
        subprocess.check_call([REF_BIN, "c", *flags, "-b1", "-t1", str(src), str(ref)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call([binary, "c", *flags, "-b1", "-n%d" % batch, str(src), str(ours)], env=env)
        assert md5(ours) == md5(ref), flags
        subprocess.check_call([binary, "d", "-n%d" % batch, str(ref), str(back)], env=env)
        subprocess.check_call([REF_BIN, "d", "-t1", str(ref), str(refback)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert md5(back) == md5(refback), flags
        if "-l" not in flags:
            assert back.read_bytes() == data


def test_reference_pipeline_with_the_gpu_worker_on_the_emulator(tmp_path):
    need_built(REF_GPU_EMU, "oracle/_ref/dsrc_ref_gpu_emu"); need_built(REF_BIN, "oracle/_ref/dsrc_ref")
    data = synth.illumina_fastq(250)
    env = dict(os.environ, DSRC_GPU_DEC_SERIAL="1")         # the wave-cooperative decoder is slow on the emulator (tests/test_emu_decode.py)
    _check(REF_GPU_EMU, tmp_path, data, [["-d0", "-q0"], ["-d3", "-q2", "-c"], ["-d2", "-q1", "-l"]], batch=2, env=env)


@pytest.mark.gpu
def test_reference_pipeline_with_the_gpu_worker(tmp_path):
    need_built(REF_GPU, "oracle/_ref/dsrc_ref_gpu"); need_built(REF_BIN, "oracle/_ref/dsrc_ref")
    from tests.cases import state_dependent_fastq
    _check(REF_GPU, tmp_path, synth.illumina_fastq(60000), [["-d0", "-q0"], ["-d3", "-q2"], ["-d1", "-q1", "-c"], ["-d2", "-q1", "-l"]], batch=7)
    # blocks that depend on the state carried from block to block, one chunk per batch, three batches in flight
    _check(REF_GPU, tmp_path, state_dependent_fastq(), [["-d0", "-q0"], ["-d1", "-q1", "-c"]], batch=1)
    _check(REF_GPU, tmp_path, synth.iontorrent_fastq(9000), [["-d2", "-q1", "-l"], ["-d0", "-q1"]], batch=3)
