"""SOLiD colour space against golden vectors of the unmodified reference
(tests/golden/solid_golden.json, made by tests/golden/make_solid_golden.py).
CPU: the oracle (blocks and whole files).  GPU: blocks through the C ABI, whole archives through the dsrc-amd CLI,
whose first-chunk analysis must detect the colour space by itself (FastqParser::Analyze, src/FastqParser.cpp:78-105)."""
import dataclasses
import hashlib
import json
import os
import subprocess

import pytest

from tests._oracle import Config
from tests.cases import fuzz_solid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = json.load(open(os.path.join(ROOT, "tests", "golden", "solid_golden.json")))
CLI = os.path.join(ROOT, "dsrc_amd", "csrc", "dsrc-amd")


def sha(b):
    return hashlib.sha256(b).hexdigest()


_cache = {}


def chunk(seed, nrec):
    if (seed, nrec) not in _cache:
        _cache[(seed, nrec)] = fuzz_solid(seed, nrec)[0]
    return _cache[(seed, nrec)]


def cfg_of(e):
    d, q, lossy, crc = e["levels"]
    return dataclasses.replace(Config.from_levels(d, q, lossy, crc), color_space=True)


def test_oracle_blocks(oracle):
    assert len(G["blocks"]) > 200
    for e in G["blocks"]:
        data = chunk(e["seed"], e["nrec"])
        assert sha(data) == e["in_sha256"], "generator drifted from the golden input"
        blk, raw, comp = oracle.compress_block(cfg_of(e), data)
        assert (len(blk), sha(blk), raw, comp) == (e["size"], e["sha256"], e["raw"], e["comp"]), e
    # both kinds of block occur: constant primer (FLAG_DELTA_CONSTANT, records shortened) and varying primer
    flags = set()
    for seed in range(24):
        try:
            blk = oracle.compress_block(dataclasses.replace(Config.from_levels(0, 0), color_space=True), chunk(seed, None if seed % 3 else 50))[0]
        except RuntimeError:
            continue                                  # undefined in the reference (e.g. two-character reads), not in the golden set
        flags.add(blk[11] & 1)
    assert flags == {0, 1}


def test_oracle_archives(oracle, tmp_path):
    for e in G["archives"]:
        data = chunk(e["seed"], e["nrec"]) + b"\n"
        assert sha(data) == e["in_sha256"]
        src = tmp_path / "in.fastq"; src.write_bytes(data)
        dst = str(tmp_path / "o.dsrc")
        f = e["flags"]
        assert oracle.compress_file(str(src), dst, int(f[0][2]), int(f[1][2]), "-l" in f, "-c" in f, 0, e["buf_mb"]) == 0
        arc = open(dst, "rb").read()
        assert (len(arc), sha(arc)) == (e["size"], e["sha256"]), e


@pytest.mark.gpu
def test_gpu_blocks():
    from dsrc_amd._lib import Handle
    by_cfg = {}
    for e in G["blocks"]:
        by_cfg.setdefault(tuple(e["levels"]), []).append(e)
    for levels, es in by_cfg.items():
        cfg = cfg_of(es[0])
        h = Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset, color_space=True)
        got = h.compress_batch([chunk(e["seed"], e["nrec"]) for e in es])
        h.close()
        for e, (blk, raw, comp) in zip(es, got):
            assert (len(blk), sha(blk), raw, comp) == (e["size"], e["sha256"], e["raw"], e["comp"]), e


@pytest.mark.gpu
def test_gpu_cli_archives(tmp_path):
    assert os.path.exists(CLI), "dsrc-amd not built"
    for e in G["archives"]:
        src = tmp_path / "in.fastq"; src.write_bytes(chunk(e["seed"], e["nrec"]) + b"\n")
        dst = str(tmp_path / "o.dsrc")
        subprocess.check_call([CLI, "c"] + e["flags"] + ["-b%d" % e["buf_mb"], "-t3", str(src), dst], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        arc = open(dst, "rb").read()
        assert (len(arc), sha(arc)) == (e["size"], e["sha256"]), e
