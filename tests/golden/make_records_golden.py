#!/usr/bin/env python3
"""Regenerates tests/golden/records_golden.json: archives written by the UNMODIFIED reference's record-level API
(wrap::FastqFile -> wrap::DsrcArchive::WriteNextRecord), driven by oracle/ref_records.cpp -> oracle/_ref/ref_records.
Run in the build container only:

    python tests/golden/make_records_golden.py

Data only: generator specs and the digests / block tables of the reference's archives."""
import hashlib
import json
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from dsrc_amd import synth                       # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_records")

FILES = {
    "illumina9000": lambda: synth.illumina_fastq(9000),
    "illumina3200": lambda: synth.illumina_fastq(3200, first=501),
    "iontorrent6000": lambda: synth.iontorrent_fastq(6000),
}
# (dna level, quality level, lossy).  Lossless quality levels 1-2 become qualityOrder 3 / 6 in this API (src/DsrcArchive.cpp:42: level * 3,
# whatever `lossy` says): the lossless proxy treats every order but 1 as order 2 without the "F" schemes (src/QualityModelerProxy.h:
# 225-283), writes that order into the footer, and the reference's DsrcArchive reads such archives back -- checked below.
LEVELS = [(0, 0, 0), (1, 0, 0), (3, 0, 0), (2, 1, 1), (3, 2, 1), (0, 2, 1), (2, 1, 0), (0, 2, 0), (3, 2, 0)]


def block_table(arc: bytes):
    foot = struct.unpack(">Q", arc[8:16])[0]
    n = struct.unpack(">Q", arc[24:32])[0]
    return list(struct.unpack("<%dI" % n, arc[foot + 1: foot + 1 + 4 * n]))


def main():
    g = {"archives": []}
    with tempfile.TemporaryDirectory() as td:
        for name, gen in FILES.items():
            data = gen()
            p = os.path.join(td, name + ".fastq")
            open(p, "wb").write(data)
            for d, q, lossy in LEVELS:
                if name.startswith("ion") and d > 0 and not lossy:
                    continue         # > 8 DNA symbols with an order model: undefined in the reference
                for buf in ((1, 2) if name == "illumina9000" else (1,)):
                    dst = os.path.join(td, "o.dsrc")
                    subprocess.check_call([REF, p, dst, str(d), str(q), str(lossy), str(buf), "33"], stderr=subprocess.DEVNULL)
                    arc = open(dst, "rb").read()
                    if q and not lossy:          # the reference reads its own archive back (DsrcArchive::ReadNextRecord): the level is defined by what it does
                        back = os.path.join(td, "back.fastq")
                        subprocess.check_call([REF, "-x", dst, back], stderr=subprocess.DEVNULL)
                        assert open(back, "rb").read() == data, (name, d, q)
                    g["archives"].append({"name": name, "in_sha256": hashlib.sha256(data).hexdigest(), "levels": [d, q, lossy], "buf_mb": buf,
                                          "quality_offset": 33, "size": len(arc), "sha256": hashlib.sha256(arc).hexdigest(),
                                          "block_sizes": block_table(arc)})
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "records_golden.json"), "w") as f:
        json.dump(g, f, indent=0, separators=(",", ":"))
    print("archives", len(g["archives"]))


if __name__ == "__main__":
    main()
