#!/usr/bin/env python3
"""Regenerates tests/golden/solid_golden.json: blocks and whole archives of SOLiD colour-space input compressed by the
UNMODIFIED reference (oracle/_ref).  Run in the build container only:

    python tests/golden/make_solid_golden.py

Data only: generator specs (tests/cases.py::fuzz_solid) and the reference's output digests."""
import dataclasses
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests._oracle import Config, Ref, Oracle, REF_BIN    # noqa: E402
from tests.cases import LEVELS, fuzz_solid                # noqa: E402


def sha(b):
    return hashlib.sha256(b).hexdigest()


def main():
    r = Ref(); o = Oracle()
    g = {"blocks": [], "archives": []}
    for seed in range(24):
        nrec = None if seed % 3 else 50
        data, desc = fuzz_solid(seed, nrec)
        for d, q, lossy, crc in LEVELS:
            cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), color_space=True)
            try:
                o.compress_block(cfg, data)          # only to skip inputs that are undefined behaviour in the reference
            except RuntimeError:
                continue
            blk, raw, comp = r.compress_block(cfg, data)
            g["blocks"].append({"seed": seed, "nrec": nrec, "in_sha256": sha(data), "levels": [d, q, lossy, crc],
                                "raw": raw, "comp": comp, "size": len(blk), "sha256": sha(blk)})
    with tempfile.TemporaryDirectory() as td:
        for seed, nrec in ((8, 30000), (21, 12000), (3, 20000)):
            data, desc = fuzz_solid(seed, nrec)
            p = os.path.join(td, "in.fastq"); open(p, "wb").write(data + b"\n")
            for flags in (["-d0", "-q0"], ["-d2", "-q2"], ["-d1", "-q2", "-l"], ["-d3", "-q1"]):
                dst = os.path.join(td, "o.dsrc")
                subprocess.check_call([REF_BIN, "c"] + flags + ["-b1", "-t1", p, dst], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                arc = open(dst, "rb").read()
                g["archives"].append({"seed": seed, "nrec": nrec, "in_sha256": sha(data + b"\n"), "flags": flags, "buf_mb": 1,
                                      "size": len(arc), "sha256": sha(arc)})
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "solid_golden.json"), "w") as f:
        json.dump(g, f, indent=0, separators=(",", ":"))
    print("blocks", len(g["blocks"]), "archives", len(g["archives"]))


if __name__ == "__main__":
    main()
