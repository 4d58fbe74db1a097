#!/usr/bin/env python3
"""Regenerates tests/golden/decode_golden.json: for every block vector of golden.json and solid_golden.json, what the
UNMODIFIED reference's BlockCompressor::Read (oracle/_ref, ref_decompress_block) makes of the block -- digest and size
of the decoded chunk text.  Run in the build container only:

    python tests/golden/make_decode_golden.py

Data only.  Blocks are not stored: they are re-made from the generator specs by the oracle's encoder, which the golden
block digests pin.  Blocks the reference cannot decode deterministically (its decoder runs off the end of the block or
indexes a table with a character, see oracle/dsrc_oracle_dec.c) are listed as refused."""
import dataclasses
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests._oracle import Config, Ref, Oracle                # noqa: E402
from tests.cases import fuzz_solid                           # noqa: E402
from tests.test_oracle_golden import G as GOLD, get_input    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sha(b):
    return hashlib.sha256(b).hexdigest()


def main():
    r = Ref(); o = Oracle()
    out = {"blocks": [], "solid": []}

    def one(cfg, data, want_sha):
        blk = o.compress_block(cfg, data)[0]
        assert sha(blk) == want_sha
        cap = 2 * len(data) + 4096
        try:
            o.decompress_block(cfg, blk, cap)
        except RuntimeError:
            return {"refused": True}                      # never handed to the reference: it would read stale memory
        text = r.decompress_block(cfg, blk, cap)
        return {"text_size": len(text), "text_sha256": sha(text)}

    for e in GOLD["blocks"]:
        if e.get("ref_ub"):
            continue                                      # no block exists: the encoder side is undefined
        d, q, lossy, crc = e["levels"]
        cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), tag_flags=e.get("tag_flags", 0))
        res = one(cfg, get_input(e), e["sha256"])
        out["blocks"].append({"name": e["name"], "levels": e["levels"], "tag_flags": e.get("tag_flags", 0), **res})
    S = json.load(open(os.path.join(HERE, "solid_golden.json")))
    for e in S["blocks"]:
        d, q, lossy, crc = e["levels"]
        cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), color_space=True)
        res = one(cfg, fuzz_solid(e["seed"], e["nrec"])[0], e["sha256"])
        out["solid"].append({"seed": e["seed"], "nrec": e["nrec"], "levels": e["levels"], **res})
    with open(os.path.join(HERE, "decode_golden.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    for k in out:
        print(k, len(out[k]), "refused", sum(1 for x in out[k] if x.get("refused")))


if __name__ == "__main__":
    main()
