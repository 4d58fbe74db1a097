"""Decode golden cases shared by the CPU (oracle) and GPU tests: tests/golden/decode_golden.json holds, for every
block vector of golden.json / solid_golden.json, the digest of what the unmodified reference's BlockCompressor::Read
makes of the block (tests/golden/make_decode_golden.py).  Blocks are re-made from the generator specs with the
oracle's encoder, which the golden block digests pin."""
from __future__ import annotations

import dataclasses
import hashlib
import json
import os

from tests._oracle import Config
from tests.cases import fuzz_solid

HERE = os.path.dirname(os.path.abspath(__file__))
D = json.load(open(os.path.join(HERE, "golden", "decode_golden.json")))


def sha(b):
    return hashlib.sha256(b).hexdigest()


def cases(oracle, solid: bool = False, stride: int = 1):
    """Yields (label, cfg, block, cap, expected-entry)."""
    from tests.test_oracle_golden import G, get_input
    if not solid:
        by_key = {(e["name"], tuple(e["levels"]), e.get("tag_flags", 0)): e for e in G["blocks"] if not e.get("ref_ub")}
        for i, x in enumerate(D["blocks"]):
            if i % stride:
                continue
            e = by_key[(x["name"], tuple(x["levels"]), x["tag_flags"])]
            d, q, lossy, crc = x["levels"]
            cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), tag_flags=x["tag_flags"])
            data = get_input(e)
            blk = oracle.compress_block(cfg, data)[0]
            assert sha(blk) == e["sha256"]
            yield (x["name"], x["levels"], x["tag_flags"]), cfg, blk, 2 * len(data) + 4096, x
    else:
        cache = {}
        for i, x in enumerate(D["solid"]):
            if i % stride:
                continue
            key = (x["seed"], x["nrec"])
            if key not in cache:
                cache[key] = fuzz_solid(*key)[0]
            d, q, lossy, crc = x["levels"]
            cfg = dataclasses.replace(Config.from_levels(d, q, lossy, crc), color_space=True)
            blk = oracle.compress_block(cfg, cache[key])[0]
            yield ("solid", x["seed"], x["levels"]), cfg, blk, 2 * len(cache[key]) + 4096, x
