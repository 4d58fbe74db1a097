#!/usr/bin/env python3
"""One-off soak: many more fuzz seeds than the test-suite runs, GPU path vs oracle, for a bounded time.
Usage: tools/fuzz_soak.py [first_seed] [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsrc_amd import _lib  # noqa: E402
from tests._oracle import Config, Oracle  # noqa: E402
from tests.cases import fuzz_fastq  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    limit = float(sys.argv[2]) if len(sys.argv) > 2 else 180.0
    o = Oracle()
    t0 = time.time(); n = 0; refused = 0
    cfgs = [(0, 0, False, False), (3, 2, False, True), (2, 1, True, False), (1, 1, False, False), (2, 2, False, False), (3, 2, True, False), (0, 2, False, False), (3, 0, False, False)]
    while time.time() - t0 < limit:
        nrec = [None, 3000, 7000][seed % 3]
        data, desc = fuzz_fastq(seed, nrec)
        for d, q, lossy, crc in cfgs:
            cfg = Config.from_levels(d, q, lossy, crc)
            h = _lib.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc)
            try:
                want = o.compress_block(cfg, data)
            except RuntimeError as e:
                assert "rc=-2" in str(e), e
                try:
                    h.compress_block(data)
                    raise AssertionError(f"seed {seed} {desc} -d{d} -q{q}: reference-UB input was not refused")
                except _lib.DsrcGpuError:
                    refused += 1
                h.close()
                continue
            got = h.compress_block(data)
            h.close()
            assert got == want, f"seed {seed} {desc} -d{d} -q{q} lossy={lossy} crc={crc}: GPU block differs from the oracle"
            n += 1
        seed += 1
    print(f"fuzz soak: {n} blocks identical, {refused} reference-UB inputs refused, seeds up to {seed - 1}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
