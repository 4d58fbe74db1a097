"""Compress one synthetic 20 k-read chunk at -d0 -q0 on the GPU and print where the block differs from the oracle's
(section sizes, first differing byte).  Written while chasing the k_prep_write discrepancy of round 2 (NOTES/rounds_1_to_4.md section 10):
build dsrc_amd/csrc with -DFAST_WRITE=true to reproduce it.  Usage: python tools/block_diff.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrc_amd import _lib, synth
from tests._oracle import Config, Oracle
o = Oracle()
data = synth.illumina_fastq(20000)[:-1]
cfg = Config.from_levels(0, 0, False, False)
want = o.compress_block(cfg, data)
h = _lib.Handle(cfg.dna_order, cfg.quality_order, cfg.lossy, cfg.crc, cfg.quality_offset)
got = h.compress_block(data)
print("sizes", len(got[0]), len(want[0]), got[1:], want[1:])
a, b = got[0], want[0]
for i in range(min(len(a), len(b))):
    if a[i] != b[i]:
        print("first diff at", i, a[i-4:i+12].hex(), b[i-4:i+12].hex()); break
st = o.block_stats(cfg, data)
print("oracle stats d", list(st[0])[:21] if hasattr(st[0], '__iter__') else st)
