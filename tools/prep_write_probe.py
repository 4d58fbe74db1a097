"""Which characters does the round-2 form of k_prep_write get wrong?  Compress a synthetic chunk with the library named by
DSRC_GPU_LIB (a -DDSRC_PREP_WRITE_IN_IF=1 build, tools/r03_prep_write_bisect.sh), decode the block again and list where the text
differs from the input: record, position in the read, base, quality got / wanted."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrc_amd import _lib, synth
data = synth.illumina_fastq(20000)[:-1]
h = _lib.Handle(0, 0)
blk = h.compress_block(data)[0]
txt = h.decompress_batch([blk])[0]
h.close()
a = data.split(b"\n"); b = txt.split(b"\n")
bad = 0; hist = {}
for r in range(0, min(len(a), len(b)) // 4):
    qa, qb, sa, sb = a[4 * r + 3], b[4 * r + 3], a[4 * r + 1], b[4 * r + 1]
    for j in range(min(len(qa), len(qb))):
        if qa[j] != qb[j] or sa[j] != sb[j]:
            if bad < 12:
                print(f"record {r} pos {j} (lane {j % 64}): base {chr(sa[j])}->{chr(sb[j])} quality {qa[j] - 33}->{qb[j] - 33}   neighbours bases {sa[max(0, j - 3): j + 4]} quals {[x - 33 for x in qa[max(0, j - 3): j + 4]]}")
            bad += 1; hist[j] = hist.get(j, 0) + 1
print("mismatching characters:", bad, "positions:", sorted(hist.items())[:40])
