#!/bin/bash
# Round 3: DNA stage against the number of blocks in flight (-d3 -q0: the quality stage is the per-position Huffman decoder, short)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r03_dna_scale.txt; : > $out
for B in 64 128 256 512 1024 2400; do
  D=$B; [ $D -gt 300 ] && D=300
  echo "== blocks $B" >> $out
  timeout 300 python tools/decode_bench.py --blocks $B --distinct $D -d 3 -q 0 --passes 2 --check 1 2>&1 | grep '"pass": 1' >> $out
done
cat $out
