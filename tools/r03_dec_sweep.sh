#!/bin/bash
# Round 3: pass size x instances sweep of the decoder at -d3 -q2 (the quality stage's time steps with waves per SIMD: 1024 SIMDs).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r03_dec_sweep.txt; : > $out
for cfg in "1024 1" "2048 1" "3072 1" "4096 1" "2048 2" "3072 2" "4096 2" "3600 2" "2048 3" "3072 3"; do
  set -- $cfg
  st=$(python -c "print(round(0.55*$1/1024,2))")
  echo "== blocks $1 inst $2 stagger $st" >> $out
  timeout 600 python tools/decode_bench.py --blocks $1 --distinct 300 -d 3 -q 2 --passes 2 --inst $2 --stagger $st --check 1 2>&1 | grep -E "pass\"|instances|Error|error" >> $out
done
cat $out
