#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r03_dec_sweep2.txt; : > $out
for cfg in "3600 2 1.9" "4096 2 2.2" "2400 3 1.5" "3072 3 1.8" "2048 4 1.2" "4800 2 2.5"; do
  set -- $cfg
  echo "== blocks $1 inst $2 stagger $3" >> $out
  timeout 600 python tools/decode_bench.py --blocks $1 --distinct 300 -d 3 -q 2 --passes 2 --inst $2 --stagger $3 --check 1 2>&1 | grep -E "instances|Error|error" >> $out
done
cat $out
