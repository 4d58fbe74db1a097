#!/bin/bash
# Round 3: fuzz soaks of the final library (GPU path against the oracle), fresh seeds.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r03_soak.txt; : > $out
timeout 420 python tools/fuzz_soak.py 930000 300 decode 2>&1 | tail -3 >> $out
timeout 300 python tools/fuzz_soak.py 940000 180 batch 2>&1 | tail -3 >> $out
timeout 240 python tools/fuzz_soak.py 950000 120 records 2>&1 | tail -3 >> $out
timeout 240 python tools/fuzz_soak.py 960000 120 solid 2>&1 | tail -3 >> $out
cat $out
