#!/bin/bash
# Round 3: fuzz soaks of the final library (GPU path against the oracle), fresh seeds.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r03_soak2.txt; : > $out
timeout 600 python tools/fuzz_soak.py ${SOAK_SEED:-1930000} 480 decode 2>&1 | tail -2 >> $out
timeout 420 python tools/fuzz_soak.py $((${SOAK_SEED:-1930000}+10000)) 300 batch 2>&1 | tail -2 >> $out
timeout 300 python tools/fuzz_soak.py $((${SOAK_SEED:-1930000}+20000)) 180 records 2>&1 | tail -2 >> $out
timeout 300 python tools/fuzz_soak.py $((${SOAK_SEED:-1930000}+30000)) 180 solid 2>&1 | tail -2 >> $out
cat $out
