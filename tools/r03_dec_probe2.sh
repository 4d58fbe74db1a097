#!/bin/bash
# decode-only passes at several batch sizes with kernel statistics (no pytest)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for B in ${DEC_SIZES:-64 2400 4500}; do
  rm -rf gpurun_out/prof_d
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_d -- python tools/decode_bench.py --blocks $B --distinct ${DEC_DISTINCT:-300} --passes 2 ${DEC_ARGS} > gpurun_out/r03_dec_b$B.json 2> gpurun_out/r03_dec_b$B.err
  F=$(find gpurun_out/prof_d -name "*.db" | head -1); [ -z "$F" ] && F=$(find gpurun_out/prof_d -name "*kernel_trace.csv" | head -1)
  python tools/prof_summary.py "$F" gpurun_out/r03_dec_kernels_b$B.txt > /dev/null
  rm -rf gpurun_out/prof_d
done
for B in ${DEC_SIZES:-64 2400 4500}; do cat gpurun_out/r03_dec_b$B.json; grep -v "^W2026\|^E2026" gpurun_out/r03_dec_b$B.err | tail -3; grep "k_dec" gpurun_out/r03_dec_kernels_b$B.txt | cut -c1-130; done
