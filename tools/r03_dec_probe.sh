#!/bin/bash
# Round 3, decoder: GPU parity tests of the decoder, then decode-only passes at several batch sizes with kernel statistics.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r03_dec_pytest.txt
for B in ${DEC_SIZES:-64 600 2400 4500}; do
  rm -rf gpurun_out/prof_d
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_d -- python tools/decode_bench.py --blocks $B --distinct ${DEC_DISTINCT:-300} --passes 2 > gpurun_out/r03_dec_b$B.json 2> gpurun_out/r03_dec_b$B.err
  F=$(find gpurun_out/prof_d -name "*.db" | head -1); [ -z "$F" ] && F=$(find gpurun_out/prof_d -name "*kernel_trace.csv" | head -1)
  python tools/prof_summary.py "$F" gpurun_out/r03_dec_kernels_b$B.txt > /dev/null
  rm -rf gpurun_out/prof_d
done
cat gpurun_out/r03_dec_pytest.txt
for B in ${DEC_SIZES:-64 600 2400 4500}; do cat gpurun_out/r03_dec_b$B.json; tail -3 gpurun_out/r03_dec_b$B.err; grep "k_dec" gpurun_out/r03_dec_kernels_b$B.txt | cut -c1-130; done
