#!/bin/bash
# Round 3: the whole GPU suite, the default bench line, a decode soak.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r03_gpu_suite.txt
cat gpurun_out/r03_gpu_suite.txt
timeout 1500 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
tail -c 1500 gpurun_out/r03_bench_default.json
tail -5 gpurun_out/r03_bench_default.err
( timeout 300 python tools/fuzz_soak.py 970000 200 decode 2>&1 | tail -2 ) > gpurun_out/r03_soak_final.txt
cat gpurun_out/r03_soak_final.txt
