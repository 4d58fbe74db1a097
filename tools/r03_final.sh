#!/bin/bash
# Round 3: the whole GPU suite, then the default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r03_gpu_suite.txt
cat gpurun_out/r03_gpu_suite.txt
timeout 1500 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
tail -c 3000 gpurun_out/r03_bench_default.json
tail -5 gpurun_out/r03_bench_default.err
