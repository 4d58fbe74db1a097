#!/bin/bash
# Round 3: k_dec_qrc split into one kernel per alphabet size — decode tests, then single passes with kernel statistics.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r03_split_pytest.txt
cat gpurun_out/r03_split_pytest.txt
python tools/decode_bench.py --blocks 2400 --distinct 300 -d 3 -q 2 --passes 3 2>&1 | tail -5
python tools/decode_bench.py --blocks 2400 --distinct 300 -d 0 -q 2 --passes 2 2>&1 | tail -3
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_split -o split -- python /root/repo/tools/decode_bench.py --blocks 2400 --distinct 300 -d 3 -q 2 --passes 2 > /dev/null 2>&1
f=$(find /tmp/prof_split -name '*kernel_stats.csv' | head -1)
head -12 "$f" | cut -c1-200 | tee /root/repo/gpurun_out/r03_split_stats.txt
