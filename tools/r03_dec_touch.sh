#!/bin/bash
# Round 3: DNA decoder with the look-ahead line touch — decode tests, single passes at two sizes, kernel statistics.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r03_touch_pytest.txt
cat gpurun_out/r03_touch_pytest.txt
python tools/decode_bench.py --blocks 64 --distinct 64 -d 3 -q 2 --passes 2 2>&1 | tail -3
python tools/decode_bench.py --blocks 2400 --distinct 300 -d 3 -q 2 --passes 3 2>&1 | tail -4
python tools/decode_bench.py --blocks 3600 --distinct 300 -d 3 -q 2 --passes 2 --inst 2 --stagger 1.9 --check 1 2>&1 | tail -3
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_touch -o touch -- python /root/repo/tools/decode_bench.py --blocks 2400 --distinct 300 -d 3 -q 2 --passes 2 > /dev/null 2>&1
f=$(find /tmp/prof_touch -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -8 "$f" | cut -c1-200 | tee /root/repo/gpurun_out/r03_touch_stats.txt
