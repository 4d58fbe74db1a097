#!/bin/bash
# quick look at the quality decoder: chain (64 blocks) and throughput (2400 blocks) at -d0 -q2, instruction counts at 600 blocks
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
python tools/decode_bench.py --blocks 64 --distinct 64 -d 0 -q 2 --passes 2 2>&1 | grep -E '"pass": 1|rror'
python tools/decode_bench.py --blocks 2400 --distinct 300 -d 0 -q 2 --passes 2 2>&1 | grep -E '"pass": 1|parity|rror'
for G in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_BRANCH"; do
rm -rf /tmp/pmc_d
timeout 600 rocprofv3 --pmc $G --kernel-trace -d /tmp/pmc_d -- python tools/decode_bench.py --blocks 600 --distinct 300 -d 0 -q 2 --passes 1 > /dev/null 2>&1
python tools/pmc_summary.py $(find /tmp/pmc_d -name "*.db" | head -1) | grep -E "k_dec_qrc" | cut -c1-30,100-170
done
[ -n "$QUICK_TESTS" ] && ( timeout 1200 python -m pytest tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -3 )
