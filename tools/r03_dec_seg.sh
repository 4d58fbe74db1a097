#!/bin/bash
# k_dec_qrc with the segmented loop: decode tests, single passes, instruction mix
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -3 )
python tools/decode_bench.py --blocks 64 --distinct 64 -d 3 -q 2 --passes 2 2>&1 | grep '"pass": 1'
python tools/decode_bench.py --blocks 2400 --distinct 300 -d 3 -q 2 --passes 2 2>&1 | grep '"pass": 1'
python tools/decode_bench.py --blocks 2400 --distinct 300 -d 0 -q 2 --passes 2 2>&1 | grep '"pass": 1'
python tools/decode_bench.py --blocks 3600 --distinct 300 -d 3 -q 2 --passes 2 --inst 2 --stagger 1.9 --check 1 2>&1 | grep instances
rm -rf gpurun_out/pmc_d
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d gpurun_out/pmc_d -- python tools/decode_bench.py --blocks 2400 --distinct 300 --passes 1 > /dev/null 2> gpurun_out/pmc_d.err
python tools/pmc_summary.py $(find gpurun_out/pmc_d -name "*.db" | head -1) | grep -E "k_dec_qrc" | cut -c1-30,100-170
rm -rf gpurun_out/pmc_d
timeout 600 rocprofv3 --pmc SQ_INSTS_BRANCH --kernel-trace -d gpurun_out/pmc_d -- python tools/decode_bench.py --blocks 2400 --distinct 300 --passes 1 > /dev/null 2> gpurun_out/pmc_d.err
python tools/pmc_summary.py $(find gpurun_out/pmc_d -name "*.db" | head -1) | grep -E "k_dec_qrc" | cut -c1-30,100-170
rm -rf gpurun_out/pmc_d
