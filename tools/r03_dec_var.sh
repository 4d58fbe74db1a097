#!/bin/bash
# variants of the quality decoder's loop (tools/_exp/var/libdsrc_gpu_<v>.so, built with -DQRC_E<n>): parity on a few blocks is part of decode_bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for v in base $(ls tools/_exp/var | sed 's/libdsrc_gpu_//; s/.so//'); do
  [ $v = base ] && unset DSRC_GPU_LIB || export DSRC_GPU_LIB=$PWD/tools/_exp/var/libdsrc_gpu_$v.so
  echo "== $v"
  python tools/decode_bench.py --blocks 64 --distinct 64 -d 0 -q 2 --passes 2 2>&1 | grep -E '"pass": 1|rror'
  python tools/decode_bench.py --blocks 2400 --distinct 300 -d 0 -q 2 --passes 2 2>&1 | grep -E '"pass": 1|parity|rror'
  rm -rf /tmp/pmc_d
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d /tmp/pmc_d -- python tools/decode_bench.py --blocks 600 --distinct 300 -d 0 -q 2 --passes 1 > /dev/null 2>&1
  python tools/pmc_summary.py $(find /tmp/pmc_d -name "*.db" | head -1) | grep -E "k_dec_qrc" | cut -c1-30,100-170
done
