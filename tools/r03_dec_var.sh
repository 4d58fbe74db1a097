#!/bin/bash
# variants of the quality decoder's loop (tools/_exp/var/libdsrc_gpu_<v>.so): parity on a few blocks is part of decode_bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for v in base $(ls tools/_exp/var | sed 's/libdsrc_gpu_//; s/.so//'); do
  [ $v = base ] && unset DSRC_GPU_LIB || export DSRC_GPU_LIB=$PWD/tools/_exp/var/libdsrc_gpu_$v.so
  echo "== $v"
  python tools/decode_bench.py --blocks 64 --distinct 64 -d 0 -q 2 --passes 2 2>&1 | grep -E '"pass": 1|rror'
  python tools/decode_bench.py --blocks 2400 --distinct 300 -d 0 -q 2 --passes 2 2>&1 | grep -E '"pass": 1|parity|rror'
done
