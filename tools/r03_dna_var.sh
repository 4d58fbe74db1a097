#!/bin/bash
# DNA stage diagnostics (results of the x* builds are garbage on purpose: what is left out tells what the stage waits for)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in base $(ls tools/_exp/var | sed 's/libdsrc_gpu_//; s/.so//'); do
  [ $v = base ] && unset DSRC_GPU_LIB || export DSRC_GPU_LIB=$PWD/tools/_exp/var/libdsrc_gpu_$v.so
  echo "== $v"
  python tools/decode_bench.py --blocks 64 --distinct 64 -d 3 -q 0 --passes 2 --check 1 2>&1 | grep -E '"pass": 1|rror' | cut -c1-120
  python tools/decode_bench.py --blocks 2400 --distinct 300 -d 3 -q 0 --passes 2 --check 1 2>&1 | grep -E '"pass": 1|rror' | cut -c1-120
done
