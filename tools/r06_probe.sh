#!/bin/bash
# per-variant time of the range coder kernel (HIP events of the instance's coder stream): one instance, 512 blocks
# usage: tools/r06_probe.sh <name> ...   (dsrc_amd/csrc/_var/lib_<name>.so; "built" = the shipped library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export DSRC_BENCH_NO_FORMS=1
for v in "$@"; do
  L=$PWD/dsrc_amd/csrc/libdsrc_gpu.so; [ "$v" != built ] && L=$PWD/dsrc_amd/csrc/_var/lib_$v.so
  [ -f $L ] || { echo "$v: not built"; continue; }
  DSRC_GPU_LIB=$L timeout 300 python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 3 --warmup 1 --decode-blocks 0 --check 0 2>/dev/null | tail -1 | python -c '
import sys, json
try:
    r = json.loads(sys.stdin.read()); print("'$v'", "value", r["value"], "k_rc ms", r["roofline"]["kernel_ms"], "batch ms", r["roofline"]["batch_ms"])
except Exception as e:
    print("'$v'", "failed", e)'
done
