#!/bin/bash
# the default bench (4 instances x 450 blocks) per library variant: tools/r06_full.sh <name> ...   ("built" = the shipped library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export DSRC_BENCH_NO_FORMS=1
for v in "$@"; do
  L=$PWD/dsrc_amd/csrc/libdsrc_gpu.so; [ "$v" != built ] && L=$PWD/dsrc_amd/csrc/_var/lib_$v.so
  [ -f $L ] || { echo "$v: not built"; continue; }
  DSRC_GPU_LIB=$L timeout 600 python bench.py --no-cpu --steps ${STEPS:-6} --warmup 1 --decode-blocks 0 --check ${CHECK:-1} ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c '
import sys, json
try:
    r = json.loads(sys.stdin.read()); print("'$v'", "value", r["value"], "ms_per_step", r["ms_per_step"], "k_rc ms", r["roofline"]["kernel_ms"], "batch ms", r["roofline"]["batch_ms"])
except Exception as e:
    print("'$v'", "failed", e)'
done
