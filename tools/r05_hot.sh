#!/bin/bash
# Hot contexts: windows of large buckets coded by counting (k_model, MD_PAIRS) and a window's leading bucket ranked by one add
# (k_part, PART_PEEL) -- variants of dsrc_amd/csrc/_var against the built library, on the four-level-quality shards and on the
# default ones (compression legs only).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for v in "" np a4 a4np a6; do
  L=$PWD/dsrc_amd/csrc/libdsrc_gpu.so; [ -n "$v" ] && L=$PWD/dsrc_amd/csrc/_var/lib_$v.so
  [ -f $L ] || continue
  echo "== ${v:-built}"
  DSRC_GPU_LIB=$L timeout 300 python tools/binned_bench.py 2>&1 | tail -1 | cut -c1-160
  DSRC_GPU_LIB=$L timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --decode-blocks 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default MB/s', d['value'], 'ms', d['ms_per_step'])"
done
