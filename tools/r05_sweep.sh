#!/bin/bash
# bench.py (compression only) over the element-slice size, with the hooks build (the switch exists only there): ~32 / 64 / 128 streams per launch group
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export DSRC_BENCH_NO_FORMS=1
for mb in ${SLICES:-1792 3584 7168}; do
  echo "== slice $mb MiB =="
  DSRC_GPU_LIB=$PWD/dsrc_amd/csrc/libdsrc_gpu_hooks.so DSRC_GPU_SORT_SLICE_MB=$mb python bench.py --steps ${STEPS:-6} --warmup 1 --no-cpu --decode-blocks 0 "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); print(l['value'], 'MB/s  ms_per_step', l['ms_per_step'], 'k_rc', l['roofline']['kernel_ms'], 'batch_ms', l['roofline']['batch_ms'])"
done
