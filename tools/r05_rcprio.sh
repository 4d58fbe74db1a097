#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export DSRC_BENCH_NO_FORMS=1 DSRC_GPU_LIB=$PWD/dsrc_amd/csrc/libdsrc_gpu_hooks.so
run() { python bench.py --no-cpu --decode-blocks 0 --check 1 --steps 8 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']; print(l['value'], 'MB/s  ms_per_step', l['ms_per_step'], ' k_rc ms', r['kernel_ms'], ' batch_ms', r['batch_ms'])"; }
for i in 1 2 3; do
echo "== high-priority k_rc stream (shipped)"; run
echo "== normal-priority k_rc stream"; DSRC_GPU_HOOK_RC_PRIO=0 run
echo "== CU-mask stream, all CUs"; DSRC_GPU_HOOK_RC_CUS=256 run
done
