#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export DSRC_BENCH_NO_FORMS=1
run() { python bench.py --no-cpu --decode-blocks 0 --check 1 --steps 8 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']; print(l['value'], 'MB/s  ms_per_step', l['ms_per_step'], ' k_rc ms', r['kernel_ms'], ' batch_ms', r['batch_ms'])"; }
for q in 14 8 24 32 14; do echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q run; done
echo "== 5 x 360"; run --pipeline 5
echo "== 4 x 512"; run --blocks 2048
echo "== stagger 0.5"; DSRC_BENCH_STAGGER=0.5 run
echo "== stagger 2"; DSRC_BENCH_STAGGER=2 run
