#!/bin/bash
# -m1 / -m2 of the reference's command line (src/main.cpp:195-219) are -b64 / -b256: the same records in chunks of 64 and 256 MiB.
# Compression only, device-resident, as many chunks per step as the arena of four instances allows (~9 x the input each).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export DSRC_BENCH_NO_FORMS=1
run() { echo "== $*"; timeout 900 python bench.py --no-cpu --decode-blocks 0 --check 1 "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']; print(l['value'], 'MB/s  ms_per_step', l['ms_per_step'], ' k_rc ms', r['kernel_ms'], ' batch_ms', r['batch_ms'], ' step_frac', r['step_frac'])"; }
run --buf-mb 8 --blocks 1800 --pipeline 4 --steps 5 --warmup 1
run --buf-mb 64 --blocks 224 --pipeline 4 --steps 3 --warmup 1
run --buf-mb 256 --blocks 56 --pipeline 4 --steps 2 --warmup 1
run --buf-mb 256 --blocks 64 --pipeline 8 --steps 2 --warmup 1
