#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export DSRC_BENCH_NO_FORMS=1
run() { python bench.py --no-cpu --decode-blocks 0 --check 1 --steps 6 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']; print(l['value'], 'MB/s  ms_per_step', l['ms_per_step'], ' k_rc ms', r['kernel_ms'], ' batch_ms', r['batch_ms'])"; }
echo "== base 4 x 450"; run
echo "== 6 x 300"; run --pipeline 6
echo "== 8 x 225"; run --pipeline 8
echo "== 3 x 600"; run --pipeline 3
for v in part32 place256 both; do echo "== $v 4 x 450"; DSRC_GPU_LIB=$PWD/dsrc_amd/csrc/_var/lib_$v.so run --check 2; done
