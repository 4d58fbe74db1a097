#!/bin/bash
# Data-parallel phases of the scheduler instances one after the other (FeTurn, dsrc_gpu.hip) against all at once (DSRC_GPU_NO_TURNS=1):
# the default bench's compression leg and the four-level-quality one, instance counts 3 / 4 / 6.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu --decode-blocks 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   MB/s', d['value'], 'ms', d['ms_per_step'])"; }
for P in ${PIPES:-4 3 6}; do
  for rep in 1 2; do
    echo "== turns, $P instances";       run --pipeline $P --blocks $((P * 450))
    echo "== all at once, $P instances"; DSRC_GPU_NO_TURNS=1 run --pipeline $P --blocks $((P * 450))
  done
done
echo "== four-level qualities: turns"; timeout 300 python tools/binned_bench.py 2>&1 | tail -1 | cut -c1-140
echo "== four-level qualities: all at once"; DSRC_GPU_NO_TURNS=1 timeout 300 python tools/binned_bench.py 2>&1 | tail -1 | cut -c1-140
