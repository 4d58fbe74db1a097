#!/bin/bash
# everything the round's closing numbers come from, on the shipped library: kernel-trace summaries, the PMC passes, the default bench line,
# large blocks, one handle's lanes, four-level qualities.  Output: gpurun_out/profiles/ (copied to profiles/ afterwards).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/profiles; mkdir -p $O
bash tools/r06_profiles.sh > $O/r06_profiles.log 2>&1
bash tools/r06_pmc.sh > $O/r06_pmc.log 2>&1
( time python bench.py ) > $O/r06_bench_default.json 2> $O/r06_bench_default.err
{
  export DSRC_BENCH_NO_FORMS=1
  run() { echo "== $*"; timeout 900 python bench.py --no-cpu --decode-blocks 0 --check 1 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']; print(l['value'], 'MB/s  ms_per_step', l['ms_per_step'], ' k_rcs ms', r['kernel_ms'], ' batch_ms', r['batch_ms'], ' hbm GB', l['config'].get('hbm_held_GB'))"; }
  run --buf-mb 64 --blocks 224 --pipeline 4 --steps 3 --warmup 1
  run --buf-mb 256 --blocks 56 --pipeline 4 --steps 2 --warmup 1
  run --pipeline 5 --blocks 2250 --steps 6 --warmup 1
  run --pipeline 1 --blocks 512 --steps 3 --warmup 1
  run --pipeline 1 --blocks 1800 --lanes 4 --steps 4 --warmup 1
  run --pipeline 1 --blocks 1800 --lanes 4 --sub-blocks 150 --steps 4 --warmup 1
} > $O/r06_closing_runs.txt 2>&1
python tools/binned_bench.py > $O/r06_binned.txt 2>&1
for l in 3; do echo "== DSRC_GPU_QUEUE_LANES=$l"; DSRC_GPU_QUEUE_LANES=$l timeout 600 python tools/queue_bench.py 12 192 2>&1 | grep "chunks per flush"; DSRC_GPU_QUEUE_LANES=$l timeout 600 python tools/queue_bench.py 12 384 2>&1 | grep pinned | head -1; done > $O/r06_queue_closing.txt 2>&1
ls -la $O
