#!/bin/bash
# per-kernel time of the default bench (4 instances x 450 blocks) under rocprofv3 --kernel-trace --stats -> profiles/r06_kernel_stats_bench_<tag>.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp DSRC_BENCH_NO_FORMS=1
TAG=${1:-p4_b1800}; shift
D=gpurun_out/prof_bench; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -- python bench.py --no-cpu --steps 4 --warmup 1 --decode-blocks 0 --check 0 "$@" > $D.out 2> $D.err
tail -1 $D.out | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print({k: r[k] for k in ("value","ms_per_step")}, r["roofline"]["kernel_ms"], r["roofline"]["batch_ms"])'
F=$(find $D -name "*.db" | head -1); python tools/prof_summary.py "$F" gpurun_out/r06_kernel_stats_bench_$TAG.txt > /dev/null; head -24 gpurun_out/r06_kernel_stats_bench_$TAG.txt | cut -c1-125
rm -rf $D
