#!/bin/bash
# rocprofv3 kernel statistics of the default bench command (instances sharing the GPU): how long every kernel takes under contention
OUT=${1:-gpurun_out/r04_kernel_stats_bench.txt}; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
D=gpurun_out/prof_tmp_$$; rm -rf $D
env "$@" DSRC_BENCH_NO_FORMS=1 rocprofv3 --kernel-trace --stats -d $D -- python bench.py --no-cpu --steps 3 --warmup 1 --decode-blocks 0 > $OUT.json 2> $D.err
F=$(find $D -name "*.db" | head -1); [ -z "$F" ] && F=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$F" $OUT > /dev/null
rm -rf $D $D.err
head -24 $OUT | cut -c1-150
