#!/bin/bash
# rocprofv3 kernel statistics of ONE instance alone on a 512-block batch (the bench workload), for the library as built.
# usage: tools/r04_prof_p1.sh <out.txt> [env assignments...]
OUT=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
D=gpurun_out/prof_tmp_$$; rm -rf $D
env "$@" DSRC_BENCH_NO_FORMS=1 rocprofv3 --kernel-trace --stats -d $D -- python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 2 --warmup 1 --decode-blocks 0 > /dev/null 2> $D.err
F=$(find $D -name "*.db" | head -1); [ -z "$F" ] && F=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$F" $OUT > /dev/null
rm -rf $D $D.err
head -24 $OUT
