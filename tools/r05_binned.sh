#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
# the four-level-quality line of the bench on its own, then its kernel statistics
timeout 300 python tools/binned_bench.py 2>&1 | tail -1 | cut -c1-200
D=gpurun_out/prof_binned; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -- python tools/binned_bench.py 1800 4 2 > /dev/null 2> $D.err
F=$(find $D -name "*.db" | head -1); python tools/prof_summary.py "$F" gpurun_out/r05_kernel_stats_binned.txt > /dev/null; head -16 gpurun_out/r05_kernel_stats_binned.txt | cut -c1-140
rm -rf $D $D.err
