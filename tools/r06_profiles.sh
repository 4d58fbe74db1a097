#!/bin/bash
# rocprofv3 summaries of the shipped library (copied to profiles/ afterwards):
#   r06_kernel_stats_bench_p4_b1800.txt  -- the default bench command (four handles sharing the GPU), compression only
#   r06_kernel_stats_b512_p1.txt         -- one handle, one lane, 512 blocks
#   r06_kernel_stats_lanes_b1800.txt     -- one handle, four scheduler lanes inside it, calls of 1800 blocks
#   r06_kernel_stats_decode.txt          -- decoding passes of 2400 blocks
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp DSRC_BENCH_NO_FORMS=1
O=gpurun_out/profiles; mkdir -p $O
prof() { local out=$1; shift; local D=gpurun_out/prof_tmp_$$; rm -rf $D; rocprofv3 --kernel-trace --stats -d $D -- "$@" > $D.out 2> $D.err; tail -1 $D.out | cut -c1-160; local F=$(find $D -name "*.db" | head -1); python tools/prof_summary.py "$F" $out > /dev/null; rm -rf $D $D.err $D.out; head -12 $out | cut -c1-132; }
prof $O/r06_kernel_stats_bench_p4_b1800.txt python bench.py --no-cpu --steps 3 --warmup 1 --decode-blocks 0
prof $O/r06_kernel_stats_b512_p1.txt python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 2 --warmup 1 --decode-blocks 0
prof $O/r06_kernel_stats_lanes_b1800.txt python bench.py --no-cpu --pipeline 1 --blocks 1800 --lanes 4 --steps 3 --warmup 1 --decode-blocks 0
prof $O/r06_kernel_stats_decode.txt python tools/decode_bench.py --blocks 2400 --distinct 300 -d 3 -q 2 --passes 2
