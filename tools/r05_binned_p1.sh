cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp DSRC_BENCH_NO_FORMS=1
D=gpurun_out/prof_tmp_b; rm -rf $D; rocprofv3 --kernel-trace --stats -d $D -- python tools/binned_bench.py 512 1 2 > $D.out 2> $D.err; F=$(find $D -name "*.db" | head -1); python tools/prof_summary.py "$F" $D.txt > /dev/null; head -16 $D.txt | cut -c1-125; rm -rf $D $D.err $D.out $D.txt
