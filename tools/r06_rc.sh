#!/bin/bash
# round 6: parity of the split range coder, then one instance alone under rocprofv3 (per-kernel time)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp DSRC_BENCH_NO_FORMS=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "range_coder or selftest or illumina or bucketed or hot" 2>&1 | tail -5 > gpurun_out/r06_rc_parity.txt
cat gpurun_out/r06_rc_parity.txt
D=gpurun_out/prof_rc; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -- python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 2 --warmup 1 --decode-blocks 0 > $D.out 2> $D.err
tail -1 $D.out | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print({k: r[k] for k in ("value","ms_per_step")}, r["roofline"]["kernel_ms"], r["roofline"]["batch_ms"])'
F=$(find $D -name "*.db" | head -1); python tools/prof_summary.py "$F" gpurun_out/r06_rc_p1.txt > /dev/null; head -16 gpurun_out/r06_rc_p1.txt | cut -c1-130
rm -rf $D
