#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun): rocprofv3 kernel statistics of the bench command, and of one instance alone;
# PMC traffic counters of one 512-block compression batch (separate passes, as MI355X_MICROARCH.md prescribes).
# Decoder profiles: tools/r03_dec_probe2.sh / tools/r03_dec_pmc.sh.  Summaries land in gpurun_out/ and are copied into profiles/ by hand.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_ks gpurun_out/pmc_f gpurun_out/pmc_w
DSRC_BENCH_NO_FORMS=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ks -- python bench.py --no-cpu --steps 3 --warmup 1 > gpurun_out/r03_prof_bench_line.json 2> gpurun_out/prof_ks.err
F=$(find gpurun_out/prof_ks -name "*.db" | head -1); [ -z "$F" ] && F=$(find gpurun_out/prof_ks -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$F" gpurun_out/r03_kernel_stats_bench_p5_b1500.txt > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -- python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 1 --warmup 0 --decode-blocks 0 > /dev/null 2> gpurun_out/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -- python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 1 --warmup 0 --decode-blocks 0 > /dev/null 2> gpurun_out/pmc_w.err
python tools/pmc_summary.py $(find gpurun_out/pmc_f gpurun_out/pmc_w -name "*.db") > gpurun_out/r03_pmc_b512_p1_d3q2.txt
rm -rf gpurun_out/prof_p1
DSRC_BENCH_NO_FORMS=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_p1 -- python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 2 --warmup 1 --decode-blocks 0 > /dev/null 2> gpurun_out/prof_p1.err
F=$(find gpurun_out/prof_p1 -name "*.db" | head -1); [ -z "$F" ] && F=$(find gpurun_out/prof_p1 -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$F" gpurun_out/r03_kernel_stats_b512_p1_d3q2.txt > /dev/null
rm -rf gpurun_out/prof_ks gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/prof_p1
ls -la gpurun_out | tail -5
head -12 gpurun_out/r03_kernel_stats_bench_p5_b1500.txt
