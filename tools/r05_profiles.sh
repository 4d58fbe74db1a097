#!/bin/bash
# rocprofv3 summaries of the shipped library (copied to profiles/ afterwards):
#   r05_kernel_stats_bench_p4_b1800.txt  -- the default bench command (four instances sharing the GPU), compression only
#   r05_kernel_stats_b512_p1.txt         -- one instance alone, 512 blocks
#   r05_kernel_stats_decode.txt          -- one decoding pass of 2400 blocks
#   r05_sq_decode_b2400.txt              -- instruction counts of the decoding kernels (SQ counters, one group per run) per quality symbol
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp DSRC_BENCH_NO_FORMS=1
O=gpurun_out/profiles; mkdir -p $O
prof() { local out=$1; shift; local D=gpurun_out/prof_tmp_$$; rm -rf $D; rocprofv3 --kernel-trace --stats -d $D -- "$@" > $D.out 2> $D.err; local F=$(find $D -name "*.db" | head -1); python tools/prof_summary.py "$F" $out > /dev/null; rm -rf $D $D.err $D.out; head -14 $out | cut -c1-132; }
prof $O/r05_kernel_stats_bench_p4_b1800.txt python bench.py --no-cpu --steps 3 --warmup 1 --decode-blocks 0
prof $O/r05_kernel_stats_b512_p1.txt python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 2 --warmup 1 --decode-blocks 0
prof $O/r05_kernel_stats_decode.txt python tools/decode_bench.py --blocks 2400 --distinct 300 -d 3 -q 2 --passes 2
out=$O/r05_sq_decode_b2400.txt; : > $out
for G in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  rm -rf gpurun_out/pmc_d
  timeout 600 rocprofv3 --pmc $G --kernel-trace -d gpurun_out/pmc_d -- python tools/decode_bench.py --blocks 2400 --distinct 300 -d 3 -q 2 --passes 1 > /dev/null 2> gpurun_out/pmc_d.err
  F=$(find gpurun_out/pmc_d -name "*.db" | head -1)
  [ -n "$F" ] && python tools/pmc_summary.py $F | grep -E "k_dec_qrc|k_dec_dnarc|k_dec_tags_wave" >> $out
done
rm -rf gpurun_out/pmc_d gpurun_out/pmc_d.err
python3 - $out <<'PY'
import re, sys
rows = {}
for l in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+(?:void )?(k_dec_\w+).*per_launch=\s*(\d+)", l)
    if m: rows[(m.group(2), m.group(1))] = int(m.group(3))
sym = 2400 * 3334950            # quality symbols (= bases) of 2400 blocks of the benchmark data (22233 records x 150)
with open(sys.argv[1], "a") as f:
    for k in ("k_dec_qrc", "k_dec_dnarc", "k_dec_tags_wave"):
        g = lambda c: rows.get((k, c), 0)
        per = {c: g(c) / sym for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH")}
        tot = sum(per[c] for c in per if c != "SQ_INSTS_BRANCH")      # branches are scalar instructions: counted in SQ_INSTS_SALU
        line = f"{k}: per symbol of a block (3.33 M): " + ", ".join(f"{c[9:]} {v:.1f}" for c, v in per.items()) + f"; all {tot:.1f}" + (f" (per wave of 64 streams: x 64 = {tot * 64:.0f} per wave-step)" if k == "k_dec_dnarc" else "")
        print(line); f.write(line + "\n")
PY
