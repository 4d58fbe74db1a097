#!/bin/bash
# instruction mix and wait cycles of the decoding kernels (SQ counters, one group per run)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/r03_sq_counters.txt
for B in ${DEC_SIZES:-2400}; do
  out=gpurun_out/r03_dec_sq_b$B.txt; : > $out
  for G in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_IFETCH SQ_WAIT_IFETCH"; do
    rm -rf gpurun_out/pmc_d
    timeout 600 rocprofv3 --pmc $G --kernel-trace -d gpurun_out/pmc_d -- python tools/decode_bench.py --blocks $B --distinct ${DEC_DISTINCT:-300} --passes 1 ${DEC_ARGS} > /dev/null 2> gpurun_out/pmc_d.err
    F=$(find gpurun_out/pmc_d -name "*.db" | head -1)
    [ -n "$F" ] && python tools/pmc_summary.py $F | grep -E "k_dec_qrc|k_dec_dnarc|k_dec_tags_wave|k_dec_qpos" >> $out
  done
  rm -rf gpurun_out/pmc_d
  echo "== B=$B"; cut -c1-32,100-170 $out
done
