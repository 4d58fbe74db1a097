#!/bin/bash
# decoder PMC passes: one counter group per run (as MI355X_MICROARCH.md prescribes), decode-only passes of $DEC_SIZES blocks
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --list-avail 2>/dev/null | grep -i -E "Counter_Name" | grep -i -E "UTCL1|TCC_EA|TCP_PENDING|TCP_TA|TCC_NC|TCC_UC|TCC_RW|TCC_PROBE|TCC_STREAMING|TCC_BUBBLE|HBM|MALL|TCC_READ|TCC_WRITE" | cut -c1-120 | sort | uniq > gpurun_out/r03_counters2.txt
for B in ${DEC_SIZES:-64 2400}; do
  : > gpurun_out/r03_dec_pmc_b$B.txt
  for G in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_TAG_STALL_sum" "FETCH_SIZE" "WRITE_SIZE" ${DEC_PMC_EXTRA}; do
    rm -rf gpurun_out/pmc_d
    timeout 600 rocprofv3 --pmc $G --kernel-trace -d gpurun_out/pmc_d -- python tools/decode_bench.py --blocks $B --distinct ${DEC_DISTINCT:-300} --passes 1 > /dev/null 2> gpurun_out/pmc_d.err
    python tools/pmc_summary.py $(find gpurun_out/pmc_d -name "*.db") | grep "k_dec" >> gpurun_out/r03_dec_pmc_b$B.txt
  done
  rm -rf gpurun_out/pmc_d
  echo "== B=$B"; cat gpurun_out/r03_dec_pmc_b$B.txt | cut -c1-150
done
