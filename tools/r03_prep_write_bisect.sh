#!/bin/bash
# Bisection builds of the k_prep_write discrepancy (DSRC_PREP_WRITE_IN_IF = 1 the round-2 form, 2 lossy branch compiled out,
# 3 `keep` as an integer, 4 loads hoisted; plus the round-2 form at -O1 and with the switch form of dna_index): which ones miscompare?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
C=dsrc_amd/csrc
FLAGS="--offload-arch=gfx950 -std=c++17 -Wno-unused-value -Wno-unused-result -shared -fPIC"
: > gpurun_out/r03_prep_write_bisect.txt
run() {  # name, flags
  /opt/rocm/bin/hipcc $FLAGS $2 -o /tmp/libv.so $C/dsrc_gpu.hip 2>/dev/null
  R=$(DSRC_TEST_KEEP_GPU_LIB=1 DSRC_GPU_LIB=/tmp/libv.so python -m pytest tests/test_gpu_parity.py -q -m gpu -k "illumina_20k or tiny or wave_boundaries" 2>&1 | tail -1)
  echo "$1: $R" | tee -a gpurun_out/r03_prep_write_bisect.txt
}
run "shipped form, -O3" "-O3"
run "call under if (in_r), -O3" "-O3 -DDSRC_PREP_WRITE_IN_IF=1"
run "call under if (in_r), -O1" "-O1 -DDSRC_PREP_WRITE_IN_IF=1"
run "call under if (in_r), switch form of dna_index" "-O3 -DDSRC_PREP_WRITE_IN_IF=1 -DFAST_WRITE=false"
run "call under if (in_r), lossy branch compiled out" "-O3 -DDSRC_PREP_WRITE_IN_IF=2"
run "call under if (in_r), keep as integer" "-O3 -DDSRC_PREP_WRITE_IN_IF=3"
run "call under if (in_r), loads hoisted" "-O3 -DDSRC_PREP_WRITE_IN_IF=4"
