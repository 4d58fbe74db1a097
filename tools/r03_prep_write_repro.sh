#!/bin/bash
# Round 3, VERDICT task 5: the k_prep_write form that gave a wrong quality stream on the GPU in round 2 (transform_base called under
# `if (in_r)`), rebuilt as a variant library (-DDSRC_PREP_WRITE_IN_IF), run through the GPU parity tests, and its ISA next to the
# shipped form's.  Output: gpurun_out/r03_prep_write_repro.txt, gpurun_out/r03_prep_write_{good,bad}.s
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
C=dsrc_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS -shared -fPIC -DDSRC_PREP_WRITE_IN_IF=1 -o /tmp/libdsrc_gpu_bad.so $C/dsrc_gpu.hip
for v in good bad; do
  D=""; [ $v = bad ] && D="-DDSRC_PREP_WRITE_IN_IF=1"
  /opt/rocm/bin/hipcc $FLAGS $D -S --cuda-device-only -o /tmp/all_$v.s $C/dsrc_gpu.hip 2>/dev/null
  awk '/^_Z12k_prep_write/,/\.end_amdhsa_kernel/' /tmp/all_$v.s > gpurun_out/r03_prep_write_$v.s
done
{
  echo "== shipped form"; python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiny or illumina or fuzz" 2>&1 | tail -3
  echo "== call under if (in_r)"; DSRC_TEST_KEEP_GPU_LIB=1 DSRC_GPU_LIB=/tmp/libdsrc_gpu_bad.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiny or illumina or fuzz" 2>&1 | tail -15
  echo "== ISA sizes"; wc -l gpurun_out/r03_prep_write_good.s gpurun_out/r03_prep_write_bad.s
} > gpurun_out/r03_prep_write_repro.txt 2>&1
cat gpurun_out/r03_prep_write_repro.txt
