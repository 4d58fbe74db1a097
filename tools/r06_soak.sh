#!/bin/bash
# Round 6 (same as round 5): fuzz soaks of the round's last library (GPU path against the oracle), fresh seeds; half of the time with every stream on the
# bucketed path however short (DSRC_GPU_BUCKETS_MIN=0, a switch of the hooks build: tiny buckets, empty buckets, single-window buckets)
# and with the size from which a bucket counts as large lowered (DSRC_GPU_BUCKET_BIG).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r06_soak.txt; : > $out
S=${SOAK_SEED:-2660000}
H=$PWD/dsrc_amd/csrc/libdsrc_gpu_hooks.so
echo "batch, default" >> $out; timeout 300 python tools/fuzz_soak.py $S 200 batch 2>&1 | tail -2 >> $out
echo "batch, hooks build, DSRC_GPU_BUCKETS_MIN=0 DSRC_GPU_BUCKET_BIG=256" >> $out; DSRC_GPU_LIB=$H DSRC_GPU_BUCKETS_MIN=0 DSRC_GPU_BUCKET_BIG=256 timeout 300 python tools/fuzz_soak.py $((S+10000)) 200 batch 2>&1 | tail -2 >> $out
echo "blocks, hooks build, DSRC_GPU_BUCKETS_MIN=0" >> $out; DSRC_GPU_LIB=$H DSRC_GPU_BUCKETS_MIN=0 timeout 300 python tools/fuzz_soak.py $((S+20000)) 150 2>&1 | tail -2 >> $out
echo "decode round trips" >> $out; timeout 300 python tools/fuzz_soak.py $((S+30000)) 150 decode 2>&1 | tail -2 >> $out
echo "colour space, hooks build, DSRC_GPU_BUCKETS_MIN=0" >> $out; DSRC_GPU_LIB=$H DSRC_GPU_BUCKETS_MIN=0 timeout 200 python tools/fuzz_soak.py $((S+40000)) 100 solid 2>&1 | tail -2 >> $out
cat $out
