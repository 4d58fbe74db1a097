#!/bin/bash
# Round 4: fuzz soaks of the library with the bucketed front end (GPU path against the oracle), fresh seeds; half of the time with
# every stream on the bucketed path however short (DSRC_GPU_BUCKETS_MIN=0: tiny buckets, empty buckets, single-window buckets).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r04_soak.txt; : > $out
S=${SOAK_SEED:-2440000}
echo "batch, default" >> $out; timeout 300 python tools/fuzz_soak.py $S 200 batch 2>&1 | tail -2 >> $out
echo "batch, DSRC_GPU_BUCKETS_MIN=0" >> $out; DSRC_GPU_BUCKETS_MIN=0 timeout 300 python tools/fuzz_soak.py $((S+10000)) 200 batch 2>&1 | tail -2 >> $out
echo "blocks, DSRC_GPU_BUCKETS_MIN=0" >> $out; DSRC_GPU_BUCKETS_MIN=0 timeout 300 python tools/fuzz_soak.py $((S+20000)) 150 2>&1 | tail -2 >> $out
echo "decode round trips" >> $out; timeout 300 python tools/fuzz_soak.py $((S+30000)) 150 decode 2>&1 | tail -2 >> $out
echo "colour space, DSRC_GPU_BUCKETS_MIN=0" >> $out; DSRC_GPU_BUCKETS_MIN=0 timeout 200 python tools/fuzz_soak.py $((S+40000)) 100 solid 2>&1 | tail -2 >> $out
cat $out
