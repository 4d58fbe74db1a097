#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r03_verify_probe.txt; : > $out
for i in 1 2 4; do echo "== verify inst $i" >> $out; timeout 300 python tools/verify_probe.py --blocks 300 --inst $i --passes 2 2>&1 | tail -12 >> $out; done
for i in 1 4; do echo "== decode 300 blocks inst $i" >> $out; timeout 300 python tools/decode_bench.py --blocks 300 --distinct 300 --inst $i --stagger 0 --passes 2 --check 1 2>&1 | grep -E "pass\"|instances" >> $out; done
echo "== verify inst 4, DSRC_HOST trace of the sync points" >> $out
cat $out
