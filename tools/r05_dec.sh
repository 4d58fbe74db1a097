#!/bin/bash
# decoder alone: passes of 64 / 2400 / 4500 blocks at -d0 -q2 (quality stage only) and -d3 -q2, one instance; then two instances
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for args in "--blocks 64 --distinct 64 -d 0 -q 2" "--blocks 2400 --distinct 300 -d 0 -q 2" "--blocks 2400 --distinct 300 -d 3 -q 2" "--blocks 4500 --distinct 300 -d 3 -q 2" "--blocks 3600 --distinct 300 -d 3 -q 2 --inst 2 --passes 3"; do
  echo "== $args"; DSRC_GPU_DEBUG=2 timeout 600 python tools/decode_bench.py $args 2>&1 | grep -E "decode timeline|MB/s|value|Error|error" | tail -6
done
