#!/bin/bash
# decoder: how many instances of how many blocks (HBM: ~33 MB per block in flight at -d3 -q2)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for args in "--blocks 2400 --distinct 300 --inst 3 --passes 3 --stagger 1.2" "--blocks 1800 --distinct 300 --inst 4 --passes 3 --stagger 0.9" "--blocks 3000 --distinct 300 --inst 2 --passes 3 --stagger 1.5"; do
  echo "== $args"; timeout 600 python tools/decode_bench.py -d 3 -q 2 $args 2>&1 | grep -E "instances|rror" | tail -3
done
