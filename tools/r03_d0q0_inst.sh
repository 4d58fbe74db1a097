#!/bin/bash
# -d0 -q0 decode: one pass and pipelined instances
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "2400 1 0" "3600 2 0.5" "4800 2 0.6" "3600 3 0.4" "6000 2 0.8"; do
  set -- $cfg
  echo "== blocks $1 inst $2 stagger $3"
  timeout 600 python tools/decode_bench.py --blocks $1 --distinct 300 -d 0 -q 0 --passes 3 --inst $2 --stagger $3 --check 1 2>&1 | grep -E '"pass": 2|instances|rror'
done
