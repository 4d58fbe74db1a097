#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "3600 2 1.0" "3600 2 1.5" "3600 2 2.0" "3600 2 2.6" "2400 3 1.2" "3600 3 1.5"; do
  set -- $cfg
  echo "== blocks $1 inst $2 stagger $3"
  timeout 600 python tools/decode_bench.py --blocks $1 --distinct 300 -d 3 -q 2 --passes 2 --inst $2 --stagger $3 --check 1 2>&1 | grep -E "instances|rror" | cut -c1-140
done
