#!/bin/bash
# the bench workload at other compression levels (4 instances x 450 blocks), compress only
O=gpurun_out/r04_levels; mkdir -p $O
for lv in "3 2" "2 2" "1 1" "3 0" "0 2" "0 0"; do
  set -- $lv
  DSRC_BENCH_NO_FORMS=1 python bench.py --steps 5 --warmup 1 --dna $1 --qua $2 --decode-blocks 0 --no-cpu > $O/d$1q$2.json 2> $O/d$1q$2.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/d$1q$2.json").read().strip().splitlines()[-1]); print("-d$1 -q$2", d["value"], "MB/s, ratio", d["config"]["ratio_out_in"])
except Exception as e: print("-d$1 -q$2 failed", open("$O/d$1q$2.err").read()[-300:])
PY
done
