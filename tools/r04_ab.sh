#!/bin/bash
# A/B of the bucketed context path (k_bucket.h) against k_sort / k_replay on the bench workload: one instance alone and five.
# usage: tools/r04_ab.sh <outdir>
O=${1:-gpurun_out/r04ab}; mkdir -p $O
run() { # name, pipeline, env...
  local name=$1 p=$2; shift 2
  env "$@" DSRC_BENCH_NO_FORMS=1 python bench.py --steps 6 --warmup 2 --pipeline $p --blocks $((300*p)) --decode-blocks 0 --no-cpu > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], "MB/s  batch_ms", d["roofline"]["batch_ms"], "sort_ms", d["roofline_frontend"]["kernel_ms"], "replay_ms", d["roofline_frontend"].get("model_ms", d["roofline_frontend"].get("replay_ms")), "rc_ms", d["roofline"]["kernel_ms"])
PY
}
run old_p1 1 DSRC_GPU_BUCKETS=0
run new_p1 1 DSRC_GPU_BUCKETS=1
run new_nobin_p1 1 DSRC_GPU_BUCKETS=1 DSRC_GPU_BUCKETS_BINNED=0
run old_p5 5 DSRC_GPU_BUCKETS=0
run new_p5 5 DSRC_GPU_BUCKETS=1
run new_nobin_p5 5 DSRC_GPU_BUCKETS=1 DSRC_GPU_BUCKETS_BINNED=0
