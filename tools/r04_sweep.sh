#!/bin/bash
# sweeps of the bench configuration with the bucketed path: scheduler instances per GPU, sort-slice size
O=${1:-gpurun_out/r04sweep}; mkdir -p $O
run() { # name, pipeline, blocks per instance, env...
  local name=$1 p=$2 b=$3; shift 3
  env "$@" DSRC_BENCH_NO_FORMS=1 python bench.py --steps 6 --warmup 2 --pipeline $p --blocks $((b*p)) --decode-blocks 0 --no-cpu > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], "MB/s  batch_ms", d["roofline"]["batch_ms"], "sort_ms", d["roofline_frontend"]["kernel_ms"], "replay_ms", d["roofline_frontend"].get("model_ms", d["roofline_frontend"].get("replay_ms")), "rc_ms", d["roofline"]["kernel_ms"])
except Exception as e:
    print("$name failed", e, open("$O/$name.err").read()[-300:])
PY
}
if [ -n "$SWEEP" ]; then for c in $SWEEP; do p=${c%%x*}; b=${c##*x}; run p${p}_b${b} $p $b; done; exit 0; fi
run p5_b300 5 300
run p6_b300 6 300
run p7_b300 7 300
run p6_b256 6 256
run p5_b300_s14 5 300 DSRC_GPU_SORT_SLICE_MB=14336
run p6_b300_s14 6 300 DSRC_GPU_SORT_SLICE_MB=14336
run p5_b300_s3 5 300 DSRC_GPU_SORT_SLICE_MB=3584
run p4_b400 4 400
