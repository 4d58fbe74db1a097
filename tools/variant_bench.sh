#!/bin/bash
# Compare builds of libdsrc_gpu.so (dsrc_amd/csrc/_var/lib_<name>.so, made with extra -D switches) on the GPU box:
# one scheduler instance alone (stage timers) and the default 5-instance bench.  Output: gpurun_out/variants.txt
# usage: tools/variant_bench.sh <name> [<name> ..]
mkdir -p gpurun_out
out=${VB_OUT:-gpurun_out/variants.txt}
: > $out
for v in "$@"; do
	lib=$PWD/dsrc_amd/csrc/_var/lib_$v.so
	[ -f "$lib" ] || { echo "$v: no such build" >> $out; continue; }
	echo "== $v p1" >> $out
	DSRC_GPU_LIB=$lib timeout 300 python bench.py --pipeline 1 --blocks 512 --steps 3 --no-cpu --decode-blocks 0 --check ${VB_CHECK:-2} 2>&1 | tail -1 | python -c '
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(json.dumps({k: r[k] for k in ("value", "ms_per_step")}), json.dumps(r.get("roofline", {}).get("kernel_ms")), json.dumps(r.get("roofline_frontend", r.get("stages"))))
except Exception as e:
    print("failed:", e)
' >> $out
	echo "== $v p5" >> $out
	DSRC_GPU_LIB=$lib timeout 300 python bench.py --steps ${VB_STEPS:-5} --no-cpu --decode-blocks 0 --check 0 2>&1 | tail -1 | python -c '
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(json.dumps({k: r[k] for k in ("value", "ms_per_step")}))
except Exception as e:
    print("failed:", e)
' >> $out
done
cat $out
