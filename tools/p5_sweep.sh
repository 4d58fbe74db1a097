#!/bin/bash
# p5 what-ifs on the GPU box: each argument is "ENV=.. ENV=.. [--bench-arg ..]"; prints value per configuration
for cfg in "$@"; do
	envs=""; args=""
	for w in $cfg; do case $w in --*|[0-9]*) args="$args $w";; *) envs="$envs $w";; esac; done
	v=$(env $envs timeout 300 python bench.py --steps ${P5_STEPS:-5} --no-cpu --decode-blocks 0 --check 0 $args 2>&1 | tail -1 | python -c 'import sys,json
try: print(json.loads(sys.stdin.read())["value"])
except Exception as e: print("failed", e)')
	echo "$cfg => $v"
done
