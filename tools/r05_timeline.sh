#!/bin/bash
# GPU busy / idle from the kernel trace (tools/prof_timeline.py): one instance alone, then the default four
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for cfg in "4 1800" "6 2700"; do
  set -- $cfg
  D=gpurun_out/prof_tl; rm -rf $D
  rocprofv3 --kernel-trace -d $D -- python bench.py --no-cpu --pipeline $1 --blocks $2 --steps 4 --warmup 1 --decode-blocks 0 > $D.out 2> $D.err
  tail -1 $D.out | cut -c1-120
  F=$(find $D -name "*.db" | head -1)
  # the timed steps are the last ones: look at the last 60 % of the trace
  python - "$F" <<'PY' > gpurun_out/r05_tl_range.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
a, b = cur.execute(f"select min(start), max(end) from {kd}").fetchone()
span = (b - a) / 1e6
print(span - 800, span - 60)
PY
  python tools/prof_timeline.py "$F" $(cat gpurun_out/r05_tl_range.txt) | tee gpurun_out/r05_timeline_p$1.txt | head -14
  rm -rf $D $D.err $D.out gpurun_out/r05_tl_range.txt
done
