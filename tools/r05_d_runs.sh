#!/bin/bash
# dsrc-amd d on config 3's archive with short gaps after another process has just released its HBM: the case in which
# dsrcgpu_reserve_memory (HBM asked for beside the reading of the archive) is meant to help.  Usage: tools/r05_d_runs.sh [gap seconds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
GAP=${1:-6}
OUT=gpurun_out/r05_d_runs; mkdir -p $OUT
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from dsrc_amd._lib import Handle
reads = 100_000_000
h = Handle(); total = 0; first = 1; piece = 4_000_000
with open("/dev/shm/t.fastq", "wb") as f:
    while first <= reads:
        n_rec = min(piece, reads - first + 1)
        cap = n_rec * 400; d = h.dev_alloc(cap); n = h.synth_illumina(first, n_rec, d, cap)
        f.write(h.dev_download(d, n)); h.dev_free(d); total += n; first += n_rec
h.close()
PY
SIZE=$(stat -c %s /dev/shm/t.fastq)
wall() { local t0=$(date +%s.%N); "$@"; local rc=$?; local t1=$(date +%s.%N); python3 -c "print('WALL %.3f s  %.1f MB/s  rc=%d' % ($t1 - $t0, $SIZE / ($t1 - $t0) / 1e6, $rc))"; }
sleep 10
wall dsrc_amd/csrc/dsrc-amd c -d3 -q2 -t4 /dev/shm/t.fastq /dev/shm/t.dsrc 2>&1 | tail -1
for i in 1 2 3 4 5; do
  sleep $GAP; rm -f /dev/shm/t_back.fastq
  wall env DSRC_HOST_TRACE=1 dsrc_amd/csrc/dsrc-amd d -t4 /dev/shm/t.dsrc /dev/shm/t_back.fastq > $OUT/d_$i.txt 2>&1; tail -1 $OUT/d_$i.txt
done
cmp /dev/shm/t.fastq /dev/shm/t_back.fastq && echo "round trip identical"
grep -E "read|decoded|reserved|done|start" $OUT/d_1.txt | head -30
rm -f /dev/shm/t.fastq /dev/shm/t.dsrc /dev/shm/t_back.fastq
