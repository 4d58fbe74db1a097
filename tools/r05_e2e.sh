#!/bin/bash
# dsrc-amd c / d on BASELINE configs[2]'s own file (100 M reads, 37.8 GB) in tmpfs: one traced run of each (where the time goes:
# process start -> instances ready -> first batch -> steady state -> last batch -> exit) and three plain runs (wall, min / median).
# Usage: tools/r05_e2e.sh [reads, default 100000000] [extra dsrc-amd c switches]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
READS=${1:-100000000}; shift
OUT=gpurun_out/r05_e2e; mkdir -p $OUT
python - $READS <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
from dsrc_amd._lib import Handle
reads = int(sys.argv[1])
h = Handle(); total = 0; first = 1; piece = 4_000_000
t = time.time()
with open("/dev/shm/t.fastq", "wb") as f:
    while first <= reads:
        n_rec = min(piece, reads - first + 1)
        cap = n_rec * 400; d = h.dev_alloc(cap); n = h.synth_illumina(first, n_rec, d, cap)
        f.write(h.dev_download(d, n)); h.dev_free(d); total += n; first += n_rec
h.close()
print("wrote %d reads, %d bytes in %.1f s" % (reads, total, time.time() - t))
PY
SIZE=$(stat -c %s /dev/shm/t.fastq)
free -g | head -2
nproc
wall() { local t0=$(date +%s.%N); "$@"; local rc=$?; local t1=$(date +%s.%N); python3 -c "print('WALL %.3f s  %.1f MB/s  rc=%d' % ($t1 - $t0, $SIZE / ($t1 - $t0) / 1e6, $rc))"; }
echo "== traced compress =="; sleep 5; rm -f /dev/shm/t.dsrc
wall env DSRC_HOST_TRACE=1 DSRC_GPU_DEBUG=2 dsrc_amd/csrc/dsrc-amd c -d3 -q2 -t4 "$@" /dev/shm/t.fastq /dev/shm/t.dsrc > $OUT/c_trace.txt 2>&1; tail -1 $OUT/c_trace.txt
grep -vE "timeline|arena used" $OUT/c_trace.txt | head -80
grep -E "timeline" $OUT/c_trace.txt | head -8
for i in 1 2 3; do
  sleep 6; rm -f /dev/shm/t.dsrc
  wall dsrc_amd/csrc/dsrc-amd c -d3 -q2 -t4 "$@" /dev/shm/t.fastq /dev/shm/t.dsrc 2>&1 | tail -1 | tee -a $OUT/c_runs.txt
done
ls -l /dev/shm/t.dsrc
echo "== traced decompress =="; sleep 6; rm -f /dev/shm/t_back.fastq
wall env DSRC_HOST_TRACE=1 dsrc_amd/csrc/dsrc-amd d ${DARGS:--t4} /dev/shm/t.dsrc /dev/shm/t_back.fastq > $OUT/d_trace.txt 2>&1; tail -1 $OUT/d_trace.txt
head -60 $OUT/d_trace.txt
for i in 1 2 3; do
  sleep 6; rm -f /dev/shm/t_back.fastq
  wall dsrc_amd/csrc/dsrc-amd d ${DARGS:--t4} /dev/shm/t.dsrc /dev/shm/t_back.fastq 2>&1 | tail -1 | tee -a $OUT/d_runs.txt
done
cmp /dev/shm/t.fastq /dev/shm/t_back.fastq && echo "round trip identical"
rm -f /dev/shm/t.fastq /dev/shm/t.dsrc /dev/shm/t_back.fastq
