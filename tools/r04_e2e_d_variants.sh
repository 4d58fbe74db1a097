#!/bin/bash
# dsrc-amd d on one 16.6 GB archive with different pass shapes (handles per device x blocks per pass): wall time start to exit
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
from dsrc_amd._lib import Handle
h = Handle(); total = 0; first = 1
with open("/dev/shm/t.fastq", "wb") as f:
    while total < 16e9:
        cap = 2_000_000 * 400; d = h.dev_alloc(cap); n = h.synth_illumina(first, 2_000_000, d, cap)
        f.write(h.dev_download(d, n)); h.dev_free(d); total += n; first += 2_000_000
h.close()
PY
dsrc_amd/csrc/dsrc-amd c -d3 -q2 -t4 /dev/shm/t.fastq /dev/shm/t.dsrc
for a in "-t4" "-t1 -n2000" "-t2 -n1000" "-t2 -n700" "-t3 -n500" "-t4"; do
  sleep 4; rm -f /dev/shm/t_back.fastq
  t0=$(date +%s.%N)
  DSRC_HOST_TRACE=1 dsrc_amd/csrc/dsrc-amd d $a /dev/shm/t.dsrc /dev/shm/t_back.fastq 2>&1 | grep -E "all decoded" | tail -1
  t1=$(date +%s.%N); python -c "print(\"$a: wall %.2f s\" % ($t1 - $t0))"
done
cmp /dev/shm/t.fastq /dev/shm/t_back.fastq && echo identical
rm -f /dev/shm/t.fastq /dev/shm/t.dsrc /dev/shm/t_back.fastq
