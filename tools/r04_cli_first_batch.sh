#!/bin/bash
# dsrc-amd c on a 3.3 GB file: where the first batch of every scheduler instance spends its time (DSRC_GPU_DEBUG=2 timelines)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from dsrc_amd._lib import Handle
h = Handle(); total = 0; first = 1
with open("/dev/shm/t.fastq", "wb") as f:
    while total < 6.5e9:
        cap = 2_000_000 * 400; d = h.dev_alloc(cap); n = h.synth_illumina(first, 2_000_000, d, cap)
        f.write(h.dev_download(d, n)); h.dev_free(d); total += n; first += 2_000_000
h.close()
PY
sleep 3
DSRC_HOST_TRACE=1 DSRC_GPU_DEBUG=2 dsrc_amd/csrc/dsrc-amd c -d3 -q2 -t4 /dev/shm/t.fastq /dev/shm/t.dsrc 2>&1 | grep -E "timeline|instance|batch [0-9]+ \(|closed|arena" | head -30
rm -f /dev/shm/t.fastq /dev/shm/t.dsrc
