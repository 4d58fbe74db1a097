#!/bin/bash
# timeline of `dsrc-amd d` on the 38.5 GB set: default plan and two large passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
from dsrc_amd._lib import Handle
h = Handle(); total = 0; first = 1
with open("/dev/shm/t.fastq", "wb") as f:
    while total < 38.5e9:
        cap = 2_000_000 * 400; d = h.dev_alloc(cap); n = h.synth_illumina(first, 2_000_000, d, cap)
        f.write(h.dev_download(d, n)); h.dev_free(d); total += n; first += 2_000_000
h.close()
PY
dsrc_amd/csrc/dsrc-amd c -d3 -q2 -t4 /dev/shm/t.fastq /dev/shm/t.dsrc
rm -f /dev/shm/t_back.fastq; sleep 5
echo "== default"
( time DSRC_HOST_TRACE=1 dsrc_amd/csrc/dsrc-amd d -t4 /dev/shm/t.dsrc /dev/shm/t_back.fastq ) 2>&1 | grep -v "^$" | tail -40
cmp /dev/shm/t.fastq /dev/shm/t_back.fastq && echo identical
rm -f /dev/shm/t_back.fastq; sleep 5
echo "== 2 handles, passes of 2300"
( time DSRC_HOST_TRACE=1 DSRC_HOST_DEC_INSTANCES=2 dsrc_amd/csrc/dsrc-amd d -t4 -n2300 /dev/shm/t.dsrc /dev/shm/t_back.fastq ) 2>&1 | grep -v "^$" | tail -30
