#!/bin/bash
# `dsrc-amd d` on the 38.5 GB set: first passes of the workers started apart
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
for cfg in "0 3 0" "800 3 0" "1500 3 0" "0 3 0" "800 3 0" "1500 3 0" "0 3 0" "800 3 0" "1500 3 0"; do
  set -- $cfg
  rm -f /dev/shm/t_back.fastq; sleep 5
  echo "== stagger $1 ms, $2 handles, -n$3"
  NARG=""; [ $3 != 0 ] && NARG="-n$3"
  ( time DSRC_HOST_DEC_STAGGER_MS=$1 DSRC_HOST_DEC_INSTANCES=$2 dsrc_amd/csrc/dsrc-amd d -t4 $NARG /dev/shm/t.dsrc /dev/shm/t_back.fastq ) 2>&1 | grep real
done
cmp /dev/shm/t.fastq /dev/shm/t_back.fastq && echo identical
