#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) of one 512-block compression batch.
# usage: tools/r04_pmc.sh <out.txt> [env assignments...]
OUT=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
env "$@" rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -- python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 1 --warmup 0 --decode-blocks 0 > /dev/null 2> gpurun_out/pmc_f.err
env "$@" rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -- python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 1 --warmup 0 --decode-blocks 0 > /dev/null 2> gpurun_out/pmc_w.err
python tools/pmc_summary.py $(find gpurun_out/pmc_f gpurun_out/pmc_w -name "*.db") > $OUT
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_f.err gpurun_out/pmc_w.err
head -30 $OUT
