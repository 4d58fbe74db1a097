#!/bin/bash
# A/B of library variants by per-kernel time (rocprofv3 --kernel-trace --stats), not by the bench's end-to-end figure (+-3 % run to run):
# tools/r05_ab.sh <variant> ... ("built" = the shipped library, others are dsrc_amd/csrc/_var/lib_<variant>.so).  Two workloads per
# variant: one instance alone (512 blocks, default data) and the four-level-quality shards (1800 blocks, four instances).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp DSRC_BENCH_NO_FORMS=1
prof() { local D=gpurun_out/prof_tmp_$$; rm -rf $D; rocprofv3 --kernel-trace --stats -d $D -- "$@" > $D.out 2> $D.err; local F=$(find $D -name "*.db" | head -1); python tools/prof_summary.py "$F" $D.txt > /dev/null; grep -E "k_model|k_part|k_place|k_rc|k_prep|k_index|k_count|k_tag_emit|k_tag_scan|k_assemble" $D.txt | cut -c1-118; rm -rf $D $D.err $D.out $D.txt; }
for v in "$@"; do
  L=$PWD/dsrc_amd/csrc/libdsrc_gpu.so; [ "$v" != built ] && L=$PWD/dsrc_amd/csrc/_var/lib_$v.so
  [ -f $L ] || continue
  export DSRC_GPU_LIB=$L
  echo "== $v: one instance, 512 blocks"; prof python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 2 --warmup 1 --decode-blocks 0
  echo "== $v: four-level qualities, 4 x 450 blocks"; prof python tools/binned_bench.py 1800 4 2
done
