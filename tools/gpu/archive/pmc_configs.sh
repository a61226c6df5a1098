#!/bin/bash
# PMC traffic passes (FETCH_SIZE, WRITE_SIZE, each alone, --kernel-trace only) + kernel stats of the given bench configs.
# usage: tools/gpu/pmc_configs.sh <tag> <config> ...     SMST_WORKSPACE_GIB applies (rocprofv3's counter collection crashed on the
# 1024-stream configs with the default, single-sub-batch workspace; the same kernels in two or three sub-batches move the same bytes)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
for c in "$@"; do
  mkdir -p $OUT/prof_c$c
  ( cd /tmp && export TMPDIR=/tmp
    CMD="python $ROOT/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass --no-self-check ${PMC_BENCH_ARGS:-}"
    timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_c$c/trace -o t -f csv -- $CMD > $OUT/prof_c$c/trace.log 2>&1
    timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_c$c/fetch -o f -f csv -- $CMD > $OUT/prof_c$c/fetch.log 2>&1
    timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_c$c/write -o w -f csv -- $CMD > $OUT/prof_c$c/write.log 2>&1 )
  ls $OUT/prof_c$c
done
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
find $OUT -name "*_counter_collection.csv" -size +16M -delete
