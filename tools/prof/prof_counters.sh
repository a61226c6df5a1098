#!/bin/bash
# Collect rocprofv3 evidence on the GPU box: kernel trace + stats, then PMC counters in separate passes
# (never combined with trace domains other than --kernel-trace).  Output under gpurun_out/<tag>/.
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass --no-other-configs ${BENCH_ARGS:-}"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -f csv -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f -f csv -- $CMD > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w -f csv -- $CMD > $OUT/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $OUT/sq -o s -f csv -- $CMD > $OUT/sq.log 2>&1
for f in $OUT/fetch/f_counter_collection.csv $OUT/write/w_counter_collection.csv $OUT/sq/s_counter_collection.csv; do [ -f $f ] && python3 $ROOT/tools/prof/reduce_counters.py $f; done
find $OUT -name "*.csv" | head -20
