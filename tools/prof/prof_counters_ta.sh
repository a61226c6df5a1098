#!/bin/bash
# Memory-pipeline counters (TA / TCP) of the fused vocoder kernel, two counters per pass (the TA/TCP blocks have few slots).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${PMCTAG:-pmc2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --kernel-include-regex "${KREGEX:-kVocoder}" --pmc $set -d $OUT/p$i -o c -f csv -- $CMD > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
