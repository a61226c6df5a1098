#!/bin/bash
# round 5: the parity instruments on the GPU (teacher-forced with per-bin arg-max masks, free-running subsets with their informative horizons)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_parity
mkdir -p $OUT
cd $ROOT
rm -f gpurun_out/parity_instruments.jsonl
timeout 1200 python -m pytest tests/test_parity_gpu.py -q -k "${1:-teacher_forced or subset or config5 or config2}" > $OUT/tests.log 2>&1
echo "pytest rc $?" >> $OUT/tests.log
tail -n 40 $OUT/tests.log
cp gpurun_out/parity_instruments.jsonl $OUT/ 2>/dev/null
