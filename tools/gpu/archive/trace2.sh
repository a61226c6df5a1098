#!/bin/bash
# cycle-stamp timelines of the recurrence kernel: aligned producers vs the round-3 staged ones (same box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-trace}
mkdir -p $OUT
cd $ROOT
export SMST_LIBRARY_ALLOW_MISSING=1
for kind in sine noise; do
echo "== aligned ($kind)"; SMST_LIBRARY=$ROOT/signalsmith-stretch_amd/variants/trace_aligned.so timeout 200 python tools/probes/voc_trace.py $kind 2>&1 | grep -v "SMST_LIBRARY is set" | tee -a $OUT/trace_aligned.txt
echo "== staged ($kind)"; SMST_NO_ALIGN=1 SMST_LIBRARY=$ROOT/signalsmith-stretch_amd/variants/trace_staged.so timeout 200 python tools/probes/voc_trace.py $kind 2>&1 | grep -v "SMST_LIBRARY is set" | tee -a $OUT/trace_staged.txt
done
