#!/bin/bash
# bench A/B (noalign vs aligned) + the aligned trace, one box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-abt}
cd $ROOT
bash tools/gpu/ab3.sh $TAG "noalign|SMST_NO_ALIGN=1" "aligned|SMST_X=0"
export SMST_LIBRARY_ALLOW_MISSING=1
echo "== aligned trace (sine)"; SMST_LIBRARY=$ROOT/signalsmith-stretch_amd/variants/trace_aligned.so timeout 200 python tools/probes/voc_trace.py sine 2>&1 | grep -v "SMST_LIBRARY is set\|amdgpu.ids" | tee $ROOT/gpurun_out/$TAG/trace_aligned.txt
