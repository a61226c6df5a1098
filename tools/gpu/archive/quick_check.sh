#!/bin/bash
# the bit-identity tests of the recurrence kernel's producer forms + a same-box A/B: tools/gpu/quick_check.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-qc}
mkdir -p $ROOT/gpurun_out/$TAG
cd $ROOT
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "staged or full_batch or config2 or chunk or half_state or ragged" 2>&1 | tail -5
bash tools/gpu/ab3.sh $TAG "noalign|SMST_NO_ALIGN=1" "aligned|SMST_X=0"
