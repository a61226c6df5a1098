#!/bin/bash
# bit-identity of kSynthEmitTeams against kSynthTeams + kEmit, then a same-box A/B: tools/gpu/synth_emit_check.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-se}
mkdir -p $ROOT/gpurun_out/$TAG
cd $ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "synth_emit or full_batch or config2 or chunk or half_state or ragged or fft_teams" 2>&1 | tail -5
bash tools/gpu/ab3.sh $TAG "two_kernels|SMST_SYNTH_EMIT=0" "synth_emit|SMST_X=0"
