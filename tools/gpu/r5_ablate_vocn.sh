#!/bin/bash
# kVocoderN (config 5): timing-only ablations of an -DSMST_EXPERIMENTS build (variants/experiments.so): SMST_DEBUG_MODE=1 the producers skip the
# record arithmetic and its loads (zero records), =2 the recurrence wave only acknowledges its blocks
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_ablate_vocn
mkdir -p $OUT
cd $ROOT
export SMST_LIBRARY_ALLOW_MISSING=1 SMST_LIBRARY=$ROOT/signalsmith-stretch_amd/variants/experiments.so
for mode in ${MODES:-0 1 2}; do
  SMST_DEBUG_MODE=$mode timeout 300 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-self-check > $OUT/mode$mode.json 2> $OUT/mode$mode.err
  python -c "
import json
d = json.loads(open('$OUT/mode$mode.json').read().strip().splitlines()[-1]); r = d['roofline']
print('mode $mode: %.3f ms/step  chain alone %.2f' % (d['ms_per_step'], r['kernel_ms_per_step_alone']['chain']))" || tail -2 $OUT/mode$mode.err
done
