#!/bin/bash
# config 5 with the batch cut into sub-batches of different sizes (the sub-batches alternate between the two tile workspaces: does the
# recurrence of one overlap the bulk kernels of the other?)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_sub
mkdir -p $OUT
cd $ROOT
for sub in ${@:-0 512 256 128}; do
  SMST_SUB_STREAMS=$sub timeout 300 python bench.py --config 5 --steps 4 --warmup 2 --no-cpu-baseline --no-self-check > $OUT/bench5_sub$sub.json 2> $OUT/bench5_sub$sub.err
  python -c "
import json
d = json.loads(open('$OUT/bench5_sub$sub.json').read().strip().splitlines()[-1]); r = d['roofline']
print('sub $sub: %.0f Ms/s  %.3f ms/step  frac %.4f  alone %s' % (d['value'], d['ms_per_step'], r['frac'], r['kernel_ms_per_step_alone']))" || tail -3 $OUT/bench5_sub$sub.err
done
