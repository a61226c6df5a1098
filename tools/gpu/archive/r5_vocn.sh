#!/bin/bash
# kVocoderN A/B on one box: producer passes 8 rows x 8 steps (SMST_VOCN_WIDE=1, default) against 16 rows x 4 (=0); config 5 and the fused == unfused tests
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_vocn
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -k "fused or config5 or eight or channels" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for round in 1 2; do
for w in 1 0; do
  SMST_VOCN_WIDE=$w timeout 300 python bench.py --config 5 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench5_wide${w}_$round.json 2> $OUT/bench5_wide${w}_$round.err
  python -c "
import json
d = json.loads(open('$OUT/bench5_wide${w}_$round.json').read().strip().splitlines()[-1]); r = d['roofline']
print('wide $w: %.0f Ms/s  %.3f ms/step  frac %.4f  alone %s  in place %.3f ms/launch check %s' % (d['value'], d['ms_per_step'], r['frac'], r['kernel_ms_per_step_alone'], r['dominant_kernel']['avg_launch_ms'], (d.get('self_check') or {}).get('ok')))" || tail -3 $OUT/bench5_wide${w}_$round.err
done
done
