#!/bin/bash
# round 5, first GPU call: the GPU suite on the deferred split-computation engine, then same-box A/Bs against the round-4 library
# (signalsmith-stretch_amd/variants/r4.so): headline, config 5 (split mode), the real-time pattern with presetCheaper (split mode)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_call1
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1
echo "pytest rc $?" >> $OUT/gpu_tests.log
tail -n 25 $OUT/gpu_tests.log
export SMST_LIBRARY_ALLOW_MISSING=1
for name in product r4 product r4; do
  if [ "$name" = "product" ]; then unset SMST_LIBRARY; else export SMST_LIBRARY=$ROOT/signalsmith-stretch_amd/variants/$name.so; fi
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "
import json
d = json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); r = d['roofline']
print('$name config2: %.0f Ms/s  %.3f ms/step  frac %.4f  check %s' % (d['value'], d['ms_per_step'], r['frac'], (d.get('self_check') or {}).get('ok')))"
done
for name in product r4; do
  if [ "$name" = "product" ]; then unset SMST_LIBRARY; else export SMST_LIBRARY=$ROOT/signalsmith-stretch_amd/variants/$name.so; fi
  timeout 300 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench5_$name.json 2> $OUT/bench5_$name.err
  python -c "
import json
d = json.loads(open('$OUT/bench5_$name.json').read().strip().splitlines()[-1]); r = d['roofline']
print('$name config5: %.0f Ms/s  %.3f ms/step  frac %.4f  alone %s check %s' % (d['value'], d['ms_per_step'], r['frac'], r['kernel_ms_per_step_alone'], (d.get('self_check') or {}).get('ok')))"
  timeout 300 python tools/bench_realtime.py --preset cheaper --streams 1 256 1024 4096 --quanta 400 > $OUT/realtime_cheaper_$name.json 2> $OUT/realtime_cheaper_$name.err
  python -c "
import json
d = json.loads(open('$OUT/realtime_cheaper_$name.json').read().strip().splitlines()[-1])
print('$name realtime cheaper:', [(r['streams'], r['median_ms'], r['p99_ms']) for r in d['rows']])"
done
unset SMST_LIBRARY
timeout 200 python tools/bench_realtime.py --streams 1 1024 4096 --quanta 400 > $OUT/realtime_default.json 2>&1
tail -c 600 $OUT/realtime_default.json
