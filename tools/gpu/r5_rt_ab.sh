#!/bin/bash
# real-time pattern: the library in the tree against variants/base.so (the previous commit), same box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_rt_ab
mkdir -p $OUT
cd $ROOT
export SMST_LIBRARY_ALLOW_MISSING=1
for round in 1 2; do
for name in product base; do
  if [ "$name" = "product" ]; then unset SMST_LIBRARY; else export SMST_LIBRARY=$ROOT/signalsmith-stretch_amd/variants/$name.so; fi
  for preset in default cheaper; do
    timeout 300 python tools/bench_realtime.py --preset $preset --streams 256 1024 4096 --quanta 500 > $OUT/${name}_${preset}_$round.json 2> $OUT/${name}_${preset}_$round.err
    python -c "
import json
d = json.loads(open('$OUT/${name}_${preset}_$round.json').read().strip().splitlines()[-1])
print('$name $preset:', [(r['streams'], r['median_ms'], r['p99_ms']) for r in d['rows']])"
  done
done
done
