#!/bin/bash
# same-box A/B of the library in the tree against variants (signalsmith-stretch_amd/variants/<name>.so: tools/probes/build_variant.sh) on any bench config.
# usage: tools/gpu/ab_variant.sh <tag> "<bench args>" <variant> [variant ...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; ARGS=$2; shift 2
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export SMST_LIBRARY_ALLOW_MISSING=1
for round in 1 2; do
for name in product "$@"; do
  if [ "$name" = "product" ]; then unset SMST_LIBRARY; else export SMST_LIBRARY=$ROOT/signalsmith-stretch_amd/variants/$name.so; fi
  timeout 300 python bench.py $ARGS --no-cpu-baseline --no-self-check > $OUT/${name}_$round.json 2> $OUT/${name}_$round.err
  python -c "
import json
d = json.loads(open('$OUT/${name}_$round.json').read().strip().splitlines()[-1]); r = d['roofline']
print('%-12s %.3f ms/step  alone %s' % ('${name}_$round', d['ms_per_step'], {k: v for k, v in r['kernel_ms_per_step_alone'].items() if v > 0.3}))" || tail -2 $OUT/${name}_$round.err
done
done
unset SMST_LIBRARY
