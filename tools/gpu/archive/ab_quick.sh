#!/bin/bash
# One short bench run per library, stand-alone kernel times only: tools/gpu/ab_quick.sh <tag> <lib> ...   ("product" = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = "product" ]; then unset SMST_LIBRARY; else export SMST_LIBRARY=$ROOT/$lib; fi
  timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_${name}.json 2> $OUT/bench_${name}.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${name}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-22s %.3f ms/step | alone %s" % ("$name", d["ms_per_step"], {k: v for k, v in r["kernel_ms_per_step_alone"].items() if v > 0.3}))
except Exception as e:
    print("$name failed:", e, open("$OUT/bench_${name}.err").read()[-400:])
PY
done
unset SMST_LIBRARY
