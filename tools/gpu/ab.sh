#!/bin/bash
# Same-box A/B of library builds: tools/gpu/ab.sh <tag> <lib1> <lib2> ...   (paths relative to the repo; "product" = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for round in 1 2; do
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = "product" ]; then unset SMST_LIBRARY; else export SMST_LIBRARY=$ROOT/$lib; fi
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_${name}_$round.json 2> $OUT/bench_${name}_$round.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${name}_$round.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-12s run $round: %.0f Msamples/s  ms/step mean %.3f median %.3f min %.3f | alone %s | recurrence in place %.4f ms" % ("$name", d["value"], d["ms_per_step"], r["step_ms"]["median"], r["step_ms"]["min"], {k: v for k, v in r["kernel_ms_per_step_alone"].items() if v > 0.3}, r["dominant_kernel"]["avg_launch_ms"]))
except Exception as e:
    print("$name failed:", e, open("$OUT/bench_${name}_$round.err").read()[-400:])
PY
done
done
unset SMST_LIBRARY
