#!/bin/bash
# Same-box A/B of ONE library under two environments: tools/gpu/r6_env_ab.sh <tag> "<env A>" "<env B>" [bench args]
# e.g.  r6_env_ab.sh cont "" "SMST_NO_CONTINUOUS=1"      (an empty string = the default environment)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; ENVA=$2; ENVB=$3; shift 3
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for round in 1 2; do
for which in A B; do
  if [ $which = A ]; then E="$ENVA"; else E="$ENVB"; fi
  env $E timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench_${which}_$round.json 2> $OUT/bench_${which}_$round.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${which}_$round.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-28s run $round: %.0f Msamples/s  ms/step mean %.3f median %.3f min %.3f | alone %s | recurrence in place %.4f ms | self_check %s" % ("[$E]", d["value"], d["ms_per_step"], r["step_ms"]["median"], r["step_ms"]["min"], {k: v for k, v in r["kernel_ms_per_step_alone"].items() if v > 0.3}, r["dominant_kernel"]["avg_launch_ms"], d.get("self_check", {}).get("rel_rms")))
except Exception as e:
    print("[$E] failed:", e, open("$OUT/bench_${which}_$round.err").read()[-600:])
PY
done
done
