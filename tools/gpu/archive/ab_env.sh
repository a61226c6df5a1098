#!/bin/bash
# Same-box A/B of ENVIRONMENT settings of one library: tools/gpu/ab_env.sh <tag> "NAME=VAL ..." "NAME=VAL ..." ...   ("-" = no setting)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for round in 1 2; do
i=0
for setting in "$@"; do
  i=$((i+1))
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  env $envs timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_${i}_$round.json 2> $OUT/bench_${i}_$round.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${i}_$round.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-28s run $round: %.0f Msamples/s  ms/step mean %.3f median %.3f min %.3f | alone %s" % ("$setting", d["value"], d["ms_per_step"], r["step_ms"]["median"], r["step_ms"]["min"], {k: v for k, v in r["kernel_ms_per_step_alone"].items() if v > 0.3}))
except Exception as e:
    print("$setting failed:", e, open("$OUT/bench_${i}_$round.err").read()[-400:])
PY
done
done
