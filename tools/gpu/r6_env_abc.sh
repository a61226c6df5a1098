#!/bin/bash
# Same-box comparison of ONE library under several environments: r6_env_abc.sh <tag> "<env 1>" "<env 2>" ... (an empty string = the default environment); two rounds
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for round in 1 2; do
i=0
for E in "$@"; do
  i=$((i+1))
  env $E timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs ${BENCH_ARGS:-} > $OUT/bench_${i}_$round.json 2> $OUT/bench_${i}_$round.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${i}_$round.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-50s run $round: ms/step mean %.3f median %.3f min %.3f | alone %s | recurrence in place %.4f ms" % ("[$E]", d["ms_per_step"], r["step_ms"]["median"], r["step_ms"]["min"], {k: v for k, v in r["kernel_ms_per_step_alone"].items() if v > 0.3}, r["dominant_kernel"]["avg_launch_ms"]))
except Exception as e:
    print("[$E] failed:", e, open("$OUT/bench_${i}_$round.err").read()[-600:])
PY
done
done
