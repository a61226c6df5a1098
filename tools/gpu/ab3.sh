#!/bin/bash
# Same-box A/B/C of the headline bench: tools/gpu/ab3.sh <tag> "name|ENV=VAL ENV=VAL" ...   (two rounds, 20 steps each)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for round in 1 2; do
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  env $envs timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_${name}_$round.json 2> $OUT/bench_${name}_$round.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${name}_$round.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    c = d.get("self_check") or {}
    print("%-14s %.0f Msamples/s  ms/step mean %.3f median %.3f min %.3f | alone %s | recurrence in place %.4f ms | check %s %.2e" % ("${name}_$round", d["value"], d["ms_per_step"], r["step_ms"]["median"], r["step_ms"]["min"], {k: v for k, v in r["kernel_ms_per_step_alone"].items() if v > 0.3}, r["dominant_kernel"]["avg_launch_ms"], c.get("ok"), c.get("rel_rms_first_second", float("nan"))))
except Exception as e:
    print("$name failed:", e, open("$OUT/bench_${name}_$round.err").read()[-600:])
PY
done
done
