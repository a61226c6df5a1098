#!/bin/bash
# same-box A/B of library builds on several bench configs: tools/gpu/ab_configs.sh <tag> "<configs>" <product|variant> ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; CONFIGS=$2; shift 2
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export SMST_LIBRARY_ALLOW_MISSING=1
for c in $CONFIGS; do
for name in "$@"; do
  if [ "$name" = "product" ]; then unset SMST_LIBRARY; else export SMST_LIBRARY=$ROOT/signalsmith-stretch_amd/variants/$name.so; fi
  timeout 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_${name}_c$c.json 2> $OUT/bench_${name}_c$c.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${name}_c$c.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("config %-3s %-8s %.0f Msamples/s  %.2f ms/step | alone %s | check %s" % ("$c", "$name", d["value"], d["ms_per_step"], {k: v for k, v in r["kernel_ms_per_step_alone"].items() if v > 0.3}, (d.get("self_check") or {}).get("ok")))
except Exception as e:
    print("$name config $c failed:", e, open("$OUT/bench_${name}_c$c.err").read()[-300:])
PY
done
done
unset SMST_LIBRARY
