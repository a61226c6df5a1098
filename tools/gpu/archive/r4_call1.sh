#!/bin/bash
# Round 4, first GPU call: GPU tests of the tree's library, then a same-box A/B of the round-3 library (variants/r3.so), the tree's library
# with SMST_NO_ALIGN=1 (round-3 staged producers inside the new build) and the tree's library (line-aligned producers, lag 8).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4_call1
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1
echo "pytest rc $?" >> $OUT/gpu_tests.log
tail -n 6 $OUT/gpu_tests.log
run() { # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-16s %.0f Msamples/s  ms/step mean %.3f median %.3f min %.3f | alone %s | recurrence in place %.4f ms | check %s" % ("$name", d["value"], d["ms_per_step"], r["step_ms"]["median"], r["step_ms"]["min"], {k: v for k, v in r["kernel_ms_per_step_alone"].items() if v > 0.3}, r["dominant_kernel"]["avg_launch_ms"], d.get("self_check")))
except Exception as e:
    print("$name failed:", e, open("$OUT/bench_$name.err").read()[-600:])
PY
}
for round in 1 2; do
  run r3_$round SMST_LIBRARY=$ROOT/signalsmith-stretch_amd/variants/r3.so SMST_LIBRARY_ALLOW_MISSING=1
  run noalign_$round SMST_NO_ALIGN=1
  run aligned_$round SMST_X=0
done
