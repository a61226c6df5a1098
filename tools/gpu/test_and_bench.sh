#!/bin/bash
# GPU tests + headline bench (no CPU baseline) of the library in the tree.  usage: tools/gpu/test_and_bench.sh <tag> [extra bench args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-run}; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1
echo "pytest rc $?" >> $OUT/gpu_tests.log
tail -n 15 $OUT/gpu_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.0f Msamples/s  ms/step %.3f  step_ms %s  frac %.4f" % (d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in r["step_ms"].items() if k != "note"}, r["frac"]))
print("alone:", r["kernel_ms_per_step_alone"], " dominant in place: %.4f ms" % r["dominant_kernel"]["avg_launch_ms"])
PY
