#!/bin/bash
# chain-alone / in-place time of ablation variants (timing only): tools/gpu/ablate.sh <tag> <variant names...>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export SMST_LIBRARY_ALLOW_MISSING=1
for name in "$@"; do
  if [ "$name" = "product" ]; then unset SMST_LIBRARY; else export SMST_LIBRARY=$ROOT/signalsmith-stretch_amd/variants/$name.so; fi
  timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-self-check ${BENCH_ARGS:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-14s ms/step mean %.3f | alone %s | recurrence in place %.4f ms" % ("$name", d["ms_per_step"], {k: v for k, v in r["kernel_ms_per_step_alone"].items() if v > 0.3}, r["dominant_kernel"]["avg_launch_ms"]))
except Exception as e:
    print("$name failed:", e, open("$OUT/bench_$name.err").read()[-300:])
PY
done
unset SMST_LIBRARY
echo "== aligned trace (sine)"; SMST_LIBRARY=$ROOT/signalsmith-stretch_amd/variants/trace_aligned.so timeout 200 python tools/probes/voc_trace.py sine 2>&1 | grep -v "SMST_LIBRARY is set\|amdgpu.ids" | tee $OUT/trace_aligned.txt
