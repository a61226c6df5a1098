# usage: r6_vocn_ab.sh <tag> "<modes>": config 5 with the product library (twice), then the experiments variant in the given SMST_DEBUG_MODEs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 300 python -m pytest tests -m gpu -x -q -k "channel or fused or golden or config5" > gpurun_out/$1/gpu_tests.log 2>&1; tail -2 gpurun_out/$1/gpu_tests.log
for i in 1 2; do
timeout 600 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/$1/config5_$i.json 2> gpurun_out/$1/config5_$i.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$1/config5_$i.json").read().strip().splitlines()[-1])
print("product config5 %.3f ms/step frac %.4f chain alone %.2f" % (d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step_alone"]["chain"]))
PY
done
MODES="$2" bash tools/gpu/r5_ablate_vocn.sh
