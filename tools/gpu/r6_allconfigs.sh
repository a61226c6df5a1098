# usage: r6_allconfigs.sh <tag>: the whole GPU suite, then configs 5 / 3 / 4b and the headline (short runs, no CPU baseline)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$1/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/$1/gpu_tests.log
tail -4 gpurun_out/$1/gpu_tests.log
for c in 5 3 4b 2; do
timeout 600 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/$1/config$c.json 2> gpurun_out/$1/config$c.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$1/config$c.json").read().strip().splitlines()[-1])
print("config $c: %.0f Msamples/s %.3f ms/step frac %.4f alone %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: v for k, v in d["roofline"]["kernel_ms_per_step_alone"].items() if v > 0.3}))
PY
done
