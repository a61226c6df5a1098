# usage: r6_env_ab5.sh <tag> <ENVVAR> "<values>": multi-channel GPU tests, then config 5 under each value of the switch, two rounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 400 python -m pytest tests -m gpu -x -q -k "channel or fused or golden or config5" > gpurun_out/$1/gpu_tests.log 2>&1; tail -2 gpurun_out/$1/gpu_tests.log
for round in 1 2; do for v in $3; do
env $2=$v timeout 600 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/$1/config5_${v}_$round.json 2> gpurun_out/$1/config5_${v}_$round.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$1/config5_${v}_$round.json").read().strip().splitlines()[-1])
print("$2=$v config5 %.3f ms/step frac %.4f chain alone %.2f" % (d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step_alone"]["chain"]))
PY
done; done
