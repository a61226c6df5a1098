# usage: r6_switch_ab.sh <tag> <ENVVAR> "<values>" "<configs>" [pytest -k expression]: GPU tests (a part), then the configs under each value of the switch, two rounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 600 python -m pytest tests -m gpu -x -q -k "${5:-fused or golden or map or gather}" > gpurun_out/$1/gpu_tests.log 2>&1; tail -2 gpurun_out/$1/gpu_tests.log
for c in $4; do for round in 1 2; do for v in $3; do
env $2=$v timeout 600 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/$1/config${c}_${v}_$round.json 2> gpurun_out/$1/config${c}_${v}_$round.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$1/config${c}_${v}_$round.json").read().strip().splitlines()[-1])
print("$2=$v config $c %.3f ms/step frac %.4f chain alone %.2f" % (d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step_alone"]["chain"]))
PY
done; done; done
