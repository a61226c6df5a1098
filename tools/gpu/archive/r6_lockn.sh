# usage: r6_lockn.sh <tag>: the whole GPU suite, then config 5 (8 channels) and the headline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$1/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/$1/gpu_tests.log
tail -5 gpurun_out/$1/gpu_tests.log
for i in 1 2; do
timeout 600 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/$1/config5_$i.json 2> gpurun_out/$1/config5_$i.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$1/config5_$i.json").read().strip().splitlines()[-1])
print("config5 value %.0f ms/step %.3f frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/$1/bench.json 2> gpurun_out/$1/bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$1/bench.json").read().strip().splitlines()[-1])
print("headline value %.0f ms/step %.3f frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
