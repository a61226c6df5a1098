# usage: r6_quick.sh <tag> [bench args]: one headline bench without the extra blocks, printing the host-time block
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1; TAG=$1; shift
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs "$@" > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
print("host", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["host_ms_per_step"].items() if k != "note"})
PY
./signalsmith-stretch_amd/bench_dropin 64 1 6
