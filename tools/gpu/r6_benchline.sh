cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
( time timeout 900 python bench.py ) > gpurun_out/$1/bench_line.json 2> gpurun_out/$1/bench_line.err
tail -5 gpurun_out/$1/bench_line.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$1/bench_line.json").read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f frac %.4f host %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["host_ms_per_step"].items() if k != "note"}))
for k, v in (d.get("other_configs") or {}).items():
    print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk != "workload"})
print("realtime", d.get("realtime"))
print("dropin", {k: v for k, v in (d.get("dropin_objects_vs_batch") or {}).items() if k != "what"})
print("cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k in ("value", "cores", "kind")})
PY
