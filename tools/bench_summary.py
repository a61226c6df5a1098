"""Prints the few numbers of a bench.py JSON line that matter while iterating: python bench.py ... | python tools/bench_summary.py [label]"""
import json
import sys

label = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin.read().strip().splitlines():
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline") or {}
    print(label, "%.0f Msamples/s  %.2f ms/step  pipeline_frac %.3f  dominant %s %.3f ms in place  alone %s" % (
        d["value"], d["ms_per_step"], r.get("pipeline_frac") or 0, r.get("kernel"), r.get("avg_launch_ms") or 0, r.get("kernel_ms_per_step_alone")))
