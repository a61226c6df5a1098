"""Prints the few numbers of a bench.py JSON line that matter while iterating:
python bench.py ... | python tools/bench_summary.py [label]      or      python tools/bench_summary.py FILE.json [label]"""
import json
import os
import sys

args = sys.argv[1:]
source = open(args.pop(0)) if args and os.path.isfile(args[0]) else sys.stdin
label = args[0] if args else ""
for line in source.read().strip().splitlines():
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline") or {}
    print(label, "%.0f Msamples/s  %.2f ms/step  pipeline_frac %.3f  dominant %s %.3f ms in place  alone %s" % (
        d["value"], d["ms_per_step"], r.get("pipeline_frac") or 0, r.get("kernel"), r.get("avg_launch_ms") or 0, r.get("kernel_ms_per_step_alone")))
