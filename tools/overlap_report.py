"""Reads a rocprofv3 --kernel-trace CSV and reports how much the kernels of the tile pipeline actually ran side by side:
for every kernel class the total busy time, the time during which some OTHER class was running too, and a timeline of the
last step.  python tools/overlap_report.py gpurun_out/<dir>/.../t_kernel_trace.csv"""
import csv
import sys
from collections import defaultdict


def cls(name):
    for key in ("kVocoder", "kAnalyseTeams", "kSynthEmitTeams", "kEmitProducts", "kSynthTeams", "kAnalyseFast", "kSynthFast", "kEmit", "kCarryFeed", "kCarryOut", "kPredict", "kChain", "kFeed", "kHistory"):
        if key in name:
            return key
    return "other"


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls(r["Kernel_Name"])))
rows.sort()
t0 = rows[0][0]
# sweep: time with exactly the set of running classes
events = []
for s, e, c in rows:
    events.append((s, 1, c))
    events.append((e, -1, c))
events.sort()
running = defaultdict(int)
busy = defaultdict(float)
shared = defaultdict(float)
union = 0.0
last = events[0][0]
for t, delta, c in events:
    dt = t - last
    active = [k for k, v in running.items() if v > 0]
    if active:
        union += dt
    for k in active:
        busy[k] += dt
        if len(active) > 1:
            shared[k] += dt
    running[c] += delta
    last = t
print("span %.3f ms, some kernel running %.3f ms" % ((rows[-1][1] - t0)/1e6, union/1e6))
for k in sorted(busy, key=lambda k: -busy[k]):
    print("  %-14s busy %8.3f ms   alongside another class %8.3f ms (%.0f%%)" % (k, busy[k]/1e6, shared[k]/1e6, 100*shared[k]/max(busy[k], 1)))
print("timeline of the last 40 launches (ms from first launch):")
for s, e, c in rows[-40:]:
    print("  %-14s %9.3f -> %9.3f  (%.3f)" % (c, (s - t0)/1e6, (e - t0)/1e6, (e - s)/1e6))
