"""Shrinks a rocprofv3 counter_collection CSV to one row per (kernel, counter) -- the mean over the dispatches, which is all that
profiles/summarize.py uses -- so that the SQ pass (8 counters x every dispatch x every XCC: tens of MB) fits what gpurun merges back.
usage: python tools/prof/reduce_counters.py <x_counter_collection.csv>     (rewrites the file in place, same columns)"""
import collections
import csv
import sys

path = sys.argv[1]
acc = collections.defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(path, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatches"])
    for (k, c), v in sorted(acc.items()):
        w.writerow([k, c, sum(v)/len(v), len(v)])
