"""Summarises the TA / TCP counter passes of tools/prof/prof_counters_ta.sh (two counters per rocprofv3 run) into one JSON:
per kernel, every counter averaged per launch, plus the same divided by TCP_GATE_EN1 (= CU-cycles of the launch summed
over the CUs), e.g. TA_TA_BUSY/TCP_GATE_EN1 = fraction of the time the texture-address unit was busy.
usage: python profiles/summarize_ta.py gpurun_out/<tag> profiles/<name>.json ["note"]"""
import collections, csv, glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(src, "p*", "c_counter_collection.csv"))):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smst::", "")
        acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        res[k][c] = sum(v)/len(v)
out = {"note": sys.argv[3] if len(sys.argv) > 3 else "", "kernels": {}}
for k, v in res.items():
    g = v.get("TCP_GATE_EN1_sum")
    out["kernels"][k] = {"per_launch": {c: round(x) for c, x in v.items()},
                         "per_cu_cycle": {c: round(x/g, 4) for c, x in v.items()} if g else None}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for k, v in out["kernels"].items():
    print(k, v["per_cu_cycle"])
