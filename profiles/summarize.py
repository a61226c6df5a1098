"""Turns the rocprofv3 CSVs collected by tools/prof/prof_counters.sh (kernel trace + stats, FETCH_SIZE / WRITE_SIZE / SQ
passes, each in its own run) into the small summaries committed under profiles/.
usage: python profiles/summarize.py gpurun_out/<tag> r1 [steps profiled, default 3] [bench config, default 2: traffic_latest.json;
       another config N writes traffic_latest_configN.json, which `bench.py --config N` quotes] [streams the passes ran on, if a subset]
HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts half of the bytes of a coalesced stream, so traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024.
(Calibration in this repo: kAnalyse writes exactly S*hops*C*2*24576 B per launch and WRITE_SIZE reports that number;
kPredictA reads one 24576-B row per channel-hop and FETCH_SIZE reports half of it.)"""
import collections
import csv
import json
import os
import sys


def short(name):
    name = name.split("(")[0]
    name = name.replace("void ", "").replace("smst::", "")
    return name.split("<")[0] + ("<" + name.split("<")[1] if "<" in name else "")


def main():
    src, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    out = {}
    stats = os.path.join(src, "trace", "t_kernel_stats.csv")
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(here, tag + "_kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for r in rows:
            if "smst::" not in r["Name"]:
                continue
            f.write("%s,%s,%.3f,%.1f,%s\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
            out.setdefault(short(r["Name"]), {})["avg_us"] = float(r["AverageNs"])/1e3
            out[short(r["Name"])]["calls"] = int(r["Calls"])
    for sub, fname in (("fetch", "f_counter_collection.csv"), ("write", "w_counter_collection.csv"), ("sq", "s_counter_collection.csv")):
        path = os.path.join(src, sub, fname)
        if not os.path.exists(path):
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            if "smst::" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, counters in acc.items():
            for c, vals in counters.items():
                out.setdefault(k, {})[c] = sum(vals)/len(vals)
    traffic, best = {}, {}
    for k, v in out.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes_per_launch"] = (2*v["FETCH_SIZE"] + v["WRITE_SIZE"])*1024
            if "avg_us" in v:
                v["hbm_GBps"] = v["hbm_bytes_per_launch"]/(v["avg_us"]*1e-6)/1e9
            base = k.split("<")[0] # template variants share a key: keep the one launched most often
            if base not in traffic or v.get("calls", 0) > best.get(base, 0):
                traffic[base] = v["hbm_bytes_per_launch"]
                best[base] = v.get("calls", 0)
    json.dump(out, open(os.path.join(here, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    # what bench.py quotes as `traffic`: only valid for the library build the passes ran on (bench.py compares the hash)
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3  # tools/prof/prof_counters.sh: --steps 2 --warmup 1
    per_step = sum(v["hbm_bytes_per_launch"]*v.get("calls", 0) for v in out.values() if "hbm_bytes_per_launch" in v)/steps
    import hashlib
    lib = os.path.join(os.path.dirname(here), "signalsmith-stretch_amd", "libsmst_hip.so")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    config = sys.argv[4] if len(sys.argv) > 4 else "2"
    streams = int(sys.argv[5]) if len(sys.argv) > 5 else None  # passes that ran on a subset of the config's streams (bench.py --streams)
    latest = {"library_sha16": sha, "bytes_per_step": per_step, "kernels": traffic, "steps_profiled": steps, "config": config, "streams": streams,
              "command": "bench.py%s --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass (tools/prof/prof_counters.sh)" % ("" if config == "2" else " --config " + config),
              "summary": tag + "_pmc_summary.json"}
    json.dump(latest, open(os.path.join(here, "traffic_latest.json" if config == "2" else "traffic_latest_config%s.json" % config), "w"), indent=1, sort_keys=True)
    for k in sorted(out, key=lambda k: -out[k].get("avg_us", 0)*out[k].get("calls", 0)):
        v = out[k]
        print("%-28s calls %4d avg %8.1f us  hbm %8.1f MB/launch  %6.0f GB/s" % (k, v.get("calls", 0), v.get("avg_us", 0),
              v.get("hbm_bytes_per_launch", 0)/1e6, v.get("hbm_GBps", 0)))


if __name__ == "__main__":
    main()
