#!/bin/bash
# Quick HBM-traffic check of the headline bench on the GPU box: one FETCH_SIZE and one WRITE_SIZE pass (separate runs),
# printed per kernel.  usage: bash tools/traffic_quick.sh [tag]   (extra bench flags through BENCH_ARGS)
TAG=${1:-tq}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass ${BENCH_ARGS:-}"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f -f csv -- $CMD > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w -f csv -- $CMD > $OUT/write.log 2>&1
python - "$OUT" <<'PY'
import collections, csv, sys, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub, f in (("fetch", "f_counter_collection.csv"), ("write", "w_counter_collection.csv")):
    for r in csv.DictReader(open(os.path.join(sys.argv[1], sub, f))):
        if "smst::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smst::", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    fe = sum(c["FETCH_SIZE"])/max(len(c["FETCH_SIZE"]), 1)*1024*2
    wr = sum(c["WRITE_SIZE"])/max(len(c["WRITE_SIZE"]), 1)*1024
    print("%-34s launches %3d  read %.3f GB  write %.3f GB  total %.3f GB" % (k[:34], len(c["WRITE_SIZE"]), fe/1e9, wr/1e9, (fe + wr)/1e9))
PY
