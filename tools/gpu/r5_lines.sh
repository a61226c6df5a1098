#!/bin/bash
# the bench lines once profiles/traffic_latest*.json carry the hash of the library in the tree (bench.py quotes the PMC traffic only then)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-r5_lines}
mkdir -p $OUT
cd $ROOT
timeout 500 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
for c in 3 4b 5; do timeout 400 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err; done
python - <<PY
import json
for f in ("bench_line", "bench_config3", "bench_config4b", "bench_config5"):
    d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-16s %.0f Msamples/s  %.3f ms/step  frac %.4f  traffic %s  (%s)" % (f, d["value"], d["ms_per_step"], r["frac"], r.get("traffic"), str(r.get("traffic_source"))[:90]))
PY
