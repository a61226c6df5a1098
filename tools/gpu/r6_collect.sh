#!/bin/bash
# gpurun_out/<tag>/ (tools/gpu/r6_evidence.sh, parts A and B) -> the summaries committed under profiles/ (prefix r6_)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=${1:-r6_final}
SRC=$ROOT/gpurun_out/$TAG
P=$ROOT/profiles
cd $ROOT
cp $SRC/bench_line.json $P/r6_bench_line.json
cp $SRC/bench_line_as_profiled.json $P/r6_bench_line_under_rocprofv3.json 2>/dev/null
for c in 3 4b 5; do cp $SRC/bench_config$c.json $P/r6_bench_config$c.json; done
cp $SRC/bench_config5_fp16.json $P/r6_bench_config5_fp16.json
cp $SRC/bench_line_torchrun_n1.json $P/r6_bench_line_torchrun_n1.json
cp $SRC/oversub_gloo_selflaunched.json $P/r6_oversub_gloo_selflaunched.json
cat $SRC/bench_gpus2_refused.out $SRC/bench_gpus2_refused.err > $P/r6_bench_gpus2_refused.txt
cp $SRC/gpu_tests.log $P/r6_gpu_tests.log
cp $SRC/parity_instruments.jsonl $P/r6_parity_instruments.jsonl
cp $SRC/stream_sweep.json $P/r6_stream_sweep.json
cp $SRC/presets.json $P/r6_presets.json
cp $SRC/realtime_quanta.json $P/r6_realtime_quanta.json
cp $SRC/realtime_quanta_cheaper_split.json $P/r6_realtime_quanta_cheaper_split.json
cp $SRC/prof/trace/t_kernel_stats.csv $P/r6_rocprofv3_kernel_stats_raw.csv
python profiles/summarize.py $SRC/prof r6 3 2
python profiles/summarize.py $SRC/prof_c3 r6_config3 3 3 256
python profiles/summarize.py $SRC/prof_c4b r6_config4b 3 4b 256
# config 5: the per-kernel passes joined into one fetch / one write file
mkdir -p $SRC/prof_c5/fetch $SRC/prof_c5/write
python - "$SRC/prof_c5" <<'PY'
import csv, glob, os, sys
src = sys.argv[1]
for counter, sub, name in (("FETCH_SIZE", "fetch", "f"), ("WRITE_SIZE", "write", "w")):
    rows = []
    for path in sorted(glob.glob(os.path.join(src, "*_" + counter, "*counter_collection.csv"))):
        rows += list(csv.DictReader(open(path)))
    with open(os.path.join(src, sub, name + "_counter_collection.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatches"])
        w.writeheader()
        for r in rows:
            w.writerow({k: r[k] for k in w.fieldnames})
PY
python profiles/summarize.py $SRC/prof_c5 r6_config5 3 5 128
cp $SRC/prof_c5/outcomes.txt $P/r6_config5_pmc_outcomes_final.txt
