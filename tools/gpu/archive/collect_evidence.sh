#!/bin/bash
# After tools/gpu/final_evidence.sh <tag> has run on the GPU box: summarise the PMC passes (writes profiles/traffic_latest.json with the
# library hash) and copy the judged files from gpurun_out/<tag>/ into profiles/ under the round's prefix.
# usage: tools/gpu/collect_evidence.sh <tag> <prefix>      e.g.  collect_evidence.sh r3final3 r3
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=$1; PFX=$2
cd $ROOT
python profiles/summarize.py gpurun_out/$TAG/prof $PFX 3 | head -5
grep -h '^{"metric"' gpurun_out/$TAG/prof/trace.log | tail -1 > profiles/${PFX}_bench_line_under_rocprofv3.json
for f in bench_config3 bench_config4b bench_config5 bench_config5_fp16 bench_line_torchrun_n1 stream_sweep presets realtime_quanta; do cp gpurun_out/$TAG/$f.json profiles/${PFX}_$f.json; done
cp gpurun_out/$TAG/gpu_tests.log profiles/${PFX}_gpu_tests.log
cp gpurun_out/$TAG/parity_instruments.jsonl profiles/${PFX}_parity_instruments.jsonl
cp gpurun_out/$TAG/prof/trace/t_kernel_stats.csv profiles/${PFX}_rocprofv3_kernel_stats_raw.csv
cp gpurun_out/$TAG/oversub_gloo.json profiles/${PFX}_oversubscribe_2ranks_final_build.json
echo "now re-run bench.py on the GPU (traffic_latest.json carries this build's hash) and copy the line to profiles/${PFX}_bench_line.json"
