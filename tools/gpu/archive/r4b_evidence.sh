#!/bin/bash
# Round-4 evidence after kSynthEmitTeams, ONE box, final library build: GPU tests, the bench line (CPU baseline + cmd/ binary), the other
# configs, the bench command under rocprofv3 (kernel trace + stats, then the PMC passes, each alone), PMC traffic of config 4b and of
# config 3 on 256 of its streams (the counter collection crashes on 1024-stream runs and on config 5's kernels: EXPERIMENTS.md).
# Part B (`r4b_evidence.sh <tag> B`): batch-size sweep, preset table, real-time quanta.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4b}
PART=${2:-A}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export MASTER_ADDR=127.0.0.1
if [ "$PART" = "A" ]; then
rm -f gpurun_out/parity_instruments.jsonl
timeout 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?" >> $OUT/gpu_tests.log
cp gpurun_out/parity_instruments.jsonl $OUT/ 2>/dev/null
timeout 400 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
for c in 3 4b 5; do timeout 400 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err; done
timeout 400 python bench.py --config 5 --half-state --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_config5_fp16.json 2> $OUT/bench_config5_fp16.err
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --force-dist --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_line_torchrun_n1.json 2> $OUT/torchrun_n1.err
bash tools/prof/prof_counters.sh $TAG/prof > $OUT/prof.log 2>&1
cd $ROOT
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass > $OUT/bench_line_as_profiled.json 2> /dev/null
bash tools/gpu/pmc_configs.sh $TAG 4b > $OUT/pmc_4b.log 2>&1
PMC_BENCH_ARGS="--streams 256" bash tools/gpu/pmc_configs.sh $TAG 3 > $OUT/pmc_3.log 2>&1
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
find $OUT -name "*_counter_collection.csv" -size +16M -delete
du -sh $OUT
tail -n 3 $OUT/gpu_tests.log
python - <<PY
import json
for f in ("bench_line", "bench_config3", "bench_config4b", "bench_config5", "bench_config5_fp16", "bench_line_torchrun_n1"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        print("%-28s %.0f Msamples/s  %.3f ms/step  frac %.4f  n_gpus %s world %s check %s" % (f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["n_gpus"], d.get("dist_world_size"), (d.get("self_check") or {}).get("ok")))
    except Exception as e:
        print(f, "failed:", e)
PY
else
timeout 400 python tools/bench_sweep.py > $OUT/stream_sweep.json 2> $OUT/stream_sweep.err
timeout 400 python tools/bench_presets.py > $OUT/presets.json 2> $OUT/presets.err
timeout 400 python tools/bench_realtime.py > $OUT/realtime_quanta.json 2> $OUT/realtime.err
du -sh $OUT
fi
