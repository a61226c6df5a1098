#!/bin/bash
# Round-6 evidence on ONE box, final library build.  Part A: GPU tests (-> parity instruments), the bench line (CPU baseline + the cmd/ binary),
# the other configs, the launch-path checks, the headline command under rocprofv3 (kernel trace + stats, then PMC passes, each alone).
# Part B: PMC traffic passes of configs 3 / 4b (256 streams) and 5 (per kernel through --kernel-include-regex, 128 streams: the un-narrowed
# collection crashed inside the profiled process in round 4), batch-size sweep, preset table, real-time quanta (presetDefault and presetCheaper).
# Everything lands in gpurun_out/<tag>/; the judged summaries are copied to profiles/ (profiles/summarize.py, tools/gpu/r5_collect.sh).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r6_final}
PART=${2:-A}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export MASTER_ADDR=127.0.0.1
if [ "$PART" = "A" ]; then
rm -f gpurun_out/parity_instruments.jsonl
timeout 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?" >> $OUT/gpu_tests.log
cp gpurun_out/parity_instruments.jsonl $OUT/ 2>/dev/null
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
for c in 3 4b 5; do timeout 400 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err; done
timeout 400 python bench.py --config 5 --half-state --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_config5_fp16.json 2> $OUT/bench_config5_fp16.err
python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_gpus2_refused.out 2> $OUT/bench_gpus2_refused.err; echo "rc $?" >> $OUT/bench_gpus2_refused.err
timeout 300 python bench.py --gpus 2 --oversubscribe --dist-backend gloo --steps 5 --warmup 2 --no-cpu-baseline > $OUT/oversub_gloo_selflaunched.json 2> $OUT/oversub_gloo_selflaunched.err
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --force-dist --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $OUT/bench_line_torchrun_n1.json 2> $OUT/torchrun_n1.err
bash tools/prof/prof_counters.sh $TAG/prof > $OUT/prof.log 2>&1
cd $ROOT
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass --no-other-configs > $OUT/bench_line_as_profiled.json 2> /dev/null
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
find $OUT -name "*_counter_collection.csv" -size +16M -delete
tail -n 3 $OUT/gpu_tests.log
tail -n 2 $OUT/bench_gpus2_refused.err
else
for c in 3 4b; do
  mkdir -p $OUT/prof_c$c
  ( cd /tmp && export TMPDIR=/tmp
    CMD="python $ROOT/bench.py --config $c --streams 256 --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass --no-self-check"
    timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_c$c/trace -o t -f csv -- $CMD > $OUT/prof_c$c/trace.log 2>&1
    timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_c$c/fetch -o f -f csv -- $CMD > $OUT/prof_c$c/fetch.log 2>&1
    timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_c$c/write -o w -f csv -- $CMD > $OUT/prof_c$c/write.log 2>&1 )
done
# config 5: one counter, one kernel class per pass
mkdir -p $OUT/prof_c5
( cd /tmp && export TMPDIR=/tmp
  CMD="python $ROOT/bench.py --config 5 --streams 128 --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass --no-self-check"
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_c5/trace -o t -f csv -- $CMD > $OUT/prof_c5/trace.log 2>&1
  for K in kVocoderN kAnalyse kSynth kEmit kFeed kCarry kHistory kEnergy kPending; do
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "$K" -d $OUT/prof_c5/${K}_$C -o c -f csv -- $CMD > $OUT/prof_c5/${K}_$C.log 2>&1
      echo "$K $C rc $?" >> $OUT/prof_c5/outcomes.txt
    done
  done )
timeout 400 python tools/bench_sweep.py > $OUT/stream_sweep.json 2> $OUT/stream_sweep.err
timeout 400 python tools/bench_presets.py > $OUT/presets.json 2> $OUT/presets.err
timeout 400 python tools/bench_realtime.py > $OUT/realtime_quanta.json 2> $OUT/realtime.err
timeout 400 python tools/bench_realtime.py --preset cheaper > $OUT/realtime_quanta_cheaper_split.json 2> $OUT/realtime_cheaper.err
# what travels back is bounded (64 MiB): counter files shrink to one row per (kernel, counter), kernel traces of the counter passes go
for f in $(find $OUT -name "*_counter_collection.csv"); do python3 $ROOT/tools/prof/reduce_counters.py $f; done
find $OUT -path "*prof_c*" -name "*_kernel_trace.csv" -delete
find $OUT -name "*_agent_info.csv" -delete
cat $OUT/prof_c5/outcomes.txt
fi
du -sh $OUT
python - <<PY
import json
for f in ("bench_line", "bench_config3", "bench_config4b", "bench_config5", "bench_config5_fp16", "bench_line_torchrun_n1", "oversub_gloo_selflaunched"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        print("%-28s %.0f Msamples/s  %.3f ms/step  frac %.4f  n_gpus %s world %s check %s" % (f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["n_gpus"], d.get("dist_world_size"), (d.get("self_check") or {}).get("ok")))
    except Exception as e:
        pass
PY
