#!/bin/bash
# The 1/2/4/8-GPU curve of the headline benchmark on ONE node, launched exactly as the round driver launches it: one process
# per GPU, batch sharded by stream (256 stereo streams per GPU, weak scaling), no collective on the data path -- RCCL carries
# only the barrier and the max-over-ranks clock.  Each rank pins its host scheduler to the CPUs of its GPU's NUMA node
# (bench.py: pin_rank_to_numa_node).  Usage: tools/run_scaling.sh [config] [out_dir]   (config: 2 (default), 3, 4b, 5)
CONFIG=${1:-2}
OUT=${2:-gpurun_out/scaling}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 1 2 4 8; do
  if [ "$N" = 1 ]; then
    python "$ROOT/bench.py" --gpus 1 --config "$CONFIG" --steps 5 --warmup 2 > "$OUT/bench_c${CONFIG}_n1.json"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      "$ROOT/bench.py" --gpus "$N" --config "$CONFIG" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_c${CONFIG}_n${N}.json" || echo "N=$N failed (fewer GPUs on this node?)"
  fi
done
python - "$OUT" "$CONFIG" <<'PY'
import json, sys, os
out, cfg = sys.argv[1], sys.argv[2]
base = None
for n in (1, 2, 4, 8):
    p = os.path.join(out, "bench_c%s_n%d.json" % (cfg, n))
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        continue
    base = base or d["value"]
    print("N=%d  %.0f Msamples/s  x%.2f vs N=1  (%.2f ms/step)" % (n, d["value"], d["value"]/base, d["ms_per_step"]))
PY
