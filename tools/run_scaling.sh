#!/bin/bash
# The 1/2/4/8-GPU curve of the headline benchmark on ONE node through bench.py's own launch path: `python bench.py --gpus N` starts its N
# ranks itself exactly as the round driver does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...),
# refuses when the node has fewer devices, pins every rank's host scheduler to the CPUs of its GPU's NUMA node and says on stderr and in
# the line where each rank ran (`config.ranks`).  One process per GPU, batch sharded by stream (256 stereo streams per GPU, weak scaling),
# no collective on the data path -- RCCL carries only the barrier and the max-over-ranks clock.
# Usage: tools/run_scaling.sh [config] [out_dir]   (config: 2 (default), 3, 4b, 5)
CONFIG=${1:-2}
OUT=${2:-gpurun_out/scaling}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 1 2 4 8; do
  EXTRA=""; [ "$N" != 1 ] && EXTRA="--no-cpu-baseline"
  MASTER_PORT=$((29600 + N)) python "$ROOT/bench.py" --gpus "$N" --config "$CONFIG" --steps 5 --warmup 2 $EXTRA > "$OUT/bench_c${CONFIG}_n${N}.json" 2> "$OUT/bench_c${CONFIG}_n${N}.err" \
    || { echo "N=$N: $(tail -n 1 "$OUT/bench_c${CONFIG}_n${N}.err")"; rm -f "$OUT/bench_c${CONFIG}_n${N}.json"; }
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
    ranks = d.get("config", {}).get("ranks") or []
    print("N=%d  %.0f Msamples/s  x%.2f vs N=1  (%.2f ms/step; dist_world_size %s; ranks on %s)" % (
        n, d["value"], d["value"]/base, d["ms_per_step"], d.get("dist_world_size"), [(r.get("pci"), r.get("numa_node"), r.get("cpus")) for r in ranks]))
PY
