#!/bin/bash
# does the number of hardware queues explain the slower step once an RCCL communicator exists?  (HIP maps streams onto GPU_MAX_HW_QUEUES = 4 queues)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export MASTER_ADDR=127.0.0.1
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-34s %.0f Ms/s  mean %.3f  median %.3f  min %.3f max %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], r["step_ms"]["median"], r["step_ms"]["min"], r["step_ms"]["max"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
A="--steps 20 --warmup 3 --no-cpu-baseline"
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2961$q bench.py --gpus 1 --force-dist $A > /tmp/b.json 2>/tmp/b.err; show "torchrun force-dist HWQ=$q" /tmp/b.json
done
tail -3 /tmp/b.err
