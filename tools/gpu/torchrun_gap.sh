#!/bin/bash
# why is the bench slower under torch.distributed.run at N = 1?  plain / OMP_NUM_THREADS=1 / torchrun, 20 steps each
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export MASTER_ADDR=127.0.0.1
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%-26s %.0f Ms/s  mean %.3f  median %.3f  min %.3f max %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], r["step_ms"]["median"], r["step_ms"]["min"], r["step_ms"]["max"]))
PY
}
A="--steps 20 --warmup 3 --no-cpu-baseline"
python bench.py $A > /tmp/a.json 2>/dev/null; show plain /tmp/a.json
OMP_NUM_THREADS=1 python bench.py $A > /tmp/b.json 2>/dev/null; show "plain OMP=1" /tmp/b.json
python bench.py $A --force-dist > /tmp/c.json 2>/tmp/c.err; show "plain --force-dist" /tmp/c.json || tail -3 /tmp/c.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 $A > /tmp/d.json 2>/dev/null; show "torchrun (no force)" /tmp/d.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29614 bench.py --gpus 1 --force-dist $A > /tmp/e.json 2>/dev/null; show "torchrun --force-dist" /tmp/e.json
OMP_NUM_THREADS=8 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29615 bench.py --gpus 1 --force-dist $A > /tmp/f.json 2>/dev/null; show "torchrun OMP=8 force" /tmp/f.json
