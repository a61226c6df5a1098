#!/bin/bash
# Round-3 GPU call 1: instruction-cost probes, same-box baseline of the round-2 library, batch-size sweep, and the N>1 launch path
# with two ranks on the one GPU.  Everything lands in gpurun_out/r3a/.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3a
mkdir -p $OUT
cd $ROOT
export MASTER_ADDR=127.0.0.1
timeout 120 tools/probes/valu_rate > $OUT/valu_rate.txt 2>&1
timeout 120 tools/probes/wg_placement > $OUT/wg_placement.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_base.json 2> $OUT/bench_base.err
timeout 400 python tools/bench_sweep.py > $OUT/stream_sweep.json 2> $OUT/stream_sweep.err
# two ranks on ONE device: RCCL first, gloo if RCCL refuses duplicate devices
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --oversubscribe \
    --steps 5 --warmup 2 --no-cpu-baseline --dump-output $OUT/oversub_nccl > $OUT/oversub_nccl.json 2> $OUT/oversub_nccl.err
echo "nccl rc $?" > $OUT/oversub_rc.txt
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --oversubscribe --dist-backend gloo \
    --steps 5 --warmup 2 --no-cpu-baseline --dump-output $OUT/oversub_gloo > $OUT/oversub_gloo.json 2> $OUT/oversub_gloo.err
echo "gloo rc $?" >> $OUT/oversub_rc.txt
for r in 0 1; do
  timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-serial-pass --as-rank $r --dump-output $OUT/single > $OUT/single_rank$r.json 2> $OUT/single_rank$r.err
done
python - <<PY > $OUT/oversub_compare.txt 2>&1
import numpy as np, os
out = "$OUT"
for tag in ("oversub_nccl", "oversub_gloo"):
    for r in (0, 1):
        a, b = os.path.join(out, "%s.rank%d.npy" % (tag, r)), os.path.join(out, "single.rank%d.npy" % r)
        if os.path.exists(a) and os.path.exists(b):
            x, y = np.load(a), np.load(b)
            print(tag, "rank", r, "identical to the single-process run of the same streams:", bool(np.array_equal(x, y)), "max abs diff", float(np.abs(x - y).max()))
        else:
            print(tag, "rank", r, "missing dump")
PY
rm -f $OUT/*.npy
tail -n 30 $OUT/valu_rate.txt $OUT/wg_placement.txt $OUT/oversub_rc.txt $OUT/oversub_compare.txt
tail -c 1500 $OUT/bench_base.json
