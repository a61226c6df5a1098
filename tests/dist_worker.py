"""Worker for test_multi_gpu_gloo.py: one process per (emulated) GPU, gloo backend, batch sharded by stream."""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from conftest import synth_input
    pkg = importlib.import_module("signalsmith-stretch_amd")
    sharding = importlib.import_module("signalsmith-stretch_amd.sharding")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    emu = pkg.bind(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libsmst_emu.so")))
    # argv: <output file> [total streams] [n in] [n out]; beyond 2x the hops draw random time factors, so the per-rank seed matters:
    # global stream g carries the reference engine seeded g = the rank's first stream + i (bench.py: seed = rank * streams per GPU)
    total = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
    nout = int(sys.argv[4]) if len(sys.argv) > 4 else 3600
    C = 2
    lo, hi = sharding.shard_range(total, rank, world)
    xs = np.stack([synth_input(s, C, n, 48000) for s in range(lo, hi)])
    b = pkg.StretchBatch(hi - lo, C, block=512, interval=128, lib=emu, seed=lo)
    for i, s in enumerate(range(lo, hi)):
        b.setTransposeSemitones(float(s % 5 - 2), 0.0, stream=i)
    dist.barrier()
    y = b.process(xs, nout)
    dist.barrier()
    # gather every shard on rank 0 (results only -- the data path itself needed no exchange)
    gathered = [None]*world
    dist.all_gather_object(gathered, (lo, hi, y))
    # the max-over-ranks clock bench.py uses
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    if rank == 0:
        full = np.concatenate([g[2] for g in sorted(gathered, key=lambda g: g[0])], axis=0)
        np.save(sys.argv[1], full)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
