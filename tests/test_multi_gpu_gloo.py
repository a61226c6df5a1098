"""The N>1 path on CPU: world_size 2, gloo, streams sharded across ranks with no data-path collective; the gathered
result equals the single-process run of the whole batch bit for bit."""
import importlib
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT, package, synth_input


def test_shard_range_partitions():
    sharding = importlib.import_module("signalsmith-stretch_amd.sharding")
    for total in (1, 5, 256, 4096, 8192, 13):
        for world in (1, 2, 3, 4, 8):
            ranges = [sharding.shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_sizes(4096, 8) == [512]*8  # BASELINE config 4


def test_world_size_2_gloo_matches_single_process(emu, tmp_path):
    pkg = package()
    out = str(tmp_path/"gathered.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29511", os.path.join(ROOT, "tests", "dist_worker.py"), out]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    sharded = np.load(out)
    total, C, n, nout = 5, 2, 3000, 3600
    xs = np.stack([synth_input(s, C, n, 48000) for s in range(total)])
    b = pkg.StretchBatch(total, C, block=512, interval=128, lib=emu)
    for s in range(total):
        b.setTransposeSemitones(float(s - 2), 0.0, stream=s)
    whole = b.process(xs, nout)
    assert sharded.shape == whole.shape
    assert np.array_equal(sharded, whole)
