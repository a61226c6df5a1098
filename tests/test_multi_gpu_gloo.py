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
        b.setTransposeSemitones(float(s % 5 - 2), 0.0, stream=s)
    whole = b.process(xs, nout)
    assert sharded.shape == whole.shape
    assert np.array_equal(sharded, whole)


def test_world_size_8_gloo_per_rank_seeds_match_single_process(emu, tmp_path):
    """Eight ranks, as BASELINE config 4 shards its 4096 streams (8 x 512, seed of rank r = 512 r: bench.py), here with 3 streams per
    rank so that the CPU stand-in finishes: every hop runs beyond 2x (1200 -> 3000 samples), where the per-bin time factors come from
    the stream's random engine -- the gathered shards equal ONE batch of all 24 streams seeded 0 only if every rank seeded its shard
    with its first global stream index."""
    sharding = importlib.import_module("signalsmith-stretch_amd.sharding")
    assert sharding.shard_sizes(4096, 8) == [512]*8 and [sharding.shard_range(4096, r, 8)[0] for r in range(8)] == [512*r for r in range(8)]
    pkg = package()
    out = str(tmp_path/"gathered8.npy")
    total, C, n, nout = 24, 2, 1200, 3000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29518", os.path.join(ROOT, "tests", "dist_worker.py"), out, str(total), str(n), str(nout)]
    res = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    sharded = np.load(out)
    xs = np.stack([synth_input(s, C, n, 48000) for s in range(total)])
    b = pkg.StretchBatch(total, C, block=512, interval=128, lib=emu, seed=0)
    for s in range(total):
        b.setTransposeSemitones(float(s % 5 - 2), 0.0, stream=s)
    whole = b.process(xs, nout)
    assert sharded.shape == whole.shape and np.array_equal(sharded, whole)
    # ... and the seeds do matter: the same shard seeded 0 on every rank would not reproduce streams 3..23
    wrong = pkg.StretchBatch(3, C, block=512, interval=128, lib=emu, seed=0)
    for i in range(3):
        wrong.setTransposeSemitones(float((3 + i) % 5 - 2), 0.0, stream=i)
    assert not np.array_equal(wrong.process(xs[3:6], nout), whole[3:6])


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("smst_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_self_launch_command():
    """`python bench.py --gpus N` without a launcher starts its own N ranks the way the driver does -- or refuses loudly."""
    import pytest
    bench = _bench_module()
    cmd, env = bench.self_launch_command(4, False, 8, ["--gpus", "4", "--steps", "3"], {"MASTER_PORT": "29999"})
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert os.path.basename(cmd[-5]) == "bench.py" and env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    with pytest.raises(SystemExit) as e:
        bench.self_launch_command(2, False, 1, ["--gpus", "2"], {})
    assert "only 1 GPU(s) are visible" in str(e.value)
    cmd, _ = bench.self_launch_command(2, True, 1, ["--gpus", "2", "--oversubscribe"], {})  # N ranks on one device: allowed, says so in the line
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2"


def test_bench_refuses_more_gpus_than_visible():
    """End to end on whatever box this runs on: asking for more GPUs than are visible is an error, not an n_gpus: 1 line."""
    import torch
    ask = torch.cuda.device_count() + 1 if torch.cuda.device_count() > 0 else 2
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ask), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert res.returncode != 0
    assert "GPU(s) are visible" in res.stderr and '"n_gpus"' not in res.stdout


def test_bench_refuses_world_size_mismatch():
    """Under a launcher with the wrong rank count the bench refuses as well (it used to print n_gpus: 1 for --gpus 8 with WORLD_SIZE unset)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0 and "WORLD_SIZE (1) != --gpus (2)" in res.stderr


def test_numa_pinning_is_loud(monkeypatch):
    """Every rank says where its host scheduler runs; a device whose PCI address cannot be read stops a multi-GPU run (eight ranks silently
    on rank 0's CPUs would look like bad scaling); a readable address without a NUMA node in sysfs still gives every rank CPUs of its own."""
    import pytest
    bench = _bench_module()
    assert bench.cpu_ranges([0, 1, 2, 3, 8, 9, 11]) == "0-3,8-9,11"
    before = os.sched_getaffinity(0)
    try:
        monkeypatch.setattr(bench, "device_pci_address", lambda i: None)
        with pytest.raises(SystemExit) as e:
            bench.pin_rank_to_numa_node(1, 4, strict=True)
        assert "PCI address" in str(e.value)
        relaxed = bench.pin_rank_to_numa_node(1, 4, strict=False)
        assert relaxed["pinned"] is False and relaxed["cpus"]
        if len(before) >= 4:
            monkeypatch.setattr(bench, "device_pci_address", lambda i: "0000:ff:1f.0")  # no such device in sysfs: the node is unknown
            a = bench.pin_rank_to_numa_node(0, 4, strict=True)
            os.sched_setaffinity(0, before)
            b = bench.pin_rank_to_numa_node(1, 4, strict=True)
            assert a["pinned"] and b["pinned"] and a["cpus"] != b["cpus"] and a["pci"] == "0000:ff:1f.0"
    finally:
        os.sched_setaffinity(0, before)
