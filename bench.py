#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): Msamples/sec (in+out) at 48 kHz stereo presetDefault.

One "step" = one process() pass of the hot path over the whole resident batch (BASELINE configs[1]: 256 stereo
streams, 48 kHz, presetDefault, 1.5x stretch, fp32, 10 s of input per stream), inputs and outputs resident in HBM.
N GPUs: one process per GPU, the batch of streams is sharded (256 streams per GPU, weak scaling), no collective on
the data path; torch.distributed (RCCL) is used only for the barrier and the max-over-ranks clock.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the engine's own stream) and
`cpu_baseline` (the CPU reference timed on this box's host cores; N=1 only)."""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 48000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_channel_hop(B, I, M, r):
    """SURVEY.md section 8(d) state-streaming model: new input + output + Band.output/Prediction.energy read+write
    (+ prevInput carried when not re-analysed) + overlap-add partial sums read+write."""
    return 4*(I/r) + 4*I + 2*12*M + (2*8*M if r == 1 else 0) + 2*4*(B - I)


def make_inputs(torch, S, C, n, device, first_stream=0, sr=SR):
    """Synthetic streams of SURVEY.md 8(d): type = s mod 3 (sine pair / chirp / uniform noise), generated on the GPU."""
    t = torch.arange(n, device=device, dtype=torch.float64)/sr
    x = torch.empty((S, C, n), dtype=torch.float32, device=device)
    gen = torch.Generator(device=device)
    for s in range(S):
        sg = first_stream + s
        for c in range(C):
            if sg % 3 == 0:
                f1 = 110*2**((sg % 37)/12)
                v = 0.4*torch.sin(2*torch.pi*f1*t + 0.5*c) + 0.2*torch.sin(2*torch.pi*3.17*f1*t)
            elif sg % 3 == 1:
                k = (0.4*sr - 50)/(n/sr)
                v = 0.5*torch.sin(2*torch.pi*(50*t + 0.5*k*t*t) + 0.5*c)
            else:
                gen.manual_seed(1_000_003*sg + c)
                v = torch.rand(n, generator=gen, device=device, dtype=torch.float32)*0.6 - 0.3
            x[s, c] = v.to(torch.float32)
    return x


def host_topology():
    """(physical core count, one logical CPU per physical core, logical CPU count) from sysfs thread-sibling lists."""
    logical = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    firsts = {}
    for cpu in logical:
        path = "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu
        try:
            sib = open(path).read().strip()
            first = int(sib.replace("-", ",").split(",")[0])
        except (OSError, ValueError):
            first = cpu
        firsts.setdefault(first, cpu)
    per_core = sorted(firsts.values())
    return len(per_core), per_core, len(logical)


def cpu_baseline(budget_s=6.0):
    """The CPU reference (oracle/_ref) on this box's host cores: ONE PROCESS PER PHYSICAL CORE, each pinned to its core and
    rendering 10 s stereo streams at 1.5x back to back for `budget_s` seconds of process() time (bounded sample).  The
    earlier one-process-per-LOGICAL-core figure swung by 2x between boxes (SMT siblings + unpinned processes contending);
    `single_core` is the same binary on one otherwise idle core, measured first."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_oracle
    if not ref_oracle.available():
        return None
    physical, cpus, logical = host_topology()
    script = os.path.join(ROOT, "oracle", "cpu_baseline.py")

    def launch(cpu, budget, first):
        def pin():
            try:
                os.sched_setaffinity(0, {cpu})
            except OSError:
                pass
        return subprocess.Popen([sys.executable, script, str(budget), "10", "1.5", str(first)], stdout=subprocess.PIPE, text=True, preexec_fn=pin)
    one = json.loads(launch(cpus[0], 1.5, 0).communicate()[0].strip().splitlines()[-1])
    single = one["samples"]/one["process_s"]
    t0 = time.perf_counter()
    procs = [launch(cpu, budget_s, 3*i) for i, cpu in enumerate(cpus)]
    outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
    wall = time.perf_counter() - t0
    rate = sum(o["samples"]/o["process_s"] for o in outs)  # all cores busy concurrently: sum of per-core rates
    streams = sum(o["streams"] for o in outs)
    return dict(value=rate/1e6, unit="Msamples/s", cores=physical, kind="reference",
                sample="%d pinned processes (one per PHYSICAL core; %d logical CPUs) x ~%.0f s of process() each = %d stereo 48 kHz "
                       "presetDefault streams of 10 s at 1.5x; unmodified reference header, g++ -O3, L1 (signalsmith-linear) restated "
                       "in oracle/linear_shim; wall %.1f s" % (physical, logical, budget_s, streams, wall),
                logical_cpus=logical, single_core_Msamples_s=single/1e6, per_core_when_all_busy_Msamples_s=rate/1e6/max(physical, 1),
                realtime_x=rate/(2*2.5*SR))


def pin_rank_to_numa_node(local_rank, world):
    """Multi-GPU runs: keep each rank's host scheduler (the block scheduler of process() is single-threaded host code) on
    the CPUs of the NUMA node its GPU hangs off; ranks that share a node split its CPUs.  Best effort, silent when sysfs
    has no answer."""
    try:
        import torch
        bus = torch.cuda.get_device_properties(local_rank).pci_bus_id if hasattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id") else None
        node = None
        if bus is not None:
            for cand in ("/sys/bus/pci/devices/0000:%02x:00.0/numa_node" % bus,):
                if os.path.exists(cand):
                    node = int(open(cand).read().strip())
        cpus = sorted(os.sched_getaffinity(0))
        if node is not None and node >= 0:
            spec = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
            node_cpus = []
            for part in spec.split(","):
                a, _, b = part.partition("-")
                node_cpus += list(range(int(a), int(b or a) + 1))
            cpus = [c for c in cpus if c in set(node_cpus)] or cpus
        share = max(1, len(cpus)//max(world, 1))
        mine = cpus[(local_rank*share) % len(cpus):][:share] or cpus
        os.sched_setaffinity(0, set(mine))
        return mine
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=256, help="streams per GPU (BASELINE configs[1]: 256)")
    ap.add_argument("--seconds", type=float, default=10.0, help="seconds of input per stream per step")
    ap.add_argument("--stretch", type=float, default=1.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serial-pass", action="store_true", help="skip the extra, serialised per-kernel-class profiling call (used under rocprofv3, so that every launch it sees is an in-place one)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank: exercises the barrier / max-over-ranks path of the N>1 runs on a 1-GPU box")
    ap.add_argument("--half-state", action="store_true", help="BASELINE config 5 'fp16 internal': carried state and overlap-add sums stored in fp16 (SMST_FLAG_HALF_STATE)")
    ap.add_argument("--config", default="2", choices=["2", "3", "4", "4b", "5"],
                    help="BASELINE.json config (default 2 = the one the headline metric is quoted on; the others are "
                         "reported in DESIGN.md, they are not the bench line)")
    args = ap.parse_args()
    preset, C, sr_cfg = "default", 2, SR
    setup = None
    per_stream = None
    if args.config == "3":
        args.streams, args.stretch = 1024, 1.0
        setup = lambda b: b.setTransposeSemitones(12, 8000/48000)  # noqa: E731
    elif args.config in ("4", "4b"):
        args.streams, args.stretch = 512, 0.75
        def setup(b):
            if args.config == "4b":
                b.setTransposeSemitones(4, 8000/48000)
            b.setFormantFactor(1, True)
            b.setFormantBase(200/48000)
    elif args.config == "5":
        args.streams, args.seconds, preset, C, sr_cfg = 1024, 2.0, "cheaper", 8, 96000
        per_stream = True

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    affinity = pin_rank_to_numa_node(local_rank, world) if world > 1 else None
    pkg = importlib.import_module("signalsmith-stretch_amd")
    S = args.streams
    n_in = int(args.seconds*sr_cfg)
    n_out = int(round(n_in*args.stretch))
    batch = pkg.StretchBatch(S, C, preset=preset, sample_rate=sr_cfg, device=local_rank, seed=rank, half_state=args.half_state)
    if setup:
        setup(batch)
    if per_stream:  # config 5: per-stream random stretch 0.75-1.5x and +-12 st
        import numpy as np
        g = np.random.Generator(np.random.PCG64(5))
        stretches = g.uniform(0.75, 1.5, 8192)[rank*S:(rank + 1)*S]
        semis = g.uniform(-12, 12, 8192)[rank*S:(rank + 1)*S]
        for i in range(S):
            batch.setTransposeSemitones(float(semis[i]), 0.0, stream=i)
        n_out = [int(round(n_in*float(v))) for v in stretches]
    x = make_inputs(torch, S, C, n_in, device, first_stream=rank*S, sr=sr_cfg)
    n_out_max = max(n_out) if per_stream else n_out
    y = torch.empty((S, C, n_out_max), dtype=torch.float32, device=device)
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        batch.process(x, n_out, out=y, ordered=False)
    batch.synchronize()
    barrier()
    torch.cuda.synchronize()
    # the recurrence kernel is timed IN PLACE over the timed region: a HIP-event pair on the stream it is launched on
    # (the engine's chain stream), the other streams keep overlapping it (two event records per 64-hop tile)
    batch.enableProfiling(0 if os.environ.get("SMST_BENCH_NO_LIVE") else 2)
    t0 = time.perf_counter()
    # inputs are complete (synchronised above) and the outputs are only looked at after batch.synchronize(): no per-call
    # ordering against torch's stream, so the host scheduling of step n+1 overlaps the kernels of step n
    for _ in range(args.steps):
        batch.process(x, n_out, out=y, ordered=False)
    batch.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    live_ms, live_launches = batch.takeTimings()
    batch.enableProfiling(0)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ok = bool(torch.isfinite(y).all().item()) and float(y.abs().max().item()) > 0.01

    total_out = sum(n_out) if per_stream else S*n_out
    samples_local = C*(S*n_in + total_out)
    if use_dist:  # per-stream stretch factors differ between ranks (config 5): add up what every rank really processed
        tot = torch.tensor([samples_local], dtype=torch.float64, device=device)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        samples_per_step = int(tot.item())
    else:
        samples_per_step = samples_local
    value = samples_per_step*args.steps/elapsed/1e6
    B, I, M = batch.blockSamples(), batch.intervalSamples(), batch.bands()
    hops_per_stream = -(-(total_out//S)//I)
    bytes_per_chop = algorithmic_bytes_per_channel_hop(B, I, M, args.stretch)

    roofline = None
    if rank == 0:
        # per-kernel-class device time: HIP events recorded on the engine's own stream around every launch
        if args.no_serial_pass:
            tiles = live_launches.get("chain_live", 0)/max(args.steps, 1)
            ms = {k: 0.0 for k in ("analyse", "feed", "predict", "chain", "synth", "emit", "other")}
            ms["chain"] = live_ms.get("chain_live", 0.0)/max(args.steps, 1)
            launches = {k: 0 for k in ("analyse", "predict", "chain", "synth", "emit")}
            launches["chain"] = int(tiles)
        else:
            batch.enableProfiling(1)
            batch.process(x, n_out, out=y, ordered=False)
            batch.synchronize()
            ms, launches = batch.takeTimings()
            batch.enableProfiling(0)
            ms.pop("chain_live", None)
            launches.pop("chain_live", None)
        launch_count = {"analyse": launches["analyse"], "predict": launches["predict"], "chain": launches["chain"],
                        "synth": launches["synth"], "emit": launches["emit"]}
        dom = max(launch_count, key=lambda k: ms[k])
        avg_ms_serial = ms[dom]/max(launch_count[dom], 1) or float("nan")
        avg_ms = avg_ms_serial
        if dom == "chain" and live_launches.get("chain_live", 0) > 0:  # in place, over the timed steps (agrees with rocprofv3)
            avg_ms = live_ms["chain_live"]/live_launches["chain_live"]
        chops_per_launch = S*C*hops_per_stream/max(launch_count[dom], 1)
        achieved = bytes_per_chop*chops_per_launch/(avg_ms*1e-3)/1e9 if avg_ms == avg_ms and avg_ms > 0 else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get({"analyse": "kAnalyseFast", "predict": "kPredictB", "chain": "kVocoder" if C <= 2 else "kChain",
                                                      "synth": "kSynthFast", "emit": "kEmit"}[dom])
            except Exception:
                traffic = None
        names = {"analyse": "kAnalyseFast", "predict": "kPredictB", "chain": "kVocoder" if C <= 2 else "kChain", "synth": "kSynthFast", "emit": "kEmit"}
        roofline = dict(bound="hbm", kernel=names[dom],
                        achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=(achieved/HBM_PEAK_GBS if achieved else None), traffic=traffic,
                        avg_launch_ms=avg_ms, avg_launch_ms_alone=avg_ms_serial, launches_per_step=launch_count[dom],
                        algorithmic_bytes_per_channel_hop=bytes_per_chop, channel_hops_per_launch=chops_per_launch,
                        kernel_ms_per_step_alone={k: round(v, 3) for k, v in ms.items()},
                        pipeline_frac=(bytes_per_chop*S*C*hops_per_stream/(elapsed/args.steps))/1e9/HBM_PEAK_GBS)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline()
        except Exception as e:  # the baseline is reported, never required
            cpu = dict(error=str(e))
    if rank == 0:
        line = {
            "metric": "Msamples/sec (in+out) at 48kHz stereo presetDefault",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed/args.steps*1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "state_storage": "f16" if args.half_state else "f32",
            "config": {"workload": ("BASELINE configs[1]: %d stereo streams per GPU, 48 kHz, presetDefault, %.2fx stretch, fp32, "
                                    "%.0f s input per stream per step, device-resident I/O" % (S, args.stretch, args.seconds))
                       if args.config == "2" else "BASELINE config %s (not the headline): %d streams x %d ch per GPU, %d Hz, preset %s, %.0f s per step"
                       % (args.config, S, C, sr_cfg, preset, args.seconds),
                       "streams_total": world*S, "channels": C, "block": B, "interval": I, "fft": batch.fftSamples(),
                       "hops_per_stream_per_step": hops_per_stream, "sharding": "streams/%d, no collective" % world,
                       "rank0_cpu_affinity": ("%d CPUs from %d" % (len(affinity), affinity[0])) if affinity else None},
            "realtime_x": world*S*args.seconds*args.steps/elapsed,
            "channels": C,
            "output_finite_nonzero": ok,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()
    if rank == 0 and not ok:
        raise SystemExit("bench: output not finite / silent -- the line above is not a valid measurement")


if __name__ == "__main__":
    main()
