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
# The engine pipelines a step over three HIP streams (+ one for its tables); HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4), and two pipeline streams that share a queue run in submission order.  An RCCL communicator brings streams of its own:
# with the default, the same step took 15.8 instead of 14.0 ms once torch.distributed was initialised (profiles/r4_hw_queues.txt).
# Set before the HIP runtime starts (INTEGRATION.md says the same to applications).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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


def cpu_baseline(budget_s=5.0):
    """The CPU reference (oracle/_ref) on this box's host cores, timed TWICE on a bounded sample: one pinned process per
    PHYSICAL core, and one pinned process per LOGICAL CPU (SMT siblings busy too).  `value` is the LARGER aggregate (the
    better the CPU does, the more honest any ratio against it); both are reported, with the single-idle-core rate measured
    first.  Each process renders 10 s stereo streams at 1.5x back to back for `budget_s` seconds of process() time."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_oracle
    if not ref_oracle.available():
        return None
    physical, cpus, logical = host_topology()
    all_cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    script = os.path.join(ROOT, "oracle", "cpu_baseline.py")

    def launch(cpu, budget, first):
        def pin():
            try:
                os.sched_setaffinity(0, {cpu})
            except OSError:
                pass
        return subprocess.Popen([sys.executable, script, str(budget), "10", "1.5", str(first)], stdout=subprocess.PIPE, text=True, preexec_fn=pin)

    def aggregate(cpu_list):
        t0 = time.perf_counter()
        procs = [launch(cpu, budget_s, 3*i) for i, cpu in enumerate(cpu_list)]
        outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
        wall = time.perf_counter() - t0
        # all processes busy concurrently: sum of per-process rates (time inside process() only)
        return sum(o["samples"]/o["process_s"] for o in outs), sum(o["streams"] for o in outs), wall

    one = json.loads(launch(cpus[0], 1.5, 0).communicate()[0].strip().splitlines()[-1])
    single = one["samples"]/one["process_s"]
    rate_phys, streams_phys, wall_phys = aggregate(cpus)
    rate_log, streams_log, wall_log = (rate_phys, streams_phys, wall_phys) if logical == physical else aggregate(all_cpus)
    use_logical = rate_log > rate_phys
    rate, threads = (rate_log, logical) if use_logical else (rate_phys, physical)
    return dict(value=rate/1e6, unit="Msamples/s", cores=threads, kind="reference",
                sample="the larger of two aggregates, each = pinned processes x ~%.0f s of process() on stereo 48 kHz presetDefault "
                       "streams of 10 s at 1.5x: %d processes (one per PHYSICAL core) = %.1f Msamples/s over %d streams, wall %.1f s; "
                       "%d processes (one per LOGICAL CPU) = %.1f Msamples/s over %d streams, wall %.1f s; unmodified reference "
                       "header, g++ -O3, L1 (signalsmith-linear) restated in oracle/linear_shim"
                       % (budget_s, physical, rate_phys/1e6, streams_phys, wall_phys, logical, rate_log/1e6, streams_log, wall_log),
                physical_cores=physical, logical_cpus=logical,
                per_physical_core_pinned_Msamples_s=rate_phys/1e6, per_logical_cpu_pinned_Msamples_s=rate_log/1e6,
                single_core_Msamples_s=single/1e6, realtime_x=rate/(2*2.5*SR))


def cmd_binary_baseline(physical_cpus, seconds=30.0, stretch=1.5):
    """The reference's OWN cmd/ binary (oracle/_ref/ref_cli = /root/reference/cmd/main.cpp compiled unmodified) on the same kind of
    input, WAV file in -> WAV file out: alone on an idle core, then one pinned process per physical core for a quarter of the cores and
    for all of them.  Every process works in a directory of its own on tmpfs with its own copy of the input (round 4 had all of them
    read one file and write into one directory).  The binary decodes and encodes 16-bit WAV and starts a process per file: that is
    INSIDE its time by construction (the reference has no other entry point for files), so per process the wall time is reported next
    to the CPU time the kernel charged it (user + system, os.wait4): where the two part, the processes are waiting for each other in
    the file system, not computing.  Samples counted as the GPU line counts them (in + out, per channel).  The figure to compare a GPU
    number with is `cpu_baseline.value` -- the same reference code timed around process() only."""
    import shutil
    import struct
    import tempfile
    import numpy as np
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_cli")
    if not os.path.exists(exe):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import synth_input
    n = int(seconds*SR)
    x = 0.8*synth_input(0, 2, n, SR)
    tmp = tempfile.mkdtemp(prefix="smst_cmd_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    data = np.clip(np.round(x.T*32768.0), -32768, 32767).astype("<i2").tobytes()
    wav = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, SR, SR*4, 4, 16) + b"data" + struct.pack("<I", len(data)) + data
    samples = 2*(n + int(round(n*stretch)))

    def prepare(i):
        d = os.path.join(tmp, "p%d" % i)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "in.wav"), "wb") as f:
            f.write(wav)
        return d

    def launch(cpu, d):
        def pin():
            try:
                os.sched_setaffinity(0, {cpu})
            except OSError:
                pass
        return subprocess.Popen([exe, os.path.join(d, "in.wav"), os.path.join(d, "out.wav"), "--time=%g" % stretch], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, preexec_fn=pin)

    def run(cpus):
        dirs = [prepare(i) for i in range(len(cpus))]  # (the input copies are written before the clock starts)
        t0 = time.perf_counter()
        procs = [launch(cpu, d) for cpu, d in zip(cpus, dirs)]
        cpu_s, ok = [], True
        for p in procs:
            _, status, ru = os.wait4(p.pid, 0)
            p.returncode = os.waitstatus_to_exitcode(status) if hasattr(os, "waitstatus_to_exitcode") else (status >> 8)
            ok = ok and p.returncode == 0
            cpu_s.append(ru.ru_utime + ru.ru_stime)
        wall = time.perf_counter() - t0
        for d in dirs:
            shutil.rmtree(d, ignore_errors=True)
        return dict(processes=len(cpus), wall_s=wall, aggregate_Msamples_s=(samples*len(cpus)/wall/1e6 if ok else None),
                    mean_cpu_s_per_process=sum(cpu_s)/len(cpu_s), Msamples_s_per_process_cpu_time=samples/(sum(cpu_s)/len(cpu_s))/1e6, ok=ok)
    try:
        single = run(physical_cpus[:1])
        if not single["ok"]:
            return dict(error="ref_cli failed")
        quarter = run(physical_cpus[:max(1, len(physical_cpus)//4)])
        full = run(physical_cpus)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return dict(binary="oracle/_ref/ref_cli (cmd/main.cpp, unmodified; g++ -O3; L1 restated)", what="%.0f s stereo 48 kHz 16-bit WAV -> WAV at %.2fx, presetDefault; "
                "whole process incl. start-up and file I/O, a tmpfs directory and input copy per process" % (seconds, stretch),
                single_process_s=single["wall_s"], single_process_Msamples_s=single["aggregate_Msamples_s"],
                processes=full["processes"], all_processes_wall_s=full["wall_s"], aggregate_Msamples_s=full["aggregate_Msamples_s"],
                runs=[single, quarter, full],
                authoritative="cpu_baseline.value (the same reference code as a library, timed around process() only) is the CPU figure to hold a GPU "
                              "number against; this binary's aggregate is bounded by process start-up and WAV file I/O once many run at the same time "
                              "(compare wall_s with mean_cpu_s_per_process in `runs`)")


# rel-RMS of stream 0's first second against oracle/_ref as measured on the MI355X (profiles/r4_bench_*.json; stream 0 is the tonal stream):
# the bound of the line is TEN times that -- a regression of the arithmetic by an order of magnitude fails the bench
SELF_CHECK_MEASURED = {"2": 1.34e-4, "3": 2.3e-5, "4": 7.5e-5, "4b": 7.5e-5, "5": 4.7e-4}


def self_check(batch_first_output, x0, n_out0, C, sr_cfg, preset, setup, seconds=1.0, config="2"):
    """Stream 0 of the benched batch's FIRST call (from the reset state; taken outside the timed region) against oracle/_ref on the same
    input: relative RMS over the first `seconds` of output (the free-running phase recurrence is chaotic, so only a short horizon says
    anything sample by sample) and the level ratio over the whole call."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import ref_oracle
    if not ref_oracle.available():
        return dict(checked=False, reason="oracle/_ref not built on this box")
    r = ref_oracle.RefStretch(seed=0)
    (r.presetCheaper if preset == "cheaper" else r.presetDefault)(C, sr_cfg)
    if setup:
        setup(r)
    ref = r.process(x0, n_out0)
    got = batch_first_output[:, :n_out0]
    k = min(n_out0, int(seconds*sr_cfg))
    err = float(np.sqrt(np.mean((got[:, :k] - ref[:, :k])**2)/max(np.mean(ref[:, :k]**2), 1e-30)))
    level = float(np.sqrt(np.mean(got**2)/max(np.mean(ref**2), 1e-30)))
    bound = 10*SELF_CHECK_MEASURED.get(str(config), 2e-3)
    return dict(checked=True, stream=0, call="first call of the benched batch (reset state), outside the timed region", rel_rms_first_second=err,
                level_ratio_whole_call=level, ok=bool(err < bound and abs(level - 1) < 1e-3),
                bound="rel-RMS < %.1e over the first second (10 x the value measured for this config on the MI355X), level within 0.1 %%" % bound)


def cpu_ranges(cpus):
    """[0, 1, 2, 3, 8, 9] -> "0-3,8-9" """
    out, cpus = [], sorted(cpus)
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def device_pci_address(index):
    """"dddd:bb:dd.f" of HIP device `index`: from the HIP runtime itself (hipDeviceGetPCIBusId through the library torch has loaded),
    else from torch's device properties; None if neither answers."""
    import ctypes
    try:
        import torch
        libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
        for name in ("libamdhip64.so", "libamdhip64.so.6", "libamdhip64.so.7"):
            path = os.path.join(libdir, name)
            if not os.path.exists(path):
                continue
            hip = ctypes.CDLL(path)
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0 and buf.value:
                return buf.value.decode().lower()
    except Exception:
        pass
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        if hasattr(p, "pci_bus_id"):
            return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
    except Exception:
        pass
    return None


def pin_rank_to_numa_node(local_rank, world, share_index=None, strict=True):
    """Multi-GPU runs: keep each rank's host scheduler (the block scheduler of process() is single-threaded host code) on the CPUs of
    the NUMA node its GPU hangs off; ranks whose GPUs share a node split its CPUs (share_index of `world` shares).  LOUD: returns what
    it did -- PCI address, NUMA node, the CPU set it chose -- for the JSON line of EVERY rank, and with strict=True (--gpus > 1 on real
    devices) it raises when the device's PCI address or NUMA node cannot be read: eight ranks silently sharing rank 0's CPUs would
    look like bad scaling of the engine."""
    report = dict(device=local_rank, pci=None, numa_node=None, cpus=None, pinned=False)
    pci = device_pci_address(local_rank)
    report["pci"] = pci
    node = None
    if pci is not None:
        path = "/sys/bus/pci/devices/%s/numa_node" % pci
        if os.path.exists(path):
            node = int(open(path).read().strip())
    report["numa_node"] = node
    if pci is None:
        if strict:
            raise SystemExit("bench: cannot read the PCI address of device %d: refusing to run %d ranks unpinned "
                             "(--no-numa-pinning to run them where the launcher put them)" % (local_rank, world))
        report["cpus"] = cpu_ranges(os.sched_getaffinity(0))
        return report
    if node is None:  # the address is known, sysfs does not say which node it hangs off (some containers hide it): say so, and still give every rank CPUs of its own
        print("bench: /sys/bus/pci/devices/%s/numa_node is not readable: rank on device %d gets its share of ALL CPUs instead of its NUMA node's" % (pci, local_rank), file=sys.stderr)
        node = -1
    cpus = sorted(os.sched_getaffinity(0))
    if node >= 0:  # (-1: the platform has a single node / does not say: the whole affinity mask)
        node_cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            node_cpus.update(range(int(a), int(b or a) + 1))
        cpus = [c for c in cpus if c in node_cpus] or cpus
    share = max(1, len(cpus)//max(world, 1))
    k = local_rank if share_index is None else share_index
    mine = cpus[(k*share) % len(cpus):][:share] or cpus
    os.sched_setaffinity(0, set(mine))
    report["cpus"] = cpu_ranges(mine)
    report["pinned"] = True
    return report


def own_algorithmic_bytes(B, I, M, r):
    """The SURVEY.md 8(d) figure split by the kernel class that owns each term (they add up to the whole):
    analysis = the new input samples; recurrence = the carried per-bin state read + written (Band.output 8 B +
    Prediction.energy 4 B, + prevInput when it is carried instead of re-analysed); synthesis + emission = the output
    samples and the overlap-add partial sums read + written."""
    return {"analyse": 4*(I/r), "chain": 2*12*M + (2*8*M if r == 1 else 0), "synth+emit": 4*I + 2*4*(B - I)}


def library_sha16():
    import hashlib
    pkg = importlib.import_module("signalsmith-stretch_amd")
    with open(pkg.LIBRARY_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def measured_traffic(sha16, config="2"):
    """HBM bytes per step from the PMC passes (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 correction:
    profiles/summarize.py) -- ONLY if profiles/traffic_latest.json was measured on this very library build; a bench
    run cannot collect PMC counters itself, so anything else is reported as null."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json" if config in ("2", None) else "traffic_latest_config%s.json" % config)
    try:
        t = json.load(open(path))
    except Exception:
        return None, "none (no PMC summary %s in profiles/)" % os.path.basename(path)
    if t.get("library_sha16") != sha16:
        return None, "none (profiles/%s was measured on library %s, this run uses %s)" % (os.path.basename(path), t.get("library_sha16"), sha16)
    return t, "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `%s` on library %s%s)" % (
        os.path.basename(path), t.get("command", "bench.py"), sha16,
        ("; measured on %d streams and scaled by the stream count: every stream does the same work" % t["streams"]) if t.get("streams") else "")


def self_launch_command(gpus, oversubscribe, visible_devices, argv, environ):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE): the command that starts the N ranks exactly as the driver does (one
    process per GPU under torch.distributed.run, rendezvous on 127.0.0.1) -- or a loud refusal when the box has fewer devices than
    asked for, instead of a line that says n_gpus: 1."""
    if not oversubscribe and visible_devices < gpus:
        raise SystemExit("bench: --gpus %d asked for, but only %d GPU(s) are visible on this box -- refusing to print a line for fewer "
                         "devices than requested (use --oversubscribe --dist-backend gloo to exercise the N-rank launch path on one device)" % (gpus, visible_devices))
    port = environ.get("MASTER_PORT", str(29500 + (os.getpid() % 400)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + list(argv)
    env = dict(environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return cmd, env


def other_measurements():
    """Short runs of BASELINE configs 3 / 4b / 5 (2 timed steps each) and 300 quanta of the real-time calling pattern at 4096 streams, each
    in a fresh process (this script / tools/bench_realtime.py).  Then tools/bench_dropin.cpp: 64 reference-style objects against the batch API.
    Returns (other_configs, realtime, dropin); a run that fails is reported as such."""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    others = {}
    for cfg in ("3", "4b", "5"):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-serial-pass", "--no-other-configs"]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rl = d.get("roofline") or {}
            sc = d.get("self_check") or {}
            others[cfg] = dict(workload=d["config"]["workload"], value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"], steps=d["steps"], frac=rl.get("frac"),
                               realtime_x=d.get("realtime_x"), traffic=rl.get("traffic"), recurrence_ms_per_launch=(rl.get("dominant_kernel") or {}).get("avg_launch_ms"),
                               self_check={k: sc.get(k) for k in ("checked", "ok", "rel_rms", "bound", "level_ratio") if k in sc}, run_seconds=round(time.perf_counter() - t0, 1))
        except Exception as e:
            others[cfg] = dict(error=str(e)[:300], run_seconds=round(time.perf_counter() - t0, 1))
    realtime = None
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, os.path.join(here, "tools", "bench_realtime.py"), "--streams", "4096", "--quanta", "300"], env=env, capture_output=True, text=True, timeout=240)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        realtime = dict(pattern=d["pattern"], quanta_timed=d["quanta_timed"], **d["rows"][0], run_seconds=round(time.perf_counter() - t0, 1))
    except Exception as e:
        realtime = dict(error=str(e)[:300], run_seconds=round(time.perf_counter() - t0, 1))
    dropin = None
    t0 = time.perf_counter()
    try:
        exe = os.path.join(here, "signalsmith-stretch_amd", "bench_dropin")
        r = subprocess.run([exe, "64", "1", "6"], env=env, capture_output=True, text=True, timeout=240)
        dropin = json.loads(r.stdout.strip().splitlines()[-1])
        dropin["what"] = ("64 independent SignalsmithStretch<float> objects (the reference's class through the drop-in header: one single-stream engine per object, host "
                          "buffers, synchronous copies per call) called from one thread, against the same 64 streams through smst_batch_* with host buffers")
        dropin["run_seconds"] = round(time.perf_counter() - t0, 1)
    except Exception as e:
        dropin = dict(error=str(e)[:300], run_seconds=round(time.perf_counter() - t0, 1))
    return others, realtime, dropin


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=None, help="streams per GPU (default: the config's own count -- config 2 = BASELINE configs[1]: 256)")
    ap.add_argument("--seconds", type=float, default=10.0, help="seconds of input per stream per step")
    ap.add_argument("--stretch", type=float, default=1.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-self-check", action="store_true", help="skip the comparison of stream 0's first call with oracle/_ref")
    ap.add_argument("--no-serial-pass", action="store_true", help="skip the extra, serialised per-kernel-class profiling call (used under rocprofv3, so that every launch it sees is an in-place one)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with one rank: exercises the barrier / max-over-ranks path of the N>1 runs on a 1-GPU box")
    ap.add_argument("--oversubscribe", action="store_true", help="N ranks on ONE GPU (every LOCAL_RANK maps to device 0): exercises the N>1 launch path -- rendezvous, barrier, "
                    "all_reduce(MAX/SUM), NUMA pinning, engines sharing a device -- where only one GPU exists.  The line says so in config.sharding; it is not a scaling measurement")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL (the multi-GPU default); gloo only for --oversubscribe if RCCL refuses two ranks on one device")
    ap.add_argument("--as-rank", type=int, default=None, help="single-process run with the stream indices and seed of this rank (cross-check of an --oversubscribe run)")
    ap.add_argument("--dump-output", default=None, help="write rank-local output of the first streams to this .npy prefix (oversubscribe cross-check)")
    ap.add_argument("--no-numa-pinning", action="store_true", help="with --gpus > 1: leave every rank's CPU affinity as the launcher set it (default: pin each rank's host scheduler to the CPUs of its GPU's NUMA node, and FAIL if that node cannot be read)")
    ap.add_argument("--half-state", action="store_true", help="BASELINE config 5 'fp16 internal': carried state and overlap-add sums stored in fp16 (SMST_FLAG_HALF_STATE)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of configs 3 / 4b / 5 and of the real-time pattern that the default (config 2, one GPU) "
                    "invocation appends to its line under `other_configs` / `realtime` (each in a process of its own, AFTER the timed region)")
    ap.add_argument("--config", default="2", choices=["2", "3", "4", "4b", "5"],
                    help="BASELINE.json config (default 2 = the one the headline metric is quoted on; the others are "
                         "reported in DESIGN.md, they are not the bench line)")
    args = ap.parse_args()
    preset, C, sr_cfg = "default", 2, SR
    setup = None
    per_stream = None
    explicit_streams = args.streams
    if args.streams is None:
        args.streams = 256
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
    if explicit_streams is not None:  # e.g. the PMC passes of the 1024-stream configs run on a subset (rocprofv3's counter collection crashes on the full ones)
        args.streams = explicit_streams

    import torch
    import torch.distributed as dist
    if args.gpus < 1:
        raise SystemExit("bench: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        cmd, env = self_launch_command(args.gpus, args.oversubscribe, torch.cuda.device_count(), sys.argv[1:], os.environ)
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    stream_rank = args.as_rank if (args.as_rank is not None and world == 1) else rank
    if world != args.gpus:
        raise SystemExit("bench: WORLD_SIZE (%d) != --gpus (%d): launch with --nproc-per-node equal to --gpus" % (world, args.gpus))
    if not args.oversubscribe and torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench: rank %d has no GPU of its own (LOCAL_RANK %d, %d visible)" % (rank, local_rank, torch.cuda.device_count()))
    dev_index = 0 if args.oversubscribe else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_device = device if args.dist_backend == "nccl" else torch.device("cpu")
    dist_world = dist.get_world_size() if use_dist else None
    if use_dist and dist_world != args.gpus:
        raise SystemExit("bench: the process group has %d ranks, --gpus says %d" % (dist_world, args.gpus))

    # one line per rank about where its host scheduler runs: PCI address and NUMA node of its GPU, the CPU set it was pinned to.  With
    # more than one rank a device whose NUMA node cannot be read is an error (--no-numa-pinning: run where the launcher put the ranks)
    if world > 1 and not args.no_numa_pinning:
        placement = pin_rank_to_numa_node(dev_index, world, share_index=rank if args.oversubscribe else local_rank, strict=not args.oversubscribe)
    else:
        placement = dict(device=dev_index, pci=device_pci_address(dev_index), numa_node=None, cpus=cpu_ranges(os.sched_getaffinity(0)), pinned=False)
    placement["rank"] = rank
    placements = [placement]
    if use_dist:
        placements = [None]*dist_world
        dist.all_gather_object(placements, placement)
    if world > 1:
        print("bench: rank %d -> device %d (pci %s, NUMA node %s), host scheduler on CPUs %s%s" % (
            rank, dev_index, placement["pci"], placement["numa_node"], placement["cpus"], "" if placement["pinned"] else " (NOT pinned)"), file=sys.stderr)
    pkg = importlib.import_module("signalsmith-stretch_amd")
    S = args.streams
    n_in = int(args.seconds*sr_cfg)
    n_out = int(round(n_in*args.stretch))
    batch = pkg.StretchBatch(S, C, preset=preset, sample_rate=sr_cfg, device=dev_index, seed=stream_rank*S, half_state=args.half_state)  # global stream g carries the engine of seed g
    if setup:
        setup(batch)
    if per_stream:  # config 5: per-stream random stretch 0.75-1.5x and +-12 st
        import numpy as np
        g = np.random.Generator(np.random.PCG64(5))
        stretches = g.uniform(0.75, 1.5, 8192)[stream_rank*S:(stream_rank + 1)*S]
        semis = g.uniform(-12, 12, 8192)[stream_rank*S:(stream_rank + 1)*S]
        for i in range(S):
            batch.setTransposeSemitones(float(semis[i]), 0.0, stream=i)
        n_out = [int(round(n_in*float(v))) for v in stretches]
    x = make_inputs(torch, S, C, n_in, device, first_stream=stream_rank*S, sr=sr_cfg)
    n_out_max = max(n_out) if per_stream else n_out
    y = torch.empty((S, C, n_out_max), dtype=torch.float32, device=device)
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    first_call = None
    for w in range(args.warmup):
        batch.process(x, n_out, out=y, ordered=False)
        if w == 0 and rank == 0 and not args.no_self_check:  # stream 0 of the batch's first call, for the self-check below (untimed)
            batch.synchronize()
            first_call = (y[0].cpu().numpy().copy(), x[0].cpu().numpy().copy(), (n_out[0] if per_stream else n_out))
    batch.synchronize()
    barrier()
    torch.cuda.synchronize()
    # the recurrence kernel is timed IN PLACE over the timed region: a HIP-event pair on the stream it is launched on
    # (the engine's chain stream), the other streams keep overlapping it (two event records per 64-hop tile)
    batch.enableProfiling(0 if os.environ.get("SMST_BENCH_NO_LIVE") else 2)
    # per-step completion stamps: a side stream waits (by event) for the batch's stream after every call and records a timing
    # event -- the steps stay pipelined (no host synchronisation between them), the stamps give each step's device-side period
    side = torch.cuda.Stream(device=device)
    stamps = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    pkg_lib = batch.lib
    import ctypes

    def stamp(i):
        pkg_lib.smst_batch_signal_stream(batch.h, ctypes.c_void_p(side.cuda_stream))
        stamps[i].record(side)
    stamp(0)
    t0 = time.perf_counter()
    # inputs are complete (synchronised above) and the outputs are only looked at after batch.synchronize(): no per-call
    # ordering against torch's stream, so the host scheduling of step n+1 overlaps the kernels of step n
    host_wall = 0.0
    batch.takeHostTimes()  # (zero the engine's own host clock)
    host_cpu0 = time.thread_time()
    for i in range(args.steps):
        tc = time.perf_counter()
        batch.process(x, n_out, out=y, ordered=False)
        host_wall += time.perf_counter() - tc
        stamp(i + 1)
    host_cpu = time.thread_time() - host_cpu0
    host_engine = batch.takeHostTimes()
    batch.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    live_ms, live_launches = batch.takeTimings()
    batch.enableProfiling(0)
    step_periods = [stamps[i].elapsed_time(stamps[i + 1]) for i in range(args.steps)]  # in order: a slow FIRST step is a warm-up artefact, a slow one in the middle is jitter
    step_ms = sorted(step_periods)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ok = bool(torch.isfinite(y).all().item()) and float(y.abs().max().item()) > 0.01
    if args.dump_output:
        import numpy as np
        np.save("%s.rank%d.npy" % (args.dump_output, stream_rank), y[:4].cpu().numpy())

    total_out = sum(n_out) if per_stream else S*n_out
    samples_local = C*(S*n_in + total_out)
    if use_dist:  # per-stream stretch factors differ between ranks (config 5): add up what every rank really processed
        tot = torch.tensor([samples_local], dtype=torch.float64, device=red_device)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        samples_per_step = int(tot.item())
    else:
        samples_per_step = samples_local
    value = samples_per_step*args.steps/elapsed/1e6
    B, I, M = batch.blockSamples(), batch.intervalSamples(), batch.bands()
    fft_samples = batch.fftSamples()
    hops_per_stream = -(-(total_out//S)//I)
    bytes_per_chop = algorithmic_bytes_per_channel_hop(B, I, M, args.stretch)
    own = own_algorithmic_bytes(B, I, M, args.stretch)
    chops_per_step = S*C*hops_per_stream

    roofline = None
    if rank == 0:
        # per-kernel-class device time, each class ALONE on the GPU: one extra, serialised call with HIP events around every launch
        if args.no_serial_pass:
            ms = {k: 0.0 for k in ("analyse", "feed", "predict", "chain", "synth", "emit", "other")}
            ms["chain"] = live_ms.get("chain_live", 0.0)/max(args.steps, 1)
            launches = {k: 0 for k in ("analyse", "predict", "chain", "synth", "emit")}
            launches["chain"] = int(live_launches.get("chain_live", 0)/max(args.steps, 1))
        else:
            batch.enableProfiling(1)
            batch.process(x, n_out, out=y, ordered=False)
            batch.synchronize()
            ms, launches = batch.takeTimings()
            batch.enableProfiling(0)
            ms.pop("chain_live", None)
            launches.pop("chain_live", None)
        teams = M in (2560, 3072) and os.environ.get("SMST_FFT_TEAMS", "1") != "0"  # persistent-team FFT kernels (the 48-kHz presets' geometries)
        names = {"analyse": "kAnalyseTeams" if teams else ("kAnalyseFast" if M in (5120, 6144, 2560, 3072) else "kAnalyse"), "predict": "kPredictA/B",
                 "chain": "kVocoder" if C <= 2 else "kVocoderN",
                 "synth": "kSynthTeams" if teams else ("kSynthFast" if M in (5120, 6144, 2560, 3072) else "kSynth"), "emit": "kEmit"}
        launch_count = {k: launches[k] for k in ("analyse", "predict", "chain", "synth", "emit")}
        one_kernel = teams and launch_count["emit"] == 0 and launch_count["synth"] > 0  # synthesis + overlap-add + emission in kSynthEmitTeams
        if one_kernel:
            names["synth"] = "kSynthEmitTeams"

        dom = max(launch_count, key=lambda k: ms[k])  # the class with the largest stand-alone time per step
        avg_ms_alone = ms[dom]/max(launch_count[dom], 1) or float("nan")
        avg_ms_in_place = avg_ms_alone
        if dom == "chain" and live_launches.get("chain_live", 0) > 0:  # in place, over the timed steps (agrees with rocprofv3 --stats)
            avg_ms_in_place = live_ms["chain_live"]/live_launches["chain_live"]
        chops_per_launch = chops_per_step/max(launch_count[dom], 1)
        own_key = {"analyse": "analyse", "chain": "chain", "synth": "synth+emit", "emit": "synth+emit", "predict": "chain"}[dom]

        def gbs(nbytes, millis):
            return nbytes/(millis*1e-3)/1e9 if millis == millis and millis > 0 else None
        kernel_achieved = gbs(own[own_key]*chops_per_launch, avg_ms_in_place)
        whole_bytes_on_kernel = gbs(bytes_per_chop*chops_per_launch, avg_ms_in_place)
        step_mean_ms = elapsed/args.steps*1e3
        pipeline_achieved = gbs(bytes_per_chop*chops_per_step, step_mean_ms)
        sha = library_sha16()
        traffic, traffic_source = measured_traffic(sha, args.config)
        per_class = {}
        rows = [("analyse", names["analyse"], ms["analyse"], launch_count["analyse"]), ("chain", names["chain"], ms["chain"], launch_count["chain"]),
                ("synth+emit", names["synth"] + ("" if one_kernel else " + kEmit"), ms["synth"] + ms["emit"], launch_count["synth"])]
        for key, label, alone, count in rows:
            if count:
                per_class[label] = dict(ms_per_step_alone=round(alone, 3), launches_per_step=count, own_algorithmic_bytes_per_channel_hop=own[key],
                                        own_frac_alone=gbs(own[key]*chops_per_step, alone)/HBM_PEAK_GBS)
        roofline = dict(
            bound="hbm", scope="whole hot path: every kernel of one process() step, pipelined (the figure north_star's '% of HBM peak' asks for)",
            achieved=pipeline_achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=(pipeline_achieved/HBM_PEAK_GBS if pipeline_achieved else None),
            traffic=((traffic.get("bytes_per_step")*(S/float(traffic["streams"]) if traffic.get("streams") else 1.0)) if traffic else None), traffic_source=traffic_source,
            algorithmic_bytes_per_channel_hop=bytes_per_chop, channel_hops_per_step=chops_per_step,
            step_ms=dict(mean_wall=step_mean_ms, median=step_ms[len(step_ms)//2], min=step_ms[0], max=step_ms[-1], n=len(step_ms), in_order=[round(v, 3) for v in step_periods],
                         note="mean_wall = host clock around the K steps / K (the figure `value` uses); median/min/max = device-side period of each step (event stamps)"),
            pipeline_frac=(pipeline_achieved/HBM_PEAK_GBS if pipeline_achieved else None),
            dominant_kernel=dict(
                kernel=names[dom], chosen_by="largest stand-alone time per step", avg_launch_ms=avg_ms_in_place, avg_launch_ms_alone=avg_ms_alone,
                launches_per_step=launch_count[dom], channel_hops_per_launch=chops_per_launch,
                own_algorithmic_bytes_per_channel_hop=own[own_key], kernel_achieved=kernel_achieved,
                kernel_frac=(kernel_achieved/HBM_PEAK_GBS if kernel_achieved else None),
                whole_path_bytes_over_this_kernel_frac=(whole_bytes_on_kernel/HBM_PEAK_GBS if whole_bytes_on_kernel else None),
                note="kernel_frac prices this kernel's launch time with ITS OWN share of the algorithmic bytes; the last field is the round-1/2 "
                     "figure (whole-path bytes over one kernel's time), kept for continuity -- it is not a statement about this kernel",
                traffic=(traffic.get("kernels", {}).get(names[dom]) if traffic else None)),
            kernels=per_class, kernel_ms_per_step_alone={k: round(v, 3) for k, v in ms.items()}, library_sha16=sha)
    others, realtime, dropin = None, None, None
    if rank == 0 and world == 1 and args.config == "2" and explicit_streams is None and not args.no_other_configs:
        # The other BASELINE configs and the real-time calling pattern under the same clock as the headline: short runs, each in a process
        # of its own, after the timed region (the headline's batch is released first).  Reported, never part of `value`.
        batch.close()
        del x, y
        torch.cuda.empty_cache()
        others, realtime, dropin = other_measurements()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline()
            if cpu is not None:
                cpu["cmd_binary"] = cmd_binary_baseline(host_topology()[1])
        except Exception as e:  # the baseline is reported, never required
            cpu = dict(error=str(e))
    check = None
    if rank == 0:
        if first_call is None:
            check = dict(checked=False, reason="--no-self-check" if args.no_self_check else "no warm-up call to take the first output from")
        else:
            try:
                ref_setup = None
                if setup or per_stream:
                    def ref_setup(r):
                        if setup:
                            setup(r)
                        if per_stream:
                            r.setTransposeSemitones(float(semis[0]), 0.0)
                check = self_check(first_call[0], first_call[1], first_call[2], C, sr_cfg, preset, ref_setup, config=args.config)
            except Exception as e:
                check = dict(checked=False, reason="self-check failed to run: %s" % e)
    if rank == 0:
        sharding = "streams/%d, no collective" % world
        if args.oversubscribe:
            sharding = "OVERSUBSCRIBED: %d ranks on ONE device (cuda:0), %d streams each, no collective on the data path; %s for barrier + all_reduce" % (world, S, args.dist_backend)
        line = {
            "metric": "Msamples/sec (in+out) at 48kHz stereo presetDefault",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "dist_world_size": dist_world, "dist_backend": (args.dist_backend if use_dist else None),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed/args.steps*1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "state_storage": "f16" if args.half_state else "f32",
            "config": {"workload": ("BASELINE configs[1]: %d stereo streams per GPU, 48 kHz, presetDefault, %.2fx stretch, fp32, "
                                    "%.0f s input per stream per step, device-resident I/O" % (S, args.stretch, args.seconds))
                       if args.config == "2" else "BASELINE config %s (not the headline): %d streams x %d ch per GPU, %d Hz, preset %s, %.0f s per step"
                       % (args.config, S, C, sr_cfg, preset, args.seconds),
                       "streams_total": world*S, "channels": C, "block": B, "interval": I, "fft": fft_samples,
                       "hops_per_stream_per_step": hops_per_stream, "sharding": sharding,
                       "ranks": placements},
            "realtime_x": world*S*args.seconds*args.steps/elapsed,
            "host_ms_per_step": {"work": host_engine["work_ms"]/args.steps, "waiting_for_the_device": (host_engine["wait_tables_ms"] + host_engine["wait_gate_ms"])/args.steps,
                                 "of_which_silence_gate_readback": host_engine["wait_gate_ms"]/args.steps,
                                 "in_process_call": host_wall/args.steps*1e3, "cpu": host_cpu/args.steps*1e3, "threads": 1,
                                 "note": "per step, on the ONE host thread of this rank: `work` = the engine's own host work inside process() (the per-stream block scheduler, "
                                         "table fills, uploads and kernel enqueues: smst_batch_take_host_times); `waiting_for_the_device` = back-pressure (the host runs at most "
                                         "two calls ahead of the device) plus the silence gate's 64-byte-per-stream readback -- HIP spins while it waits, so this shows up as "
                                         "CPU time without being work; in_process_call / cpu = wall and CPU time of the calling thread inside the Python call.  `work` is what "
                                         "must stay below ms_per_step; 8 ranks need 8 such threads on cores of their own (bench.py pins each to its GPU's NUMA node)"},
            "other_configs": others, "realtime": realtime, "dropin_objects_vs_batch": dropin,
            "channels": C,
            "output_finite_nonzero": ok, "self_check": check,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()
    if rank == 0 and not ok:
        raise SystemExit("bench: output not finite / silent -- the line above is not a valid measurement")
    if rank == 0 and check and check.get("checked") and not check.get("ok"):
        raise SystemExit("bench: stream 0 of the first call disagrees with the checker (%s) -- the line above is not a valid measurement" % check)


if __name__ == "__main__":
    main()
