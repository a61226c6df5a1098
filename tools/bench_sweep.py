"""Throughput of the headline workload as a function of the batch size on ONE GPU (the recurrence
kernel runs one workgroup per stream and one workgroup per CU, so its time is quantised in units of 256 streams).
Same inputs, step and timing as bench.py (config 2: stereo, 48 kHz, presetDefault, 1.5x, 10 s per stream per step).
usage: python tools/bench_sweep.py [--sizes 64,128,256,288,320,384,512] [--steps 5] > profiles/rN_stream_sweep.json"""
import argparse
import importlib
import json
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # the engine's pipeline streams get hardware queues of their own; before the HIP runtime starts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="64,128,192,256,288,320,384,512")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=10.0)
    args = ap.parse_args()
    import torch
    import bench
    pkg = importlib.import_module("signalsmith-stretch_amd")
    sizes = [int(v) for v in args.sizes.split(",")]
    sr, C = 48000, 2
    n_in = int(args.seconds*sr)
    n_out = int(round(n_in*1.5))
    device = torch.device("cuda", 0)
    x_all = bench.make_inputs(torch, max(sizes), C, n_in, device)
    rows = []
    for S in sizes:
        batch = pkg.StretchBatch(S, C, preset="default", sample_rate=sr, device=0)
        x = x_all[:S]
        y = torch.empty((S, C, n_out), dtype=torch.float32, device=device)
        for _ in range(args.warmup):
            batch.process(x, n_out, out=y, ordered=False)
        batch.synchronize()
        torch.cuda.synchronize()
        batch.enableProfiling(2)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            batch.process(x, n_out, out=y, ordered=False)
        batch.synchronize()
        dt = (time.perf_counter() - t0)/args.steps
        live_ms, live_n = batch.takeTimings()
        batch.enableProfiling(1)
        batch.process(x, n_out, out=y, ordered=False)
        batch.synchronize()
        ms, _ = batch.takeTimings()
        batch.enableProfiling(0)
        rows.append(dict(streams=S, ms_per_step=dt*1e3, Msamples_s=C*S*(n_in + n_out)/dt/1e6, us_per_stream_step=dt/S*1e6,
                         recurrence_ms_per_launch_in_place=live_ms["chain_live"]/max(live_n["chain_live"], 1),
                         alone_ms_per_step={k: round(ms[k], 3) for k in ("analyse", "chain", "synth", "emit")}))
        batch.close()
        del y
    ref = next((r for r in rows if r["streams"] == 256), rows[0])
    for r in rows:
        r["per_stream_rate_vs_256"] = ref["us_per_stream_step"]/r["us_per_stream_step"]
    print(json.dumps(dict(workload="config 2 (stereo, 48 kHz, presetDefault, 1.5x, %.0f s per stream per step), one MI355X" % args.seconds, rows=rows), indent=1))


if __name__ == "__main__":
    main()
