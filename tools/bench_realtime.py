#!/usr/bin/env python3
"""Latency of the real-time calling pattern: every render quantum (128 frames at 48 kHz =
2.67 ms) each stream gets process(128 in, 128 out), device-resident buffers, one synchronisation per quantum.
Prints one JSON object: per batch size the median / p99 time per quantum and how many streams that sustains in real time.
This pattern is bound by kernel launches and the per-call host scheduling, not by the kernels."""
import argparse, importlib, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # the engine's pipeline streams get hardware queues of their own; before the HIP runtime starts
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quantum", type=int, default=128)
    ap.add_argument("--quanta", type=int, default=600)
    ap.add_argument("--streams", type=int, nargs="*", default=[1, 64, 256, 1024, 4096])
    ap.add_argument("--stretch", type=float, default=1.0)
    ap.add_argument("--preset", default="default", choices=["default", "cheaper"], help="cheaper = split computation, the preset real-time hosts use (signalsmith-stretch.h:66-68)")
    ap.add_argument("--channels", type=int, default=2)
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("signalsmith-stretch_amd")
    sr, C, Q = 48000, args.channels, args.quantum
    n_in = int(round(Q/args.stretch))
    rows = []
    for S in args.streams:
        b = pkg.StretchBatch(S, C, preset=args.preset, sample_rate=sr)
        t = torch.arange(n_in*(args.quanta + 50), device="cuda", dtype=torch.float32)/sr
        x = (0.4*torch.sin(2*torch.pi*220.0*t)).expand(S, C, -1).contiguous()
        y = torch.empty((S, C, Q), dtype=torch.float32, device="cuda")
        times = []
        for q in range(args.quanta + 50):
            xin = x[:, :, q*n_in:(q + 1)*n_in]
            t0 = time.perf_counter()
            b.process(xin, Q, out=y, ordered=False)  # inputs complete, the batch is synchronised right below
            b.synchronize()
            if q >= 50:
                times.append(time.perf_counter() - t0)
        b.close()
        times.sort()
        med, p99 = times[len(times)//2], times[int(len(times)*0.99)]
        budget = Q/sr
        rows.append({"streams": S, "median_ms": round(med*1e3, 4), "p99_ms": round(p99*1e3, 4), "mean_ms": round(sum(times)/len(times)*1e3, 4),
                     "realtime_budget_ms": round(budget*1e3, 4), "realtime_headroom_x": round(budget/p99, 2),
                     "Msamples_per_s": round(S*C*(n_in + Q)/(sum(times)/len(times))/1e6, 2)})
    print(json.dumps({"pattern": "process(%d, %d) per quantum, %d ch 48 kHz preset %s, device-resident, 1 sync per quantum" % (n_in, Q, C, args.preset),
                      "stretch": args.stretch, "quanta_timed": args.quanta, "rows": rows}))


if __name__ == "__main__":
    main()
