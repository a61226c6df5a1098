"""Per-sample rate of the hot path at every preset geometry (signalsmith-stretch.h:63-68): presetDefault / presetCheaper at 48 and 96 kHz,
stereo, 1.5x, 256 streams -- with the register-blocked FFT kernels and (SMST_NO_FAST_FFT=1) with the generic radix-4/2/3/5 ladder.
VERDICT round 2, item 7: the other presets within 15 % of the 48-kHz presetDefault per-sample rate.
usage: python tools/bench_presets.py > profiles/rN_presets.json"""
import importlib
import json
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # the engine's pipeline streams get hardware queues of their own; before the HIP runtime starts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    pkg = importlib.import_module("signalsmith-stretch_amd")
    S, C, steps, warm = 256, 2, 4, 2
    rows = []
    table = (("default", 48000, 10.0), ("cheaper", 48000, 10.0), ("default", 96000, 5.0), ("cheaper", 96000, 5.0), ("default", 44100, 10.0),
             # (round 6) 192 kHz: 12288 / 10240 bands, the generic FFT ladder with its second buffer in memory -- the only form there (both columns are it)
             ("default", 192000, 2.5), ("cheaper", 192000, 2.5))
    only = [a for a in sys.argv[1:] if not a.startswith("-")]  # e.g. `bench_presets.py default48000 cheaper48000`: a subset
    fast_only = "--fast-only" in sys.argv
    for preset, sr, seconds in table:
        if only and "%s%d" % (preset, sr) not in only:
            continue
        n_in = int(seconds*sr)
        n_out = int(round(n_in*1.5))
        x = bench.make_inputs(torch, S, C, n_in, torch.device("cuda", 0), sr=sr)
        y = torch.empty((S, C, n_out), dtype=torch.float32, device="cuda")
        row = dict(preset=preset, sample_rate=sr, seconds_per_stream=seconds)
        for generic in ((False,) if fast_only else (False, True)):
            if generic:
                os.environ["SMST_NO_FAST_FFT"] = "1"
            else:
                os.environ.pop("SMST_NO_FAST_FFT", None)
            b = pkg.StretchBatch(S, C, preset=preset, sample_rate=sr, device=0)
            for _ in range(warm):
                b.process(x, n_out, out=y, ordered=False)
            b.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                b.process(x, n_out, out=y, ordered=False)
            b.synchronize()
            dt = (time.perf_counter() - t0)/steps
            b.enableProfiling(1)
            b.process(x, n_out, out=y, ordered=False)
            b.synchronize()
            ms, _ = b.takeTimings()
            key = "generic_fft" if generic else "fast_fft"
            row[key] = dict(Msamples_s=C*S*(n_in + n_out)/dt/1e6, ms_per_step=dt*1e3, alone_ms={k: round(ms[k], 3) for k in ("analyse", "chain", "synth", "emit")})
            row["bands"], row["block"], row["interval"] = b.bands(), b.blockSamples(), b.intervalSamples()
            b.close()
        os.environ.pop("SMST_NO_FAST_FFT", None)
        rows.append(row)
        del x, y
    ref = rows[0]["fast_fft"]["Msamples_s"]
    for r in rows:
        r["rate_vs_default_48k"] = r["fast_fft"]["Msamples_s"]/ref
    print(json.dumps(dict(workload="256 stereo streams, 1.5x, device-resident I/O, one MI355X", rows=rows), indent=1))


if __name__ == "__main__":
    main()
