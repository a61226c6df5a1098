#!/bin/bash
# presetDefault @ 44.1 kHz: analysis by teams (default) against one frame per workgroup (SMST_FFT_TEAMS=0), same box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for t in 1 0; do
SMST_FFT_TEAMS=$t python - <<'PY'
import importlib, os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("signalsmith-stretch_amd")
S, C, sr = 256, 2, 44100
n_in = int(10*sr); n_out = int(round(n_in*1.5))
x = bench.make_inputs(torch, S, C, n_in, torch.device("cuda", 0), sr=sr)
y = torch.empty((S, C, n_out), dtype=torch.float32, device="cuda")
b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr, device=0)
for _ in range(2): b.process(x, n_out, out=y, ordered=False)
b.synchronize(); t0 = time.perf_counter()
for _ in range(4): b.process(x, n_out, out=y, ordered=False)
b.synchronize(); dt = (time.perf_counter() - t0)/4
b.enableProfiling(1); b.process(x, n_out, out=y, ordered=False); b.synchronize(); ms, _ = b.takeTimings()
print("SMST_FFT_TEAMS=%s: %.0f Msamples/s, %.2f ms/step, alone %s" % (os.environ["SMST_FFT_TEAMS"], S*C*(n_in + n_out)/dt/1e6, dt*1e3, {k: round(v, 2) for k, v in ms.items() if v > 0.3}))
PY
done
