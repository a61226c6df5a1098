"""Phase durations inside kAnalyseFast workgroups (every 97th workgroup of the last launch), from an instrumented build:
python tools/probes/fft_trace.py   (SMST_NO_OVERLAP=1 for the kernel alone)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib, torch
pkg = importlib.import_module("signalsmith-stretch_amd")
S, CH, sr = 256, 2, 48000
n_out = 256*1440
n_in = n_out*2//3
b = pkg.StretchBatch(S, CH, preset="default", sample_rate=float(sr))
x = (torch.rand(S, CH, n_in, device="cuda") - 0.5)*0.6
y = torch.empty(S, CH, n_out, device="cuda")
for _ in range(2):
    b.process(x, n_out, out=y, ordered=False)
torch.cuda.synchronize()
buf = np.zeros(12*400 + 8, np.uint64)
assert b.lib.smst_batch_debug_get_state(b.h, 0, 7, buf.ctypes.data_as(C.POINTER(C.c_float))) == 0
t = buf[:12*400].reshape(12, 400).astype(np.int64)
ok = (t[0] > 0) & (t[7] > t[0])
print("workgroups sampled:", int(ok.sum()))
names = ["loads back", "stage A + LDS write", "barrier", "stage B (LDS read, butterflies, write)", "barrier", "stage C + stores issued", "stores drained"]
for i, nme in enumerate(names):
    dts = (t[i + 1] - t[i])[ok]
    print("  %-42s median %6.0f   mean %6.0f cycles" % (nme, np.median(dts), dts.mean()))
tot = (t[7] - t[0])[ok]
print("  %-42s median %6.0f   mean %6.0f cycles" % ("whole workgroup", np.median(tot), tot.mean()))
