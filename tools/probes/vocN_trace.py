"""Per-block timeline of kVocoderN (8 channels, presetCheaper at 96 kHz, pitch-mapped = BASELINE config 5's kernel) from an
instrumented build (see voc_trace.py).  Producer 0: one pass = 16 rows x 4 steps; the recurrence wave: blocks of 4 steps."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib, torch
pkg = importlib.import_module("signalsmith-stretch_amd")
S, CH, sr = 256, 8, 96000
b = pkg.StretchBatch(S, CH, preset="cheaper", sample_rate=float(sr))
I = b.intervalSamples()
n_out = 64*I            # one full 64-hop tile ... twice
n_in = n_out
for i in range(S):
    b.setTransposeSemitones(float(-12 + 24.0*i/S), 0.0, stream=i)
x = (torch.rand(S, CH, n_in, device="cuda") - 0.5)*0.6
y = torch.empty(S, CH, n_out, device="cuda")
for _ in range(2):
    b.process(x, n_out, out=y, ordered=False)
torch.cuda.synchronize()
buf = np.zeros(12*400 + 8, np.uint64)
assert b.lib.smst_batch_debug_get_state(b.h, 0, 7, buf.ctypes.data_as(C.POINTER(C.c_float))) == 0
t = buf[:12*400].reshape(12, 400).astype(np.int64)
wall = buf[12*400:].astype(np.int64)
lo, hi = 100, 300
d = lambda a, bb: float(np.mean(t[bb, lo:hi] - t[a, lo:hi]))
period = lambda a: float(np.mean(np.diff(t[a, lo:hi])))
if wall[1] > wall[0]:
    ns = (wall[1] - wall[0])*10.0; cyc = float(t[5, 300] - t[5, 100])
    print("blocks 100..300: %.1f us, %.2f GHz, %.2f us per 4-step block" % (ns/1e3, cyc/ns, ns/200e3))
print("producer 0 (per pass)  : period %.0f | record compute %.0f | slot wait %.0f" % (period(0), d(1, 2), d(2, 3)))
print("   inside the pass      : first round of loads back %.0f | second round back %.0f | rest %.0f" % (d(1, 9), d(9, 10), d(10, 2)))
print("recurrence (per block) : period %.0f | wait records %.0f | wait writer %.0f | 4 steps %.0f | rest %.0f" % (period(5), d(5, 6), d(6, 7), d(7, 8), period(5) - d(5, 8)))
