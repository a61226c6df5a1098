"""Reads the per-block timestamps an instrumented build of kVocoder leaves in its trace buffer (one workgroup, first 400
blocks: producer 0, the recurrence wave, the writer) and prints where each of them spends a block.  Only meaningful with a
library built from the instrumented sources (see EXPERIMENTS.md, "what bounds kVocoder" and "Timeline of one block"); a product build has no
selector 7 and this script fails."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
import torch
pkg = importlib.import_module("signalsmith-stretch_amd")

S, CH, sr = 256, 2, 48000
n_out = 256*1440          # 256 hops: four FULL 64-hop tiles (the trace keeps the last launch; a partial tile leaves most producers idle)
n_in = n_out*2//3
b = pkg.StretchBatch(S, CH, preset="default", sample_rate=float(sr))
kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
if kind == "noise":
    x = (torch.rand(S, CH, n_in, device="cuda") - 0.5)*0.6
else:  # the bench's sine streams: two partials, a phase offset per channel
    tt = torch.arange(n_in, device="cuda", dtype=torch.float64)/sr
    f1 = 110.0*2.0**((torch.arange(S, device="cuda") % 37).double()/12.0)
    ph = 0.5*torch.arange(CH, device="cuda").double()
    x = (0.4*torch.sin(2*np.pi*f1[:, None, None]*tt[None, None, :] + ph[None, :, None]) + 0.2*torch.sin(2*np.pi*3.17*f1[:, None, None]*tt[None, None, :])).float().contiguous()
y = torch.empty(S, CH, n_out, device="cuda")
for _ in range(2):
    b.process(x, n_out, out=y, ordered=False)
torch.cuda.synchronize()
buf = np.zeros(14*400 + 8, np.uint64)
rc = b.lib.smst_batch_debug_get_state(b.h, 0, 7, buf.ctypes.data_as(C.POINTER(C.c_float)))
assert rc == 0
t = buf[:14*400].reshape(14, 400).astype(np.int64)
wall = buf[14*400:].astype(np.int64)  # 100 MHz wall clock at blocks 100 and 300 of the recurrence wave
lo, hi = 100, 300
def d(a, bb):
    return float(np.mean(t[bb, lo:hi] - t[a, lo:hi]))
def period(a):
    return float(np.mean(np.diff(t[a, lo:hi])))
print("clock ticks per block (mean over blocks %d..%d)" % (lo, hi))
if wall[1] > wall[0]:
    ns = (wall[1] - wall[0])*10.0
    cyc = float(t[5, 300] - t[5, 100])
    print("blocks 100..300: %.1f us wall, %.0f shader-clock ticks -> %.2f GHz, %.2f us per block" % (ns/1e3, cyc, cyc/ns, ns/200e3))
if t[13, lo:hi].min() > 0:  # the aligned producers' trace: an explicit wait for the loads in front of the parking
    print("producer 0 : period %.0f | wait loads %.0f | park+barrier %.0f | issue %.0f | slot wait %.0f | compute %.0f | record write %.0f | rest %.0f" % (
        period(0), d(0, 13), d(13, 1), d(1, 12), d(12, 2), d(2, 3), d(3, 4), period(0) - d(0, 4)))
else:
    print("producer 0 : period %.0f | park+barrier %.0f | issue %.0f | slot wait %.0f | compute %.0f | record write %.0f | rest %.0f" % (
        period(0), d(0, 1), d(1, 12), d(12, 2), d(2, 3), d(3, 4), period(0) - d(0, 4)))
print("recurrence : period %.0f | wait records %.0f | wait writer %.0f | 8 steps %.0f | rest %.0f" % (
    period(5), d(5, 6), d(6, 7), d(7, 8), period(5) - d(5, 8)))
print("writer     : period %.0f | wait %.0f | stores %.0f" % (period(9), d(9, 10), d(10, 11)))
# skew between the three
print("records of block n ready -> recurrence starts it: %.0f ; recurrence done -> writer done: %.0f" % (
    float(np.mean(t[7, lo:hi] - t[4, lo:hi])), float(np.mean(t[11, lo:hi] - t[8, lo:hi]))))
