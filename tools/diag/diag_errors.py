"""Diagnostic (not a test): per-hop rel-RMS of the product vs the checker for the chaotic cases, next to the
checker's own sensitivity to a 1e-7 relative input perturbation.  usage: python tools/diag/diag_errors.py [emu|hip]"""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_input, rel_rms  # noqa: E402
import ref_oracle  # noqa: E402
import scenarios  # noqa: E402

pkg = importlib.import_module("signalsmith-stretch_amd")
which = sys.argv[1] if len(sys.argv) > 1 else "emu"
lib = pkg.bind(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libsmst_emu.so"))) if which == "emu" else pkg.load_library()


def per_hop(a, b, I, first=2, count=16):
    return [float("%.1e" % rel_rms(a[:, h*I:(h + 1)*I], b[:, h*I:(h + 1)*I])) for h in range(first, min(first + count, b.shape[1]//I))]


def run(label, cfg, C, x, nout, setup):
    g = pkg.SignalsmithStretch(lib=lib); scenarios.configure(g, C, cfg); setup(g)
    r = ref_oracle.RefStretch(); scenarios.configure(r, C, cfg); setup(r)
    r2 = ref_oracle.RefStretch(); scenarios.configure(r2, C, cfg); setup(r2)
    u = np.random.default_rng(1).uniform(-1, 1, x.shape)
    y, o, o2 = g.process(x, nout), r.process(x, nout), r2.process((x*(1 + 1e-7*u)).astype(np.float32), nout)
    I = r.intervalSamples()
    print(label, "total %.2e (self %.2e)" % (rel_rms(y, o), rel_rms(o2, o)))
    print("   prod:", per_hop(y, o, I))
    print("   self:", per_hop(o2, o, I))


SMALL = dict(preset="configure", block=512, interval=128, split=False)
D48 = dict(preset="default", sample_rate=48000.0)
x = synth_input(0, 2, 9000, 48000) + 0.5*synth_input(4, 2, 9000, 48000)
run("pitch-7 small", SMALL, 2, x, 8100, lambda o: o.setTransposeSemitones(-7, 0))
run("pitch+12 small", SMALL, 2, x, 9000, lambda o: o.setTransposeSemitones(12, 8000/48000))
x = synth_input(0, 2, 20000, 48000) + 0.5*synth_input(4, 2, 20000, 48000)
run("formant-comp D48", D48, 2, x, 15000, lambda o: (o.setTransposeSemitones(4, 8000/48000), o.setFormantFactor(1, True), o.setFormantBase(200/48000)))
run("pitch-7 D48", D48, 2, x, 18000, lambda o: o.setTransposeSemitones(-7, 0))
for s in (1, 2, 5):
    xs = synth_input(s, 2, 30000, 48000)
    run("config2 stream %d" % s, D48, 2, xs, 45000, lambda o: None)
    run("config3 stream %d" % s, D48, 2, xs, 30000, lambda o: o.setTransposeSemitones(12, 8000/48000))
