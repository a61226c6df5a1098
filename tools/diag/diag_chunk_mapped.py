"""Diagnostic (not a test): one call vs hop-aligned chunks on the mapped (pitch / formant) path, per chunk."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_input
import torch
pkg = importlib.import_module("signalsmith-stretch_amd")
S, C, sr = 8, 2, 48000
x = torch.from_numpy(np.stack([synth_input(s, C, 28800, sr) for s in range(S)])).cuda()
for label, setup in (("pitch", lambda b: b.setTransposeSemitones(5, 8000/48000)),
                     ("pitch+formants", lambda b: (b.setTransposeSemitones(5, 8000/48000), b.setFormantFactor(0.9, True), b.setFormantBase(150/48000)))):
    b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr); setup(b)
    whole = b.process(x, 36000); b.synchronize(); whole = whole.clone(); b.close()
    b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr); setup(b)
    parts = [b.process(x[:, :, 5760*k:5760*(k + 1)].contiguous(), 7200) for k in range(5)]
    b.synchronize(); parts = [p.clone() for p in parts]; b.close()
    d = (whole - torch.cat(parts, dim=2)).abs()
    print(label, "peak %.3f" % float(whole.abs().max()), "max", float(d.max()), "per chunk", [float("%.2e" % float(d[:, :, 7200*k:7200*(k + 1)].max())) for k in range(5)],
          "per stream", [float("%.1e" % float(d[s].max())) for s in range(S)])
