"""Diagnostic (not a test): whole-call vs hop-aligned chunked calls on the GPU, per stream / per hop."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_input
import torch
pkg = importlib.import_module("signalsmith-stretch_amd")
C, sr = 2, 48000
for S in (4, 256):
    x = torch.from_numpy(np.stack([synth_input(s % 12, C, 28800, sr) for s in range(S)])).cuda()
    b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr)
    whole = b.process(x, 36000); b.synchronize()
    outs = []
    for rep in range(2):
        b.reset()
        parts = [b.process(x[:, :, 5760*k:5760*(k + 1)].contiguous(), 7200) for k in range(5)]
        b.synchronize()
        outs.append(torch.cat(parts, dim=2))
    b.reset()
    whole2 = b.process(x, 36000); b.synchronize()
    d = (whole - outs[0]).abs()
    per_stream = d.amax(dim=(1, 2)).cpu().numpy()
    bad = np.nonzero(per_stream > 1e-6)[0]
    print("S=%d: whole-vs-chunked max %.3e; chunked rep diff %.3e; whole rep diff %.3e; bad streams %s" % (
        S, float(d.max()), float((outs[0] - outs[1]).abs().max()), float((whole - whole2).abs().max()), bad[:10]))
    if len(bad):
        s = int(bad[0])
        print("   stream", s, "per-hop max diff:", [float("%.1e" % float(d[s][:, a*1440:(a + 1)*1440].max())) for a in range(25)])
    b.close()
