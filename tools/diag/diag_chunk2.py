"""Diagnostic (not a test): which carried state differs between one call and hop-aligned chunked calls, hop by hop."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_input
import torch
pkg = importlib.import_module("signalsmith-stretch_amd")
C, sr, S = 2, 48000, 2
x = torch.from_numpy(np.stack([synth_input(s, C, 28800, sr) for s in range(S)])).cuda()
names = ["input", "prevInput", "output", "energy"]
for hops in (5, 6, 7):
    b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr)
    yw = b.process(x[:, :, :1152*hops].contiguous(), 1440*hops); b.synchronize()
    sw = [b.debug_state(0, w) for w in range(4)]
    cw = b.debug_carry(0)
    b.close()
    b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr)
    y1 = b.process(x[:, :, :5760].contiguous(), 7200)
    parts = [y1]
    if hops > 5:
        parts.append(b.process(x[:, :, 5760:1152*hops].contiguous(), 1440*(hops - 5)))
    b.synchronize()
    sc = [b.debug_state(0, w) for w in range(4)]
    cc = b.debug_carry(0)
    b.close()
    yc = torch.cat(parts, dim=2)
    print("hops", hops, "out diff %.3e" % float((yw - yc).abs().max()),
          {n: "%.3e" % float(np.abs(a - c).max()) for n, a, c in zip(names, sw, sc)},
          "carry sums %.3e wp %.3e" % (float(np.abs(cw[0] - cc[0]).max()), float(np.abs(cw[1] - cc[1]).max())))
    d = np.abs(sw[2] - sc[2]).reshape(C, -1, 2).max(axis=2)
    for c in range(C):
        idx = np.nonzero(d[c])[0]
        print("   channel", c, "differing bins:", len(idx), idx[:24], "max at", int(d[c].argmax()), "value", float(np.abs(sw[2]).reshape(C, -1, 2)[c, int(d[c].argmax())].max()))
