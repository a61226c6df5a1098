"""Diagnostic (not a test): staged vs gathered producers of the fused kernel, one call and two calls."""
import importlib, os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    from conftest import synth_input
    import torch
    pkg = importlib.import_module("signalsmith-stretch_amd")
    C, sr, S = 2, 48000, 2
    x = torch.from_numpy(np.stack([synth_input(s, C, 28800, sr) for s in range(S)])).cuda()
    out = {}
    b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr)
    b.process(x[:, :, :6912].contiguous(), 8640); b.synchronize()
    out["whole"] = b.debug_state(0, 2)
    b.close()
    b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr)
    b.process(x[:, :, :5760].contiguous(), 7200)
    b.process(x[:, :, 5760:6912].contiguous(), 1440); b.synchronize()
    out["chunked"] = b.debug_state(0, 2)
    b.close()
    for name, cuts in (("c4_2", (4, 6)), ("c1_5", (1, 6)), ("c5_1_1", (5, 6, 7)), ("w1", (1,)), ("w7", (7,))):
        b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr)
        prev = 0
        for h in cuts:
            b.process(x[:, :, 1152*prev:1152*h].contiguous(), 1440*(h - prev)); prev = h
        b.synchronize()
        out[name] = b.debug_state(0, 2)
        b.close()
    np.savez(sys.argv[1], **out)
else:
    env = dict(os.environ, SMST_DEBUG_MODE=os.environ.get("DM", "0"))
    subprocess.run([sys.executable, __file__, "/tmp/st_on.npz"], env=env, check=True)
    subprocess.run([sys.executable, __file__, "/tmp/st_off.npz"], env=dict(env, SMST_NO_STAGE="1"), check=True)
    a, b = np.load("/tmp/st_on.npz"), np.load("/tmp/st_off.npz")
    for k in a.files:
        d = np.abs(a[k] - b[k])
        print(k, "staged vs gathered: max %.3e, differing bins %d, first %s" % (float(d.max()), int((d > 0).sum()), np.argwhere(d > 0)[:6].tolist()))
    for k in ("c4_2", "c1_5"):
        print(k, "vs whole: staged %.3e gathered %.3e" % (float(np.abs(a[k] - a["whole"]).max()), float(np.abs(b[k] - b["whole"]).max())))
    print("c5_1_1 vs w7: staged %.3e gathered %.3e" % (float(np.abs(a["c5_1_1"] - a["w7"]).max()), float(np.abs(b["c5_1_1"] - b["w7"]).max())))
    d = np.abs(a["whole"] - a["chunked"]); print("staged whole vs chunked: max %.3e differing %d first %s" % (float(d.max()), int((d > 0).sum()), np.argwhere(d > 0)[:6].tolist()))
    d = np.abs(b["whole"] - b["chunked"]); print("gathered whole vs chunked: max %.3e differing %d" % (float(d.max()), int((d > 0).sum())))
