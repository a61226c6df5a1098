import importlib, os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import synth_input
import torch
pkg = importlib.import_module("signalsmith-stretch_amd")
C, sr, S = 2, 48000, 2
x = torch.from_numpy(np.stack([synth_input(s, C, 28800, sr) for s in range(S)])).cuda()
b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr)
b.process(x[:, :, :5760].contiguous(), 7200); b.synchronize()
inp = b.debug_state(0, 0)
en = b.debug_state(0, 3)
re, im = inp.real.astype(np.float32), inp.imag.astype(np.float32)
ref = ((re*re).astype(np.float32) + (im*im).astype(np.float32)).astype(np.float32)
print("energy vs cnorm(input): differing", int((ref != en).sum()), "of", en.size, "max rel", float(np.abs(ref - en).max()/np.abs(en).max()))
