import sys, os, importlib, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from conftest import package, rel_rms, synth_input
if len(sys.argv) > 1:
    os.environ["SMST_FFT_TABLES"] = sys.argv[1]
    pkg = package()
    C, sr, S, n = 2, 48000, 6, 30000
    xs = np.stack([synth_input(s, C, n, sr) for s in range(S)])
    rng = np.random.Generator(np.random.PCG64(5))
    stretch = rng.uniform(0.75, 1.5, S); semis = rng.uniform(-12, 12, S)
    nin = [n - 317*s for s in range(S)]
    nout = [int(round(nin[s]*stretch[s])) for s in range(S)]
    b = pkg.StretchBatch(S, C, block=5760, interval=1440, split=False)
    for s in range(S): b.setTransposeSemitones(float(semis[s]), 0.0, stream=s)
    y = b.process(xs, nout, in_samples=nin)
    st = [b.debug_state(s, 0) for s in range(S)]
    np.savez("/tmp/lvf_%s.npz" % sys.argv[1], y=y, nout=np.array(nout), st=np.array(st))
else:
    for m in ("full", "lean"):
        subprocess.run([sys.executable, __file__, m], check=True)
    a, b = np.load("/tmp/lvf_full.npz"), np.load("/tmp/lvf_lean.npz")
    for s in range(6):
        no = int(a["nout"][s]); I = 1440
        print(s, "state", "%.1e" % rel_rms(np.abs(b["st"][s]), np.abs(a["st"][s])), ["%.1e" % rel_rms(b["y"][s][:, h*I:(h+1)*I], a["y"][s][:, h*I:(h+1)*I]) for h in range(min(8, no//I))])
