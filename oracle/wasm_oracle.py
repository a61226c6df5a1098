"""TEST INFRASTRUCTURE ONLY (oracle).

Python front-end for `oracle/wasm/run_wasm.js`: runs the reference's own shipped WebAssembly build
(/root/reference/web/emscripten/main.js:9 -- the real header + the real signalsmith-linear) under Node.
Only usable in the build container (needs /root/reference and `node`); it generates the committed
fixtures under tests/golden/ (see tests/golden/make_golden.py).  Nothing on the product path imports it.
"""
import json
import os
import subprocess
import tempfile

import numpy as np

REFERENCE_JS = "/root/reference/web/emscripten/main.js"
_HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    from shutil import which
    return os.path.exists(REFERENCE_JS) and which("node") is not None


def run(x, ops, channels=None, sample_rate=48000.0, preset="default", block=0, interval=0, split=False, dump_memory=False):
    """x: float32 [C, inTotal]; ops: list of dicts (see run_wasm.js).  Returns (out[C, totalOut], info); with dump_memory
    info["memory"] is the instance's linear memory after the last op as a float32 array (state probes)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    C = x.shape[0] if channels is None else channels
    with tempfile.TemporaryDirectory() as d:
        fin, fout, fjob = (os.path.join(d, n) for n in ("in.f32", "out.f32", "job.json"))
        x.tofile(fin)
        job = dict(channels=C, sampleRate=float(sample_rate), preset=preset, block=int(block), interval=int(interval),
                   split=bool(split), input=fin, inTotal=int(x.shape[1]), output=fout, ops=ops)
        if dump_memory:
            job["dumpMemory"] = os.path.join(d, "mem.bin")
        with open(fjob, "w") as f:
            json.dump(job, f)
        res = subprocess.run(["node", os.path.join(_HERE, "wasm", "run_wasm.js"), REFERENCE_JS, fjob],
                             check=True, capture_output=True, text=True)
        info = json.loads(res.stdout.strip().splitlines()[-1])
        out = np.fromfile(fout, dtype=np.float32).reshape(C, -1)
        if dump_memory:
            info["memory"] = np.fromfile(job["dumpMemory"], dtype=np.float32)
    return out, info
