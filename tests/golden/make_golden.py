"""Generates tests/golden/*.npz from the reference's own shipped WebAssembly build (the only runnable form of the
real reference).  Run in the build container only:  python tests/golden/make_golden.py
Each fixture holds the input, the op list, the WASM output and the WASM-reported geometry."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import wasm_oracle  # noqa: E402
from conftest import synth_input  # noqa: E402


def case(name, x, ops, **cfg):
    out, info = wasm_oracle.run(x, ops, **cfg)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), x=x.astype(np.float32), y=out.astype(np.float32),
                        ops=json.dumps(ops), cfg=json.dumps(cfg), info=json.dumps(info))
    print(name, x.shape, "->", out.shape, info)


def main():
    sr = 48000
    n = 14400
    # 1. config-1 core: mono 44.1 kHz, 1.0x / 0 st  (identity with delay)
    x = synth_input(0, 1, 13230, 44100)
    case("identity_mono_44k", x, [dict(op="process", inStart=0, inLen=13230, outLen=13230)], sample_rate=44100)
    # 2. config-2 per-stream: stereo 48 kHz, 1.5x, tonal
    x = synth_input(0, 2, n, sr) + 0.5*synth_input(1, 2, n, sr)
    case("stretch_1p5_stereo", x, [dict(op="process", inStart=0, inLen=n, outLen=int(n*1.5))])
    # 3. 0.75x (config-4 literal per-stream)
    case("stretch_0p75_stereo", x, [dict(op="process", inStart=0, inLen=n, outLen=int(n*0.75))])
    # 4. config-3 per-stream: +12 st with 8 kHz tonality limit
    case("pitch_p12_stereo", x, [dict(op="setTransposeSemitones", args=[12, 8000/48000]),
                                 dict(op="process", inStart=0, inLen=n, outLen=n)])
    # 5. config-5 flavour: 3 channels, 96 kHz, presetCheaper (split), 1.2x, +5 st
    x5 = synth_input(3, 3, 19200, 96000)
    case("cheaper_96k_3ch", x5, [dict(op="setTransposeSemitones", args=[5, 0]),
                                 dict(op="process", inStart=0, inLen=19200, outLen=23040)], preset="cheaper", sample_rate=96000)
    # 6. seek + real-time sized chunks (web-wrapper calling pattern, web/web-wrapper.js:313-315), rate 1
    ops = [dict(op="seek", inStart=0, inLen=7200, rate=1.0)]
    for k in range(56):
        ops.append(dict(op="process", inStart=7200 + 128*k, inLen=128, outLen=128))
    case("seek_chunks_stereo", x, ops)
    # 7. process then a short flush (<= one interval), then more processing
    ops = [dict(op="process", inStart=0, inLen=9000, outLen=9000), dict(op="flush", outLen=1000),
           dict(op="process", inStart=9000, inLen=5000, outLen=5000)]
    case("flush_short_stereo", x, ops)
    # 8. noise, 1.5x (short horizon only is comparable)
    xn = synth_input(2, 2, n, sr)
    case("stretch_1p5_noise", xn, [dict(op="process", inStart=0, inLen=n, outLen=int(n*1.5))])


if __name__ == "__main__":
    main()
