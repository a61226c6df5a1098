"""Shared scenario runner: replays a golden fixture's op list on any object with the reference's method names
(oracle RefStretch, or the product's SignalsmithStretch over the C ABI)."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# tolerances on rel-RMS vs the WASM golden output (the path is chaotic, so the bound is
# horizon-aware; every fixture is <= 16 hops = "short horizon", criterion (ii): 1e-3; identity: 1e-6)
GOLDEN_TOL = {
    "identity_mono_44k": 1e-6,
    "stretch_1p5_stereo": 1e-3,
    "stretch_0p75_stereo": 1e-3,
    "pitch_p12_stereo": 1e-3,
    "cheaper_96k_3ch": 1e-3,
    "seek_chunks_stereo": 1e-5,
    "flush_short_stereo": 1e-5,
    "stretch_1p5_noise": 1e-3,
    "cheaper_48k_stereo": 1e-3,
    "default_96k_stereo": 1e-3,
    "eight_channels_1p5": 1e-3,
}


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return z["x"], z["y"], json.loads(str(z["ops"])), json.loads(str(z["cfg"])), json.loads(str(z["info"]))


def configure(obj, channels, cfg):
    preset = cfg.get("preset", "default")
    sr = cfg.get("sample_rate", 48000.0)
    if preset == "default":
        obj.presetDefault(channels, sr)
    elif preset == "cheaper":
        obj.presetCheaper(channels, sr)  # the WASM ABI always uses the preset's default split (main.cpp:46-48)
    else:
        obj.configure(channels, cfg["block"], cfg["interval"], cfg.get("split", False))


def replay(obj, x, ops):
    outs = []
    for op in ops:
        kind = op["op"]
        if kind == "process":
            outs.append(obj.process(x[:, op["inStart"]:op["inStart"] + op["inLen"]], op["outLen"]))
        elif kind == "flush":
            outs.append(obj.flush(op["outLen"]))
        elif kind == "seek":
            obj.seek(x[:, op["inStart"]:op["inStart"] + op["inLen"]], op["rate"])
        elif kind == "reset":
            obj.reset()
        else:
            getattr(obj, kind)(*op["args"])
    return np.concatenate(outs, axis=1)
