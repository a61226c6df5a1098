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
    # 9.-11. (round 3) the other preset geometries and the widest stream the product takes:
    # presetCheaper at 48 kHz (2560 bins: register-blocked FFT 16 x 16 x 10), split computation as the WASM ABI always uses it
    case("cheaper_48k_stereo", x, [dict(op="process", inStart=0, inLen=n, outLen=int(n*1.25))], preset="cheaper")
    # presetDefault at 96 kHz (6144 bins: 16 x 16 x 24), -3 st
    x96 = synth_input(0, 2, 28800, 96000) + 0.5*synth_input(1, 2, 28800, 96000)
    case("default_96k_stereo", x96, [dict(op="setTransposeSemitones", args=[-3, 0]),
                                     dict(op="process", inStart=0, inLen=28800, outLen=28800)], sample_rate=96000)
    # (round 6) the presets at 192 kHz: 12288 / 10240 bins, beyond what two FFT buffers fit in a CU's LDS (the product's second buffer lives in
    # memory there).  Mono and short: eight hops of presetDefault (interval 5760), six of presetCheaper (interval 7680, split computation)
    x192 = synth_input(0, 1, 46080, 192000) + 0.5*synth_input(1, 1, 46080, 192000)
    case("default_192k_mono", x192, [dict(op="process", inStart=0, inLen=46080, outLen=int(46080*1.25))], sample_rate=192000)
    case("cheaper_192k_mono", x192, [dict(op="setTransposeSemitones", args=[4, 0]),
                                     dict(op="process", inStart=0, inLen=46080, outLen=46080)], preset="cheaper", sample_rate=192000)
    # 8 channels (a sine, a chirp and a noise stream among them), 48 kHz presetDefault, 1.5x: the channel lock over 8 channels
    x8 = np.concatenate([synth_input(0, 3, n, sr), synth_input(1, 3, n, sr)*0.7, synth_input(2, 2, n, sr)], axis=0)
    case("eight_channels_1p5", x8, [dict(op="process", inStart=0, inLen=n, outLen=int(n*1.5))])


def formant_revision_fixtures():
    """tests/golden/wasm_revision/: the WASM on three formant settings, plus probes of its linear memory after the last hop
    (SURVEY.md App. B technique): the Band array (found by the content of Band.input) and formantMetric (found by
    log-domain template matching).  The in-tree header does NOT reproduce these outputs -- the WASM was built from another
    revision of updateFormants; tests/test_oracle_golden.py::test_formant_revision_* measure and explain the difference."""
    import ref_oracle
    out_dir = os.path.join(HERE, "wasm_revision")
    os.makedirs(out_dir, exist_ok=True)
    sr, n, M = 48000, 14400, 3072
    x = synth_input(0, 2, n, sr) + 0.5*synth_input(1, 2, n, sr)
    settings = {
        "formant_comp": [dict(op="setTransposeSemitones", args=[4, 8000/48000]), dict(op="setFormantFactor", args=[1, 1]),
                         dict(op="setFormantBase", args=[200/48000])],
        "formant_shift": [dict(op="setFormantSemitones", args=[3, 0]), dict(op="setFormantBase", args=[200/48000])],
        "formant_shift_auto": [dict(op="setFormantSemitones", args=[3, 0])],
    }
    for name, pre in settings.items():
        ops = pre + [dict(op="process", inStart=0, inLen=n, outLen=n)]
        y, info = wasm_oracle.run(x, ops, dump_memory=True)
        mem = np.nan_to_num(info.pop("memory"), nan=0.0, posinf=0.0, neginf=0.0).astype(np.float64)
        mem[np.abs(mem) > 1e12] = 0
        # the native build (in-tree header) gives the content keys: its Band.input equals the WASM's (same L1 arithmetic)
        r = ref_oracle.RefStretch()
        r.presetDefault(2, float(sr))
        scenarios_replay(r, x, ops)
        rin = r.bands_complex(0)
        key = rin[0, 1000:1004]
        hits = [p for p in np.nonzero(np.abs(mem - key[0].real) <= 1e-3*abs(key[0].real))[0]
                if all(abs(mem[p + 7*j] - key[j].real) <= 1e-3*abs(key[j].real) + 1e-6 and
                       abs(mem[p + 7*j + 1] - key[j].imag) <= 1e-3*abs(key[j].imag) + 1e-6 for j in range(4))]
        assert 1 <= len(hits) <= 2 and hits[-1] - hits[0] in (0, 2), hits  # (Band.prevInput == Band.input after a hop: +2 matches too)
        bands = mem[hits[0] - 7*1000:hits[0] - 7*1000 + 7*M*2].reshape(2, M, 7).astype(np.float32)
        # formantMetric: bands + 2 floats; template = log of the amplitude spectrum (any smooth function of it correlates)
        lm = np.where(mem > 1e-20, np.log(np.maximum(mem, 1e-20)), -46.0)
        t = 0.5*np.log(np.maximum((np.abs(rin)**2).sum(axis=0), 1e-20))
        t0 = t - t.mean()
        tails = np.nonzero((mem[M:-1] == 0) & (mem[M + 1:] == 0) & (mem[M - 1:-2] > 0))[0]
        cand = sorted(((float(np.dot(lm[p:p + M] - lm[p:p + M].mean(), t0)/(np.linalg.norm(lm[p:p + M] - lm[p:p + M].mean())*np.linalg.norm(t0) + 1e-30)), int(p))
                       for p in tails if np.all(mem[p:p + M] > 0)), reverse=True)
        assert cand and cand[0][0] > 0.8, cand[:3]
        cand = [cand[0][1]]
        metric = mem[cand[0]:cand[0] + M + 2].astype(np.float32)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), x=x.astype(np.float32), y=y.astype(np.float32),
                            ops=json.dumps(ops), cfg=json.dumps({}), info=json.dumps(info), wasm_bands=bands, wasm_formant_metric=metric)
        print(name, "->", y.shape, "Band array @", hits[0] - 7000, "formantMetric @", cand[0])


def seek_rate_fixtures():
    """tests/golden/wasm_revision_seek/: seek() with playbackRate != 1 followed by process() on the WASM.  The in-tree header
    differs from it by 2e-4 (rate 0.8) .. 2e-3 (rate 1.25) from the first hop on (rate 1: 6e-7) --
    tests/test_oracle_golden.py::test_seek_rate_revision_measured records that."""
    out_dir = os.path.join(HERE, "wasm_revision_seek")
    os.makedirs(out_dir, exist_ok=True)
    sr, n = 48000, 14400
    x = synth_input(0, 2, n, sr) + 0.5*synth_input(1, 2, n, sr)
    for rate in (1.0, 0.8, 1.25):
        ops = [dict(op="seek", inStart=0, inLen=7200, rate=rate), dict(op="process", inStart=7200, inLen=int(5760*rate), outLen=5760)]
        y, info = wasm_oracle.run(x, ops)
        name = "seek_rate_%s" % str(rate).replace(".", "p")
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), x=x.astype(np.float32), y=y.astype(np.float32),
                            ops=json.dumps(ops), cfg=json.dumps({}), info=json.dumps(info))
        print(name, "->", y.shape)


def scenarios_replay(obj, x, ops):
    import scenarios
    return scenarios.replay(obj, x, ops)


def only(names):
    """(re)generate just the named fixtures: python tests/golden/make_golden.py only <name> ..."""
    global case
    real = case

    def filtered(name, *a, **k):
        if name in names:
            real(name, *a, **k)
    case = filtered
    main()
    case = real


def split_event_scenarios(interval, n_in, pre_intervals, post_intervals, offsets, rate=1.0, pre_mapped=None, events=None):
    """Op lists of the split-computation fixtures: process() up to `off` samples into an interval, an event, more process().
    Events: flush of one interval / of a third of it, setTransposeSemitones 3 -> 7 (the block in flight is already mapped: findPeaks
    reads the live parameters, signalsmith-stretch.h:850-856), 0 -> 4 (`mappedFrequencies` was latched at the block's start, :300),
    reset(), seek()."""
    out = []
    for event in (events or ("flush", "flush_short", "param", "param_unmapped", "reset", "seek")):
        for off in offsets:
            nout = (pre_mapped if (event == "param" and pre_mapped) else pre_intervals)*interval + off  # (a transposed run of the small geometry drifts from the WASM by 2e-3 within ten hops: keep it short)
            nin = int(round(nout*rate))
            ops = []
            if event == "param":  # only this event needs a mapped block in flight; the others stay un-transposed, where the fixtures resolve 1e-6
                ops.append(dict(op="setTransposeSemitones", args=[3, 0]))
            ops.append(dict(op="process", inStart=0, inLen=nin, outLen=nout))
            pos = nin
            if event == "flush":
                ops.append(dict(op="flush", outLen=interval))
            elif event == "flush_short":
                ops.append(dict(op="flush", outLen=max(1, interval//3)))
            elif event == "param":
                ops.append(dict(op="setTransposeSemitones", args=[7, 0]))
            elif event == "param_unmapped":
                ops.append(dict(op="setTransposeSemitones", args=[4, 0]))
            elif event == "reset":
                ops.append(dict(op="reset"))
            elif event == "seek":
                ops.append(dict(op="seek", inStart=pos, inLen=interval*3, rate=1.0))
                pos += interval*3
            npost = post_intervals*interval
            ops.append(dict(op="process", inStart=pos, inLen=int(round(npost*rate)), outLen=npost))
            assert pos + int(round(npost*rate)) <= n_in
            out.append(("%s_%d" % (event, off), ops))
    return out


def split_event_fixtures():
    """tests/golden/split_events/: what the reference's shipped WASM build does when something happens BETWEEN two interval
    boundaries in split-computation mode (signalsmith-stretch.h:292-296,321-325,407-415: the block's steps are spread over the
    interval, the output is read from the stashed ring).  The step partition of analyseSteps() / synthesiseSteps() lives in
    signalsmith-linear, which only the WASM contains: these fixtures pin it.  One file per geometry: the input once, one output
    per scenario."""
    out_dir = os.path.join(HERE, "split_events")
    os.makedirs(out_dir, exist_ok=True)
    sr = 48000
    geoms = {
        # the small geometry of tests/parity_cases.py (SMALL_SPLIT): 24 steps per stereo block, one step per ~5 samples
        "small_stereo": dict(cfg=dict(preset="configure", block=512, interval=128, split=True), channels=2, interval=128, n=6000,
                             pre=10, pre_mapped=3, post=4, offsets=(1, 5, 8, 12, 16, 32, 46, 47, 54, 55, 64, 80, 100, 110, 115, 116, 121, 122, 124, 127)),  # 46|47: findPeaks of a mapped stereo block (step 8 of 24); 54|55: first main-prediction chunk of an unmapped one (step 8 of 20); 115|116, 121|122: its two synthesis steps
        # a playback rate of 1.25: every block re-analyses its previous spectrum (:303), channels + 1 more steps in front (26 stereo, 30 mapped)
        "small_stereo_rate": dict(cfg=dict(preset="configure", block=512, interval=128, split=True), channels=2, interval=128, n=6000,
                                  pre=10, pre_mapped=3, post=4, rate=1.25, events=("flush", "param"),
                                  offsets=(3, 9, 14, 20, 40, 49, 50, 56, 57, 90, 104, 105, 109, 110, 120, 127)),  # 49|50: findPeaks (step 11 of 30); 56|57: first main-prediction chunk (step 11 of 26); 104|105, 109|110: the synthesis steps
        # presetCheaper at 48 kHz (always split in the WASM ABI, web/emscripten/main.cpp:46-48), mono
        "cheaper_48k_mono": dict(cfg=dict(preset="cheaper"), channels=1, interval=1920, n=26000,
                                 pre=4, post=3, offsets=(5, 700, 1500, 1850, 1919)),
    }
    for name, g in geoms.items():
        C = g["channels"]
        x = (synth_input(0, C, g["n"], sr) + 0.3*synth_input(3, C, g["n"], sr)).astype(np.float32)  # tonal: the fixtures must resolve single steps, not chaos
        scen = split_event_scenarios(g["interval"], g["n"], g["pre"], g["post"], g["offsets"], pre_mapped=g.get("pre_mapped"), rate=g.get("rate", 1.0), events=g.get("events"))
        blob = dict(x=x, cfg=json.dumps(g["cfg"]), names=json.dumps([s[0] for s in scen]))
        for key, ops in scen:
            y, info = wasm_oracle.run(x, ops, **g["cfg"])
            blob["ops_" + key] = json.dumps(ops)
            blob["y_" + key] = y.astype(np.float32)
        blob["info"] = json.dumps(info)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **blob)
        print(name, len(scen), "scenarios", info)

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "only":
        only(set(sys.argv[2:]))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "formants":
        formant_revision_fixtures()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "seek":
        seek_rate_fixtures()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "split":
        split_event_fixtures()
        sys.exit(0)
    main()
