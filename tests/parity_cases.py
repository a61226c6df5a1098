"""Parity cases shared by the CPU-emulated run (host logic, `-m "not gpu"`) and the real gfx950 run (`-m gpu`).
Every case drives the product through its C ABI (python ctypes mirror of the reference API) and the checker
(oracle/_ref: the unmodified reference header) with the same seeded inputs, and compares.

TOLERANCE (stated here once).  The whole path is fp32 and the phase recurrence is chaotic (a 1e-7
relative perturbation of the INPUT changes the reference's own output by 1e-5..1e-2 within 16 hops, more for noise
and for pitch-mapped material where peak decisions flip).  Three kinds of comparison, as App. D.2 prescribes:

 (a) sample domain, free running (`assert_parity`): over every horizon h in HORIZONS (hops),
     rel-RMS(product, checker) <= max(FLOOR, SELF_FACTOR * S_h),  S_h = max over 3 seeds of
     rel-RMS(checker(x*(1 + PERTURBATION*u)), checker(x)) over the same horizon -- and the bound is CAPPED at `cap`
     (CAP_TONAL 5e-3; CAP_FORMANT 5e-2 with formant processing, D.2 iii): it never exceeds the cap, and once
     SELF_FACTOR * (MEDIAN over the seeds) exceeds the cap the horizon is chaos-dominated (most perturbed runs of the
     checker itself have diverged), a sample-domain bound would assert nothing, and the comparison stops there (at
     least the first horizon must have been checked).  The median, not the maximum, decides that: a single perturbed
     run in which a discrete decision flipped (peak list, arg-max channel) does not switch the comparison off.  PERTURBATION = 1e-6 is NOT a free choice: it is the input-equivalent size of the
     arithmetic difference between the two implementations, measured by `case_teacher_forced` (the product's analysis
     spectra differ from the checker's by e rel-RMS; an input perturbation p moves them by p/sqrt(3)) and asserted there
     (test_teacher_forced_*: sqrt(3)*e <= PERTURBATION).
 (b) teacher-forced single hops (`case_teacher_forced`, D.2 i): product state := checker state, ONE hop, spectra and
     emitted samples <= 1e-5 -- no chaos amplification, a plain fp32 bound.
 (c) phase-free quantities, free running over long horizons (`case_hop_magnitudes`, D.2 iv/v): per-hop |output_c[b]|
     <= 1e-4 rel-RMS, arg-max-channel and output-map agreement rates, output RMS within 1 %.
Cases without phase-vocoder feedback (1.0x identity, ring bookkeeping) use TOL_EXACT = 2e-6."""
import numpy as np

from conftest import package, rel_rms, synth_input
import scenarios

TOL_EXACT = 2e-6
FLOOR = 1e-4
SELF_FACTOR = 5.0
PERTURBATION = 1e-6  # input-equivalent size of the two implementations' arithmetic difference (measured: see above)
HORIZONS = (3, 6, 12, 24, 48, 96, 192, 1 << 30)
SELF_SEEDS = (1, 2, 3)
CAP_TONAL = 5e-3    # D.2 (iii): stretch-only / pitch-only
CAP_FORMANT = 5e-2  # D.2 (iii): with formant processing


def make(kind, lib, ref, channels, cfg, setup=None, seed=0):
    obj = package().SignalsmithStretch(seed=seed, lib=lib) if kind == "product" else ref.RefStretch(seed)
    scenarios.configure(obj, channels, cfg)
    if setup:
        setup(obj)
    return obj


def perturbed(x, seed=1):
    u = np.random.default_rng(seed).uniform(-1, 1, x.shape)
    return (x*(1 + PERTURBATION*u)).astype(np.float32)


def assert_parity(y, o, o_self, interval, label, cap=CAP_TONAL, require_informative=True):
    """o_self: one array or a list of arrays = checker outputs for differently perturbed inputs (max is used).
    Returns the number of hops up to which the sample-domain comparison was informative (and was asserted)."""
    assert y.shape == o.shape, (label, y.shape, o.shape)
    selfs = o_self if isinstance(o_self, (list, tuple)) else [o_self]
    total = o.shape[1]
    checked = 0
    for h in HORIZONS:
        n = min(total, h*interval)
        if n <= 0:
            continue
        if np.mean(np.square(o[:, :n], dtype=np.float64))*n < 1e-6*np.mean(np.square(o, dtype=np.float64))*total:
            continue  # the checker's own output is still (numerically) silent over this horizon: nothing to compare
        owns = sorted(rel_rms(v[:, :n], o[:, :n]) for v in selfs)
        err, own, typical = rel_rms(y[:, :n], o[:, :n]), owns[-1], owns[len(owns)//2]
        if SELF_FACTOR*typical > cap:  # chaos-dominated from here on: the phase-free checks take over (case_hop_magnitudes)
            break
        tol = min(cap, max(FLOOR, SELF_FACTOR*own))
        assert err <= tol, "%s: horizon %d hops: rel-RMS %.3e > %.3e (checker self-sensitivity %.3e)" % (label, min(h, total//interval), err, tol, own)
        checked = min(h, -(-total//interval))
        if n == total:
            break
    # (noise through a frequency map is chaos-dominated from the first hop on: callers that pass require_informative=False
    # follow up with a phase-free check -- output level, per-hop magnitudes)
    assert checked > 0 or not require_informative, "%s: not even the first horizon is informative (self-sensitivity above the cap)" % label
    return checked


def check_scenario(lib, ref, cfg, x, play, label, setup=None, cap=CAP_TONAL):
    """play(obj, x) -> concatenated output; run on the product, the checker, and the checker with perturbed input."""
    C = x.shape[0]
    g, r = make("product", lib, ref, C, cfg, setup), make("ref", lib, ref, C, cfg, setup)
    y, o = play(g, x), play(r, x)
    o2 = [play(make("ref", lib, ref, C, cfg, setup), perturbed(x, seed)) for seed in SELF_SEEDS]
    assert_parity(y, o, o2, r.intervalSamples(), label, cap=cap)
    return y, o


def case_golden(lib, ref, name):
    """Product vs. the WASM golden vector AND vs. the checker (short fixtures, <= 16 hops)."""
    x, y, ops, cfg, info = scenarios.load_golden(name)
    g = make("product", lib, ref, x.shape[0], cfg)
    assert (g.blockSamples(), g.intervalSamples(), g.inputLatency(), g.outputLatency()) == \
        (info["block"], info["interval"], info["inputLatency"], info["outputLatency"])
    out, chk = check_scenario(lib, ref, cfg, x, lambda obj, xx: scenarios.replay(obj, xx, ops), name)
    assert out.shape == y.shape
    # vs the WASM itself: the checker's distance to the WASM plus the product's distance to the checker
    assert rel_rms(out, y) <= max(scenarios.GOLDEN_TOL[name], 2*rel_rms(chk, y) + FLOOR), (name, rel_rms(out, y))


SMALL = dict(preset="configure", block=512, interval=128, split=False)
SMALL_SPLIT = dict(preset="configure", block=512, interval=128, split=True)


def case_api_surface(lib, ref, cfg=SMALL, scale=1):
    """seek / process in ragged chunks / flush / process-after-flush / outputSeek / exact; `scale` stretches every
    sample count of the walk (1 for the small geometry it was written for, ~block/512 for larger blocks)."""
    C, sr = 2, 48000
    x = synth_input(0, C, 128*150*scale, sr) + 0.3*synth_input(1, C, 128*150*scale, sr)
    # many hops in one call (crosses the 64-hop tile boundary twice)
    check_scenario(lib, ref, cfg, x, lambda o, xx: o.process(xx, int(xx.shape[1]*1.25)), "one long call")

    def ragged(o, xx):  # ragged chunk sizes, ratio 1.3 (0..5 hops per call)
        rng = np.random.default_rng(0)
        pos, outs = 0, []
        while pos < xx.shape[1] - 700*scale:
            ni = int(rng.integers(1, 600*scale))
            outs.append(o.process(xx[:, pos:pos + ni], int(ni*1.3)))
            pos += ni
        return np.concatenate(outs, axis=1)
    check_scenario(lib, ref, cfg, x, ragged, "ragged chunks")

    def seek_flush(o, xx):  # seek, process, flush (short), process again, flush (exactly one interval)
        o.seek(xx[:, :640*scale], 0.8)
        k = scale
        return np.concatenate([o.process(xx[:, 640*k:4640*k], 5000*k), o.flush(100*k), o.process(xx[:, 5000*k:7000*k], 2000*k), o.flush(o.intervalSamples())], axis=1)
    check_scenario(lib, ref, cfg, x, seek_flush, "seek/process/flush")

    g, r = make("product", lib, ref, C, cfg), make("ref", lib, ref, C, cfg)
    n = r.outputSeekLength(0.8)
    assert g.outputSeekLength(0.8) == n and g.seekLength() == r.seekLength()

    def output_seek(o, xx):
        o.outputSeek(xx[:, :n])
        return o.process(xx[:, n:n + 4000*scale], 5000*scale)
    check_scenario(lib, ref, cfg, x, output_seek, "outputSeek")

    def exact(o, xx):
        out, ok = o.exact(xx[:, :6000*scale], 7000*scale)
        assert ok
        return out
    check_scenario(lib, ref, cfg, x, exact, "exact")
    (a, ok_a), (b, ok_b) = g.exact(x[:, :200], 300), r.exact(x[:, :200], 300)  # too short: false + zeroed output
    assert not ok_a and not ok_b and np.abs(a).max() == 0


def case_random_call_sequences(lib, ref, seeds=range(6), cfg=SMALL, calls=14):
    """API fuzz: a seeded random walk over the public members -- process() with ragged chunks at changing ratios (0.75 ... 2.6), input-only
    and output-only calls, setTransposeSemitones / setTransposeFactor / setFormantFactor / setFormantBase between calls, seek(), flush()
    of up to 2.3 intervals, reset() -- replayed on the product, the checker and the perturbed checkers (check_scenario).  Short walks (a
    few dozen hops), so the sample-domain comparison stays informative; 1-3 channels.  Stretches beyond 2x (output without input, long
    flushes, the first hop after a start or reset) draw random time factors: product and checker are constructed with the same seed
    and the product replicates the checker's std::default_random_engine (smst_kernels_common.h: engineDraw), so they stay comparable."""
    sr = 48000
    seeds = list(seeds)
    uninformative = []
    for seed in seeds:
        C = 1 + seed % 3
        formants = seed % 4 == 3
        x = synth_input(seed, C, 40000, sr) + 0.4*synth_input(seed + 7, C, 40000, sr)

        def play(o, xx, seed=seed, formants=formants):
            rng = np.random.default_rng(1000 + seed)
            pos, outs = 0, []
            ratio = 1.0
            for call in range(calls):
                kind = rng.choice(["process", "process", "process", "param", "flush", "seek", "reset", "empty"], p=[0.3, 0.2, 0.15, 0.15, 0.06, 0.05, 0.04, 0.05])
                if kind == "param":
                    which = int(rng.integers(0, 4 if formants else 2))
                    if which == 0:
                        o.setTransposeSemitones(float(rng.integers(-7, 8)), float(rng.choice([0.0, 0.15])))
                    elif which == 1:
                        o.setTransposeFactor(float(rng.choice([0.8, 1.0, 1.25])), 0.0)
                    elif which == 2:
                        o.setFormantFactor(float(rng.choice([0.9, 1.0, 1.15])), bool(rng.integers(0, 2)))
                    else:
                        o.setFormantBase(float(rng.choice([0.0, 150.0/sr])))
                    ratio = float(rng.choice([0.75, 1.0, 1.3, 1.6, 2.6]))
                elif kind == "flush":  # up to 2.3 intervals: beyond one interval the flush runs hops at a huge time factor (random engine, below)
                    outs.append(o.flush(int(rng.integers(1, 300))))
                elif kind == "seek":
                    n = int(rng.integers(50, 700))
                    o.seek(xx[:, pos:pos + n], float(rng.choice([0.8, 1.0, 1.2])))
                    pos += n
                elif kind == "reset":
                    o.reset()
                elif kind == "empty":  # output without new input (a stretch beyond 2x: random time factors) / input without output
                    if rng.integers(0, 2):
                        outs.append(o.process(xx[:, pos:pos], int(rng.integers(1, 200))))
                    else:
                        n = int(rng.integers(1, 200))
                        outs.append(o.process(xx[:, pos:pos + n], 0))
                        pos += n
                else:
                    n = int(rng.integers(1, 900))
                    outs.append(o.process(xx[:, pos:pos + n], max(1, int(n*ratio))))
                    pos += n
            outs.append(o.process(xx[:, pos:pos + 1500], 1800))  # every walk ends with sound coming out
            return np.concatenate([np.asarray(v) for v in outs], axis=1)
        try:
            check_scenario(lib, ref, cfg, x, play, "random walk %d (%d ch%s)" % (seed, C, ", formants" if formants else ""),
                           cap=CAP_FORMANT if formants else CAP_TONAL)
        except AssertionError as e:
            if "informative" not in str(e):
                raise
            uninformative.append(seed)  # the checker's own response to a 1e-6 perturbation exceeds the cap from the first horizon on: nothing to assert
    assert len(uninformative) <= max(1, len(seeds)//5), uninformative
    return dict(walks=len(seeds), uninformative=uninformative)


def case_realtime_quanta(lib, ref, cfg=SMALL, quantum=128, quanta=70):
    """The two calling patterns of the reference's AudioWorklet wrapper (web/web-wrapper.js:255-315), SURVEY.md 8(f)
    rank 3: (i) live input, process(quantum, quantum) per render quantum; (ii) buffered playback, where every quantum
    re-seeks with the last inputLatency+outputLatency input samples and asks for output without new input:
    seek(bufferLength, rate); process(0, quantum)."""
    C, sr = 2, 48000
    x = synth_input(0, C, quantum*(quanta + 40), sr) + 0.3*synth_input(2, C, quantum*(quanta + 40), sr)

    def live(o, xx):
        return np.concatenate([o.process(xx[:, q*quantum:(q + 1)*quantum], quantum) for q in range(quanta)], axis=1)
    check_scenario(lib, ref, cfg, x, live, "live quanta")

    for rate in (1.0, 0.8):
        def playback(o, xx, rate=rate):
            buf_len = o.inputLatency() + o.outputLatency()
            outs = []
            for q in range(quanta):
                end = int(round((q + 1)*quantum*rate)) + o.inputLatency()
                buf = np.zeros((C, buf_len), np.float32)
                lo = max(0, end - buf_len)
                buf[:, buf_len - (end - lo):] = xx[:, lo:end]
                o.seek(buf, rate)
                outs.append(o.process(xx[:, :0], quantum))
            return np.concatenate(outs, axis=1)
        check_scenario(lib, ref, cfg, x, playback, "playback quanta rate %.1f" % rate)


def case_split_mode(lib, ref):
    C, sr = 2, 48000
    x = synth_input(0, C, 12000, sr)
    g, r = make("product", lib, ref, C, SMALL_SPLIT), make("ref", lib, ref, C, SMALL_SPLIT)
    assert g.outputLatency() == r.outputLatency() == 256 + 128
    # interval-aligned flush in split mode (mid-interval flushes differ by design, see DESIGN.md "deviations")
    check_scenario(lib, ref, SMALL_SPLIT, x, lambda o, xx: np.concatenate([o.process(xx[:, :6000], 7040), o.flush(90)], axis=1), "split")


def case_pitch_and_formants(lib, ref, cfg=SMALL, n=9000):
    C, sr = 2, 48000
    x = synth_input(0, C, n, sr) + 0.5*synth_input(4, C, n, sr)
    settings = [
        ("pitch+12/tonality", lambda o: o.setTransposeSemitones(12, 8000/48000), 1.0),
        ("pitch-7", lambda o: o.setTransposeSemitones(-7, 0), 0.9),
        ("formant-comp", lambda o: (o.setTransposeSemitones(4, 8000/48000), o.setFormantFactor(1, True), o.setFormantBase(200/48000)), 0.75),
        ("formant-shift-auto-base", lambda o: (o.setFormantSemitones(3, False), o.setFormantBase(0)), 1.2),
        ("freq-map-table", lambda o: o.setFreqMapTable(np.array([(i + 0.5)/128*1.5 for i in range(64)], np.float32)), 1.0),
    ]
    for label, setup, stretch in settings:
        check_scenario(lib, ref, cfg, x, lambda o, xx, stretch=stretch: o.process(xx, int(n*stretch)), label, setup=setup,
                       cap=CAP_FORMANT if label.startswith("formant") else CAP_TONAL)


def case_large_plan_mapped(lib, ref):
    """A plan beyond the presets (8193 bins: 33 bins per thread in the feed kernels, above their register-resident limit
    of 24): pitch map + formant envelope go through the LDS form of the scan passes and the bisection form of the peak
    cover, which no preset reaches."""
    cfg = dict(preset="configure", block=15360, interval=3840, split=False)
    C, sr, n = 2, 48000, 5*3840 + 200
    x = synth_input(1, C, n, sr) + 0.5*synth_input(3, C, n, sr)
    for label, setup in (("large plan pitch+5", lambda o: o.setTransposeSemitones(5, 8000/48000)),
                         ("large plan formant-comp", lambda o: (o.setTransposeSemitones(-3, 0), o.setFormantFactor(1, True), o.setFormantBase(200/48000)))):
        check_scenario(lib, ref, cfg, x, lambda o, xx: o.process(xx, n), label, setup=setup,
                       cap=CAP_FORMANT if "formant" in label else CAP_TONAL)


def case_silence(lib, ref):
    sr = 48000
    x = synth_input(0, 1, 4000, sr)
    z = np.zeros((1, 700), np.float32)

    def play(o, xx):
        outs = []
        for chunk in (xx[:, :2000], z, z, z, z, xx[:, 2000:3000], z, xx[:, 3000:]):
            outs.append(o.process(chunk, chunk.shape[1] + 50))
        return np.concatenate(outs, axis=1)
    y, o = check_scenario(lib, ref, SMALL, x, play, "silence")
    assert np.array_equal(y[:, 3550:4300] == 0, o[:, 3550:4300] == 0)  # the pass-through region is exactly zero in both


def case_channels(lib, ref, channel_counts=(1, 3, 8)):
    sr = 48000
    for C in channel_counts:
        x = synth_input(3, C, 6000, sr)
        x *= (1 + 0.3*np.arange(C))[:, None].astype(np.float32)  # different energies: exercises the max-channel hand-over
        check_scenario(lib, ref, SMALL, x, lambda o, xx: o.process(xx, 7000), "%d channels" % C,
                       setup=lambda o: o.setTransposeSemitones(2, 0.2))


def case_fused_equals_unfused(lib, monkeypatch, channel_counts=(3, 8), geometry=None, n=9000):
    """The fused recurrence (kVocoder / kVocoderN: records in LDS) against the un-fused one (kPredictB + kChain: records
    through HBM, SMST_NO_FUSE=1): the same arithmetic in the same order, so bit-identical -- over two calls (carried state)
    and with a frequency map on half of the streams."""
    pkg = package()
    geometry = geometry or dict(block=512, interval=128, split=False)
    for C in channel_counts:
        xs = np.stack([synth_input(s, C, n, 48000)*(1 + 0.2*np.arange(C))[:, None].astype(np.float32) for s in range(4)])
        outs = []
        for unfused in (False, True):
            if unfused:
                monkeypatch.setenv("SMST_NO_FUSE", "1")
            else:
                monkeypatch.delenv("SMST_NO_FUSE", raising=False)
            b = pkg.StretchBatch(4, C, lib=lib, **geometry)
            b.setTransposeSemitones(3.0, 0.2, stream=1)
            b.setTransposeSemitones(-4.0, 0.0, stream=3)
            y1 = np.array(b.process(xs[:, :, :n//3], int(n//3*1.3)), copy=True)
            y2 = np.array(b.process(xs[:, :, n//3:], int((n - n//3)*1.3)), copy=True)
            b.close()
            outs.append(np.concatenate([y1, y2], axis=2))
        monkeypatch.delenv("SMST_NO_FUSE", raising=False)
        assert np.abs(outs[0]).max() > 0.05
        assert np.array_equal(outs[0], outs[1]), (C, float(np.abs(outs[0] - outs[1]).max()))


def case_gather_pass_shapes(lib, monkeypatch, channel_counts=(1, 2), n=12000, extra_geometries=()):
    """kVocoder's gathering producers (tiles with a frequency map): passes of 4 rows x 16 steps (the default, round 6) against 8 x 8
    (SMST_VOC_WIDE=0) -- the same records into the same slots: bit-identical, over two calls, mapped and unmapped streams side by side."""
    pkg = package()
    for geometry in (dict(block=512, interval=128, split=False), dict(block=1024, interval=192, split=False)) + tuple(extra_geometries):
        for C in channel_counts:
            xs = np.stack([synth_input(s, C, n, 48000)*(1 + 0.2*np.arange(C))[:, None].astype(np.float32) for s in range(4)])
            outs = []
            for wide in ("1", "0"):
                monkeypatch.setenv("SMST_VOC_WIDE", wide)
                before = pkg.launch_count("vocoder_gather", lib)
                b = pkg.StretchBatch(4, C, lib=lib, **geometry)
                b.setTransposeSemitones(3.0, 0.2, stream=1)
                b.setTransposeSemitones(-4.0, 0.0, stream=3)
                y1 = np.array(b.process(xs[:, :, :n//3], int(n//3*1.3)), copy=True)
                y2 = np.array(b.process(xs[:, :, n//3:], int((n - n//3)*1.3)), copy=True)
                b.close()
                assert pkg.launch_count("vocoder_gather", lib) > before, (geometry, C, "the gathering form did not run")
                outs.append(np.concatenate([y1, y2], axis=2))
            monkeypatch.delenv("SMST_VOC_WIDE", raising=False)
            assert np.abs(outs[0]).max() > 0.05
            assert np.array_equal(outs[0], outs[1]), (geometry, C, float(np.abs(outs[0] - outs[1]).max()))


def case_vocn_writer_forms(lib, monkeypatch, channel_counts=(3, 8), n=12000, extra_geometries=()):
    """kVocoderN's writer wave: whole 128-byte lines through slabs of LDS (the default, round 6), 64-byte half lines held in
    registers (SMST_VOCN_HALF_LINES=1) and the first form's 32-byte sectors (=0) move the same values: bit-identical, at two
    vertical steps (two skews of the rows against the line grid), over two calls, with a frequency map on half of the streams."""
    pkg = package()
    # (block 68 -> 40 bands: not a multiple of 16, the default falls back to half lines; block 36 -> 20 bands: to sectors)
    for geometry in (dict(block=512, interval=128, split=False), dict(block=1024, interval=192, split=False), dict(block=68, interval=17, split=False), dict(block=36, interval=9, split=False)) + tuple(extra_geometries):
        for C in channel_counts:
            xs = np.stack([synth_input(s, C, n, 48000)*(1 + 0.2*np.arange(C))[:, None].astype(np.float32) for s in range(4)])
            if geometry.get("block", 1000) < 100:
                xs = xs[:, :, :n//4]
            outs = []
            for form in ("2", "1", "0"):
                monkeypatch.setenv("SMST_VOCN_HALF_LINES", form)
                before = pkg.launch_count("vocoder_n", lib)
                b = pkg.StretchBatch(4, C, lib=lib, **geometry)
                b.setTransposeSemitones(3.0, 0.2, stream=1)
                b.setTransposeSemitones(-4.0, 0.0, stream=3)
                m = xs.shape[2]
                y1 = np.array(b.process(xs[:, :, :m//3], int(m//3*1.3)), copy=True)
                y2 = np.array(b.process(xs[:, :, m//3:], int((m - m//3)*1.3)), copy=True)
                b.close()
                assert pkg.launch_count("vocoder_n", lib) > before, (geometry, C, "kVocoderN did not run")
                outs.append(np.concatenate([y1, y2], axis=2))
            monkeypatch.delenv("SMST_VOCN_HALF_LINES", raising=False)
            assert np.abs(outs[0]).max() > 0.05
            for o in outs[1:]:
                assert np.array_equal(outs[0], o), (geometry, C, float(np.abs(outs[0] - o).max()))


def case_feed_fusion_equals_separate(lib, monkeypatch, channel_counts=(2, 3), geometry=None, n=9000, formants=False, bases_given=False):
    """Pass A folded into the feed kernel (tiles with a pitch map and no formant processing) against the separate kPredictA
    (SMST_NO_FEED_FUSION=1): the same arithmetic on the same operands, so bit-identical -- mapped and unmapped streams side
    by side, two calls.  `formants`: tiles with formant processing fold pass A into the envelope kernel (the ratios stay in LDS);
    `bases_given`: every formant stream has a base frequency, so the whole feed stage is ONE kernel (kFeedScanA<.., FUSE_FORM>, round 6) --
    compared with the two-pass form (SMST_NO_FEED_FUSION=2) and the separate kernels (=1); the launch counter proves which ran."""
    pkg = package()
    geometry = geometry or dict(block=512, interval=128, split=False)
    for C in channel_counts:
        xs = np.stack([synth_input(s, C, n, 48000)*(1 + 0.2*np.arange(C))[:, None].astype(np.float32) for s in range(4)])
        outs = []
        for mode in ((None, "2", "1") if bases_given else (None, "1")):
            if mode:
                monkeypatch.setenv("SMST_NO_FEED_FUSION", mode)
            else:
                monkeypatch.delenv("SMST_NO_FEED_FUSION", raising=False)
            before = pkg.launch_count("feed_one_pass", lib)
            b = pkg.StretchBatch(4, C, lib=lib, **geometry)
            b.setTransposeSemitones(5.0, 0.2, stream=0)
            b.setTransposeSemitones(-7.0, 0.0, stream=2)
            if formants:
                b.setFormantFactor(1.0, True, stream=0)
                b.setFormantBase(200/48000, stream=0)
                b.setFormantSemitones(3.0, False, stream=3)   # (formant processing without a pitch map; streams 1 and 2 have none)
                if bases_given:
                    b.setFormantBase(150/48000, stream=3)
            y1 = np.array(b.process(xs[:, :, :n//3], int(n//3*0.9)), copy=True)
            y2 = np.array(b.process(xs[:, :, n//3:], int((n - n//3)*0.9)), copy=True)
            b.close()
            grew = pkg.launch_count("feed_one_pass", lib) - before
            assert (grew > 0) == (bases_given and mode is None), (mode, grew)
            outs.append(np.concatenate([y1, y2], axis=2))
        monkeypatch.delenv("SMST_NO_FEED_FUSION", raising=False)
        assert np.abs(outs[0]).max() > 0.05
        for other in outs[1:]:
            assert np.array_equal(outs[0], other), (C, float(np.abs(outs[0] - other).max()))


def case_single_hop_chunks(lib, geometry=None, channel_counts=(1, 2, 3), hops=15, setup=None):
    """Calls that fire ONE hop each (the real-time pattern) run the single-hop recurrence kernel (kVocoderOne); one call with
    all the hops runs the skewed-wavefront kernels.  Same records, same order of operations: bit-identical outputs."""
    pkg = package()
    geometry = geometry or dict(block=512, interval=128, split=False)
    for C in channel_counts:
        b = pkg.StretchBatch(3, C, lib=lib, **geometry)
        if setup:
            setup(b)
        I = b.intervalSamples()
        n_out, n_in = hops*I, (hops*I*4)//5
        assert n_in*5 == n_out*4  # an exact ratio, so that chunk boundaries fall on the one-call time map (no .5 roundings)
        xs = np.stack([synth_input(s, C, n_in, 48000)*(1 + 0.3*np.arange(C))[:, None].astype(np.float32) for s in range(3)])
        whole = np.array(b.process(xs, n_out), copy=True)
        b.reset()
        parts = []
        for k in range(hops):  # the same time map as the single call: input chunk k ends where hop k+1's window ends
            lo, hi = int(round(k*I*float(n_in)/n_out)), int(round((k + 1)*I*float(n_in)/n_out))
            parts.append(np.array(b.process(xs[:, :, lo:hi], I, in_samples=hi - lo), copy=True))
        b.close()
        chunked = np.concatenate(parts, axis=2)
        assert np.abs(whole).max() > 0.05
        assert np.array_equal(whole, chunked), (C, float(np.abs(whole - chunked).max()))


TOL_HALF_MAGNITUDE = 2e-3  # fp16 state: |output| per hop vs the fp32 checker (half has 11 significant bits: 4.9e-4 per value)


def case_half_state(lib, ref, cfg, channels, stretch, label, setup=None, hops=30, streams=(0, 1, 2)):
    """BASELINE config 5 "fp16 internal" (SMST_FLAG_HALF_STATE): Band.output, Prediction.energy and the overlap-add sums that
    outlive a tile are stored in fp16, arithmetic stays fp32.  The reference has no such mode, so the comparison is against
    the fp32 checker in the magnitude domain: hop by hop (every call fires one hop, so the state
    goes through fp16 between ALL hops -- the worst case) |output| within TOL_HALF_MAGNITUDE, the output level within 1 %,
    and the samples of the first ten hops (phase errors of 5e-4 rad per hop, amplified by the recurrence) within 5 %."""
    pkg = package()
    sr = int(cfg.get("sample_rate", 48000))
    S = len(streams)
    kw = dict(preset=cfg["preset"], sample_rate=cfg.get("sample_rate", 48000.0)) if cfg.get("preset") in ("default", "cheaper") else \
        dict(block=cfg["block"], interval=cfg["interval"], split=cfg.get("split", False))
    b = pkg.StretchBatch(S, channels, lib=lib, half_state=True, **kw)
    assert b.lib.smst_batch_half_state(b.h) == 1
    refs = [make("ref", lib, ref, channels, cfg, setup, seed=i) for i, _ in enumerate(streams)]  # stream i of a batch = the instance seeded seed + i
    if setup:
        setup(b)
    I = b.intervalSamples()
    n_in = _hop_io(I, stretch, hops)[1] + 8
    xs = np.stack([synth_input(s, channels, n_in, sr) for s in streams])
    ys, os_ = [], []
    worst = 0.0
    for k in range(hops):
        lo, hi = _hop_io(I, stretch, k)
        ys.append(np.array(b.process(xs[:, :, lo:hi] if hi > lo else np.zeros((S, channels, 1), np.float32), I, in_samples=hi - lo), copy=True))
        outs = [r.process(xs[i][:, lo:hi], I) for i, r in enumerate(refs)]
        os_.append(np.stack(outs))
        if k >= 4:
            for i, r in enumerate(refs):
                e = rel_rms(np.abs(b.debug_state(i, 2)), np.abs(r.bands_complex(2)))
                worst = max(worst, e)
                assert e <= TOL_HALF_MAGNITUDE, "%s: stream %d hop %d: |output| rel-RMS %.3e" % (label, streams[i], k, e)
    b.close()
    y, o = np.concatenate(ys, axis=2), np.concatenate(os_, axis=2)
    early = slice(0, min(hops, 10)*I)
    figures = dict(magnitude=worst, level=0.0, early_samples=0.0)
    for i in range(S):
        ra, rb = np.sqrt(np.mean(y[i]**2)), np.sqrt(np.mean(o[i]**2))
        figures["level"] = max(figures["level"], abs(ra/rb - 1))
        figures["early_samples"] = max(figures["early_samples"], rel_rms(y[i][:, early], o[i][:, early]))
        assert abs(ra/rb - 1) < 0.01, (label, streams[i], ra, rb)
    assert figures["early_samples"] < 5e-2, (label, figures)
    return figures


def case_batch_ragged(lib, ref, cfg=SMALL, S=5, n=6000):
    """Batch API: per-stream parameters and ragged lengths; every stream equals its own single-stream reference run."""
    pkg = package()
    C, sr = 2, 48000
    xs = np.stack([synth_input(s, C, n, sr) for s in range(S)])
    rng = np.random.Generator(np.random.PCG64(5))
    stretch = rng.uniform(0.75, 1.5, S)
    semis = rng.uniform(-12, 12, S)
    nin = [n - 317*s for s in range(S)]
    nout = [int(round(nin[s]*stretch[s])) for s in range(S)]
    b = pkg.StretchBatch(S, C, block=cfg["block"], interval=cfg["interval"], split=cfg["split"], lib=lib)
    for s in range(S):
        b.setTransposeSemitones(float(semis[s]), 0.0, stream=s)
    y = b.process(xs, nout, in_samples=nin)
    for s in range(S):
        setup = lambda o, s=s: o.setTransposeSemitones(float(semis[s]), 0.0)  # noqa: E731
        r = make("ref", lib, ref, C, cfg, setup, seed=s)  # stream s of a batch = the instance seeded seed + s
        o = r.process(xs[s][:, :nin[s]], nout[s])
        o2 = [make("ref", lib, ref, C, cfg, setup, seed=s).process(perturbed(xs[s][:, :nin[s]], seed), nout[s]) for seed in SELF_SEEDS]
        assert_parity(y[s][:, :nout[s]], o, o2, cfg["interval"], "batch stream %d" % s, require_informative=False)
        ra, rb = np.sqrt(np.mean(y[s][:, :nout[s]]**2)), np.sqrt(np.mean(o**2))  # phase-free: output level within 1 %
        assert abs(ra/rb - 1) < 0.01, (s, ra, rb)
    b.close()


def case_random_time_factor(lib, ref):
    """stretch > 2x randomises the vertical time factors (signalsmith-stretch.h:639-640); the RNG is
    implementation-defined in the reference, so only the energy is comparable."""
    C, sr = 1, 48000
    x = synth_input(0, C, 3000, sr)
    g, r = make("product", lib, ref, C, SMALL), make("ref", lib, ref, C, SMALL)
    a, b = g.process(x, 9000), r.process(x, 9000)
    ra, rb = np.sqrt(np.mean(a[:, 2000:]**2)), np.sqrt(np.mean(b[:, 2000:]**2))
    assert abs(ra/rb - 1) < 0.1


def case_random_time_factor_seeds(lib, ref, cfg=None, streams=8, stretch=2.5, seconds=1.0, level_tol=0.02):
    """stretch > 2x at the preset geometry (signalsmith-stretch.h:639-640: uniform(4 - tf, tf) per bin and direction).  The reference
    draws from std::default_random_engine, implementation-defined, so SAMPLES are not comparable -- what is: (a) the output level
    of every stream against the checker within `level_tol`; (b) the seeded-constructor contract of :38-39,:616 on the product
    side: two instances with the same seed agree bit for bit, as does stream 0 of a batch with the single-stream object of that
    seed; another seed gives another output."""
    pkg = package()
    cfg = cfg or dict(preset="default", sample_rate=48000.0)
    C, sr = 2, int(cfg["sample_rate"])
    n = int(seconds*sr)
    nout = int(n*stretch)
    xs = np.stack([synth_input(s, C, n, sr) for s in range(streams)])
    kw = dict(preset=cfg["preset"], sample_rate=cfg["sample_rate"])
    outs = {}
    for tag, seed in (("a", 7), ("b", 7), ("c", 8)):
        b = pkg.StretchBatch(streams, C, lib=lib, seed=seed, **kw)
        outs[tag] = np.asarray(b.process(xs, nout))
        b.close()
    assert np.array_equal(outs["a"], outs["b"]), "same seed, different output"
    assert not np.array_equal(outs["a"], outs["c"]), "the seed has no effect"
    single = pkg.SignalsmithStretch(seed=7, lib=lib)
    single.presetDefault(C, sr) if cfg["preset"] == "default" else single.presetCheaper(C, sr)
    assert np.array_equal(single.process(xs[0], nout), outs["a"][0]), "batch stream 0 differs from the single-stream object of the same seed"
    worst = 0.0
    skip = 2*int(0.12*sr)
    for s in range(streams):
        r = make("ref", lib, ref, C, cfg, seed=7 + s)  # stream s of a batch = the reference instance seeded with seed + s
        o = r.process(xs[s], nout)
        ra, rb = np.sqrt(np.mean(outs["a"][s][:, skip:]**2)), np.sqrt(np.mean(o[:, skip:]**2))
        worst = max(worst, abs(ra/rb - 1))
        assert abs(ra/rb - 1) < level_tol, ("level at %.1fx" % stretch, s, ra, rb)
        assert np.isfinite(outs["a"][s]).all()
    return dict(level=worst)


def case_random_time_factor_parity(lib, ref, geometries=(SMALL,), seeds=(0, 12345, -7)):
    """Beyond 2x the reference draws a time factor per bin and direction from std::default_random_engine (:639-640, :749, :769) --
    implementation-defined in general, but defined for the checker (g++ / libstdc++: minstd_rand0 through
    uniform_real_distribution<float>), and the product replicates exactly that engine (smst_kernels_common.h: engineDraw; the host advances
    every stream's engine state by 2M - 2 draws per randomised hop).  So with the same seed the SAMPLES are comparable: teacher-forced
    single hops at 2.5x, free-running 2.5x and 4x over short horizons, output without input, and a flush of several intervals."""
    worst = {}
    for cfg in geometries:
        name = cfg.get("preset", "configure")
        worst["forced/" + name] = case_teacher_forced(lib, ref, cfg, 2, 2.5, "forced 2.5x " + name)
        scale = 1 if cfg.get("preset") == "configure" else 11
        C, sr = 2, int(cfg.get("sample_rate", 48000))
        for seed in seeds:
            x = synth_input(1, C, 4000*scale, sr) + 0.3*synth_input(5, C, 4000*scale, sr)

            def run(play, label, seed=seed):
                g, r = make("product", lib, ref, C, cfg, seed=seed), make("ref", lib, ref, C, cfg, seed=seed)
                y, o = play(g, x), play(r, x)
                o2 = [play(make("ref", lib, ref, C, cfg, seed=seed), perturbed(x, k)) for k in SELF_SEEDS]
                assert_parity(y, o, o2, r.intervalSamples(), "%s (%s, seed %d)" % (label, name, seed))
                return rel_rms(y, o)
            worst["2.5x/%s/%d" % (name, seed)] = run(lambda ob, xx: ob.process(xx[:, :1200*scale], 3000*scale), "free-running 2.5x")
            worst["4x/%s/%d" % (name, seed)] = run(lambda ob, xx: ob.process(xx[:, :600*scale], 2400*scale), "free-running 4x")
            worst["output-only/%s/%d" % (name, seed)] = run(
                lambda ob, xx: np.concatenate([ob.process(xx[:, :1500*scale], 1500*scale), ob.process(xx[:, :0], 300*scale), ob.process(xx[:, 1500*scale:2500*scale], 1000*scale)], axis=1),
                "output without input")
            worst["flush/%s/%d" % (name, seed)] = run(
                lambda ob, xx: np.concatenate([ob.process(xx[:, :2000*scale], 2400*scale), ob.flush(700*scale)], axis=1), "flush of 5 intervals")
    return worst


def case_sub_batches(lib, ref, monkeypatch):
    """A tiny workspace budget forces the engine to process the streams in several sub-batches (and several tiles
    each): results must not depend on it."""
    pkg = package()
    C, sr, S, n = 2, 48000, 5, 128*70
    xs = np.stack([synth_input(s, C, n, sr) for s in range(S)])
    nout = int(n*1.25)
    b = pkg.StretchBatch(S, C, block=512, interval=128, lib=lib)
    whole = b.process(xs, nout)
    b.close()
    full = None
    monkeypatch.setenv("SMST_WORKSPACE_GIB", "0.009")  # ~9 MiB per workspace: 2 streams per sub-batch at this geometry
    b = pkg.StretchBatch(S, C, block=512, interval=128, lib=lib)
    assert b.workspaceBytes() <= 2*9.7e6
    split = b.process(xs, nout)
    b.close()
    assert np.array_equal(whole, split)


def case_cmd_main_flow(lib, ref, sr=44100, seconds=1.5, time_factor=1.0, semitones=0.0, channels=1):
    """BASELINE config 1: the call sequence of the reference's CLI (cmd/main.cpp:44-82) -- presetDefault,
    setTransposeSemitones(st, 8000/sr), setFormantSemitones(0), setFormantBase(100/sr), outputSeek, process, flush --
    on a mono 44.1 kHz stream at 1.0x / 0 st, product vs checker; at 1.0x the result is also the input itself."""
    n = int(sr*seconds)
    x = synth_input(0, channels, n, sr)

    def play(o, xx):
        cfg = dict(preset="default", sample_rate=float(sr))
        del cfg
        o.setTransposeSemitones(semitones, 8000.0/sr)   # cmd/main.cpp:46 (defaults :22-28)
        o.setFormantSemitones(0.0, False)                # :47
        o.setFormantBase(100.0/sr)                       # :48
        out_len = int(round(n*time_factor))              # :36
        seek_len = o.outputSeekLength(1/time_factor)     # :58
        o.outputSeek(xx[:, :seek_len])                   # :59
        output_index = out_len - o.intervalSamples()     # :62
        output_pos = output_index + o.outputLatency()    # :65
        input_pos = int(round(output_pos/time_factor))   # :67
        input_index = input_pos + o.inputLatency()       # :69
        padded = np.zeros((channels, max(input_index, n)), np.float32)  # inWav.resize(inputIndex), :73
        padded[:, :n] = xx
        a = o.process(padded[:, seek_len:input_index], output_index)    # :77-78
        b = o.flush(out_len - output_index)                              # :81-82
        return np.concatenate([a, b], axis=1)
    y, o = check_scenario(lib, ref, dict(preset="default", sample_rate=float(sr)), x, play, "cmd/main.cpp flow")
    if time_factor == 1.0 and semitones == 0.0:
        # the input itself, apart from the pre-roll fold-back in the first interval (reference: 7e-3 there) and the
        # flush fade in the last one
        head = 1323 if sr == 44100 else 1440
        tail = 2*head
        assert rel_rms(y[:, head:-tail], x[:, head:y.shape[1] - tail]) < 5e-6
        assert rel_rms(o[:, head:-tail], x[:, head:o.shape[1] - tail]) < 5e-6


# ---------------------------------------------------------------------------------------------------------------
# Chaos-free parity instruments (SURVEY.md App. D.2 i, iv, v): teacher-forced single hop, per-hop magnitude spectra,
# discrete-decision agreement.  They use the state hooks of both sides (smst_batch_debug_* / smst_ref_get_*).
# ---------------------------------------------------------------------------------------------------------------
TOL_FORCED_SPECTRUM = 1e-5   # rel-RMS of Band.output after ONE hop from injected checker state (D.2 i)
TOL_FORCED_SAMPLES = 1e-5    # rel-RMS of the interval of samples that hop emits
TOL_MAGNITUDE = 1e-4         # per-hop rel-RMS of |output_c[b]| (phase-free, = sqrt(Prediction.energy), stretch.h:596-603)


def _crel(a, b):
    """rel-RMS distance of two complex arrays."""
    a, b = np.asarray(a, np.complex128), np.asarray(b, np.complex128)
    return float(np.sqrt(np.mean(np.abs(a - b)**2)/max(np.mean(np.abs(b)**2), 1e-300)))


def _crel_trimmed(a, b, drop=0.01):
    """rel-RMS distance after dropping the `drop` fraction of bins with the largest error: with a frequency map a few
    near-silent bins have an ill-conditioned phase (four tiny prediction terms that nearly cancel), and one ulp of the
    map's inputBin turns them around; they carry no energy but dominate an untrimmed spectrum distance."""
    a, b = np.asarray(a, np.complex128).ravel(), np.asarray(b, np.complex128).ravel()
    d = np.sort(np.abs(a - b)**2)
    keep = d[:max(1, int(round(len(d)*(1 - drop))))]
    return float(np.sqrt(np.sum(keep)/max(np.sum(np.abs(b)**2), 1e-300)))


MAP_AGREE = 2e-3    # output-map entries (fractional input bins, up to 3072: ulp 2.4e-4) count as equal within this many bins
MARGIN_FLIP = 2e-3  # a peak-run boundary decided by less than this (relative) is within reach of the arithmetic difference


def _flip_margin(batch, stream, r):
    """When one hop disagrees by more than the smooth bound, the cause has to be a discrete decision that was a near-tie.
    findPeaks (stretch.h:859-880) starts / ends a run where energy[b] > smoothedEnergy[b] changes: returns the smallest
    relative margin |energy - smoothed|/smoothed among the checker's run boundaries in the region where the two output maps
    differ (None if the maps agree or there is no map) -- the caller accepts the hop only if that margin is below
    MARGIN_FLIP, i.e. the product took the other side of a comparison the checker decided by a hair."""
    m = batch.debug_map(stream)
    if m is None:
        return None
    mr = r.output_map()
    bad = np.nonzero(np.abs(m[:, 0] - mr[:, 0]) > MAP_AGREE)[0]
    if len(bad) == 0:
        return None
    en, sm = r.energy()
    pk = r.peaks()
    near = [p for p in pk if bad.min() - 40 <= p[1] <= bad.max() + 40]
    if not near:
        return None
    lo = max(1, int(min(p[0] for p in near)) - 16)
    hi = min(len(en) - 1, int(max(p[0] for p in near)) + 17)
    above = en[lo - 1:hi + 1] > sm[lo - 1:hi + 1]
    edges = np.nonzero(above[1:] != above[:-1])[0] + lo - 1  # boundary between bins b and b+1
    if len(edges) == 0:
        return None
    cand = np.concatenate([edges, edges + 1])
    return float(np.min(np.abs(en[cand] - sm[cand])/np.maximum(sm[cand], 1e-30)))


ARGMAX_TIE_GAP = 1e-3   # an arg-max call the CHECKER decided by less than this (relative gap of its two largest channel energies) is a near-tie:
#                         five times the largest deviation of the product's Prediction.energy measured at such a bin (2e-4: bin 2047 of
#                         config 5's noise stream, profiles/r4_config5_forced_hop_bins.txt) -- a constant of the checker's data, not of the product's
ARGMAX_REACH_BINS = 64  # bins above a flipped call that inherit its phase through the b-1 / b-L taps (stretch.h:748-762): the difference
#                         decays to the hop's floor within ~40 bins (checker against its perturbed twin on config 5, 24 hops x 3 streams)
EXCUSED_SHARE = 1e-4          # a hop counts as EXCUSED when its near-tie regions hold more than this share of the checker's spectrum energy
TOL_EXCUSED_MAGNITUDE = 1e-4  # |output| inside an excused region: magnitudes do not depend on the arg-max choice (stretch.h:596-603)
ARGMAX_NOISE = 8e-6  # fp32 rounding noise of a bin, as a fraction of the hop's PEAK amplitude, times 2 x 4: the analysis spectra of two fp32
#                      implementations differ by ~1e-6 of the peak amplitude per bin (1.9e-7 rel-RMS over the spectrum, case_teacher_forced:
#                      `analysis`), so a bin of amplitude a has an energy that is uncertain by 2 * 1e-6 * peak / a RELATIVE -- percents at -80 dB.
#                      Near-tie threshold of a bin = max(ARGMAX_TIE_GAP, ARGMAX_NOISE * sqrt(largest bin energy of the hop / the bin's energy))


def _energy_reach(e_twin, e_r):
    """How far a 1e-6 perturbation of the input moves the checker's own Prediction.energy in this hop: (99.9th percentile, maximum) of
    the relative deviation over the bins that carry energy (above 1e-4 of the hop's largest).  With a frequency map the energies are
    interpolated at map positions that move with the peak centroids: 1e-3 at one bin in a hundred, 4e-3 at one in a thousand, 2e-2 ..
    6e-2 at the worst bin (noise stream of config 5) -- the product's own deviations have the same distribution."""
    e_r, e_t = np.asarray(e_r, np.float64), np.asarray(e_twin, np.float64)
    sig = e_r > 1e-4*e_r.max()
    if not np.any(sig):
        return 0.0, 0.0
    rel = np.abs(e_t - e_r)[sig]/e_r[sig]
    return float(np.percentile(rel, 99.9)), float(rel.max())


def _argmax_mask(e_other, e_r, e_twin=None):
    """Per-BIN treatment of the one discrete decision of the recurrence, the maximum-energy channel of a bin (stretch.h:729-737, first
    maximum wins).  e_r / e_other / e_twin: Prediction.energy [C][M] after the hop of the checker / of the implementation under test /
    of the checker that has seen the perturbed input.
    Returns (mask, ties, unexplained): `mask` marks the bins whose phase may legitimately differ -- every bin at which the two call
    the maximum channel differently AND the checker's own call was a near-tie AND the other side's energies of the two channels are
    themselves no further from the checker's than the perturbed checker's energies get in this hop -- plus the ARGMAX_REACH_BINS bins
    above it.  Near-tie: the relative gap of the checker's two largest channel energies is within max(ARGMAX_TIE_GAP, the bin's fp32
    noise, twice the 99.9th percentile of the perturbed checker's energy deviations): all three are the checker's own data.
    `unexplained` counts differing calls that were no near-tie (a defect, or a frequency map that differs -- the caller decides)."""
    e_r, e_o = np.asarray(e_r, np.float64), np.asarray(e_other, np.float64)
    M = e_r.shape[1]
    mask = np.zeros(M, bool)
    if e_r.shape[0] < 2:
        return mask, 0, 0
    diff = np.nonzero(np.argmax(e_r, axis=0) != np.argmax(e_o, axis=0))[0]
    ties = unexplained = 0
    peak = float(e_r.max())
    p999, worst = _energy_reach(e_twin, e_r) if e_twin is not None else (0.0, 0.0)
    for bn in diff:
        order = np.argsort(e_r[:, bn])
        c1, c2 = order[-1], order[-2]
        top = max(e_r[c1, bn], 1e-300)
        gap = (e_r[c1, bn] - e_r[c2, bn])/top
        dev = max(abs(e_o[c1, bn] - e_r[c1, bn]), abs(e_o[c2, bn] - e_r[c2, bn]))/top
        noise = ARGMAX_NOISE*np.sqrt(peak/top)
        if gap <= max(ARGMAX_TIE_GAP, noise, 2*p999) and dev <= max(ARGMAX_TIE_GAP, noise, 2*worst):
            ties += 1
            mask[bn:bn + ARGMAX_REACH_BINS] = True
        else:
            unexplained += 1
    return mask, ties, unexplained


def _masked_distances(out_o, out_r, mask, trim=0.0):
    """(distance of the complex spectra over the bins OUTSIDE the mask, distance of the magnitudes INSIDE it), both relative to the
    checker's whole spectrum.  trim: drop that fraction of the outside bins with the largest error first (formant processing: a few
    near-silent bins are amplified by the envelope ratio and dominate an untrimmed distance, see _crel_trimmed)."""
    o, r = np.asarray(out_o, np.complex128), np.asarray(out_r, np.complex128)
    total = max(float(np.sum(np.abs(r)**2)), 1e-300)
    d = np.sum(np.abs(o[:, ~mask] - r[:, ~mask])**2, axis=0)
    if trim > 0 and len(d):
        d = np.sort(d)[:max(1, int(round(len(d)*(1 - trim))))]
    outside = float(np.sqrt(np.sum(d)/total))
    inside = float(np.sqrt(np.sum((np.abs(o[:, mask]) - np.abs(r[:, mask]))**2)/total))
    return outside, inside


def _hop_io(interval, stretch, k):
    """Input range consumed by hop-aligned call number k (each call emits exactly one interval)."""
    lo = int(round(k*interval/stretch))
    hi = int(round((k + 1)*interval/stretch))
    return lo, hi


def _inject(batch, stream, r):
    """Product state := checker state (Band.input/.prevInput/.output, Prediction.energy, overlap-add ring)."""
    B, I = r.blockSamples(), r.intervalSamples()
    batch.debug_set_state(stream, 0, r.bands_complex(0))
    batch.debug_set_state(stream, 1, r.bands_complex(1))
    batch.debug_set_state(stream, 2, r.bands_complex(2))
    batch.debug_set_state(stream, 3, r.bands_real(4))
    sums, prods = r.output_ring()  # from the read position: index 0 = the next output sample
    cs = np.zeros((r.channels, B + I), np.float32)
    cp = np.full(B + I, 1e-30, np.float32)
    cs[:, :B] = sums
    cp[:B] = prods
    batch.debug_set_carry(stream, cs, cp)


WELL_CONDITIONED = 1e-3  # a bin whose Band.output moves by less than this (relative) when the checker's own input is perturbed by PERTURBATION


def _well_conditioned(ro, to, excused):
    """Bins of one hop on which a comparison of Band.output asserts something: the perturbed checker's value stays within
    WELL_CONDITIONED of the checker's (relative to the bin's own magnitude), the bin carries energy (above 1e-6 of the hop's
    largest), and it lies outside the arg-max near-tie regions.  Returns the mask [channels][bins]."""
    ro, to = np.asarray(ro, np.complex128), np.asarray(to, np.complex128)
    mag = np.abs(ro)
    return (np.abs(to - ro) <= WELL_CONDITIONED*mag) & (mag > 1e-3*float(mag.max())) & ~np.asarray(excused, bool)[None, :]


def case_teacher_forced(lib, ref, cfg, channels, stretch, label, setup=None, warm_hops=9, forced_hops=3, streams=(0, 1, 2), gains=None,
                        cap=CAP_TONAL, trim=0.0, cap_well_conditioned=None):
    """D.2 (i): run `warm_hops` hops on both sides, then `forced_hops` times: overwrite the product's carried state with
    the checker's, run ONE hop on both, compare the emitted interval and Band.output.  Every compared hop starts from
    identical state, so nothing is amplified over time -- but one hop still has a condition number: with a frequency
    map, the rounding noise of the analysis (a few 1e-7 of the spectrum's norm) moves the peak centroids of low-energy
    regions and with them the phase advance of whole groups of bins.  Bounds, per stream and hop:

      * EVERY hop: Band.output OUTSIDE the arg-max near-tie regions (_argmax_mask: per bin, decided from the checker's data) within the
        FIXED ceiling `cap` (5e-3 stretch / pitch only, 5e-2 with formant processing: SURVEY App. D.2 iii; with formant processing
        without the `trim` share of bins with the largest error -- near-silent bins amplified by the envelope ratio); |Band.output|
        INSIDE them within TOL_EXCUSED_MAGNITUDE; no arg-max call that differs without being a near-tie -- unless the hop is an
        explained peak-run flip (_flip_margin: the checker's own energy > smoothedEnergy comparison was a near-tie and the output
        maps differ), of which at most one in eight hops may occur.
      * EVERY hop without such a flip, again OUTSIDE the near-tie regions (the product's and the perturbed checker's together): the
        smooth bound max(TOL_FORCED_SPECTRUM, SELF_FACTOR * the CHECKER'S OWN one-hop sensitivity over the same bins), per stream,
        worst hop against worst hop and typical against typical -- the sensitivity measured with a second checker instance that has
        seen the input perturbed by PERTURBATION throughout and is forced to the first one's state before each compared hop.
      * hops whose near-tie regions hold less than EXCUSED_SHARE of the spectrum's energy ("clean"): the same smooth bound on the
        emitted samples and on the overlap-add ring (a sample domain has no bins to leave out).
      * `gains`: per-channel input gains.  The bench's inputs give every channel the same amplitude (conftest.synth_input), so
        near-ties are endemic there (a quarter of the bins of a sine stream); with gains 1 - 0.07 c they are rare: the caller bounds
        `excused_hops` and `excused_bin_fraction`.
    Returns the worst figures."""
    pkg = package()
    sr = int(cfg.get("sample_rate", 48000))
    S = len(streams)
    kw = dict(preset=cfg["preset"], sample_rate=cfg.get("sample_rate", 48000.0)) if cfg.get("preset") in ("default", "cheaper") else \
        dict(block=cfg["block"], interval=cfg["interval"], split=cfg.get("split", False))
    b = pkg.StretchBatch(S, channels, lib=lib, **kw)
    # stream i of a batch is the reference instance constructed with seed + i (include/smst.h): same random engine, same draws beyond 2x
    refs = [make("ref", lib, ref, channels, cfg, setup, seed=i) for i, _ in enumerate(streams)]
    twins = [make("ref", lib, ref, channels, cfg, setup, seed=i) for i, _ in enumerate(streams)]  # the perturbed-input checkers
    if setup:
        setup(b)
    I = b.intervalSamples()
    b_bands = b.bands()
    total_hops = warm_hops + forced_hops
    n_in = _hop_io(I, stretch, total_hops)[1] + 8
    xs = np.stack([synth_input(s, channels, n_in, sr) for s in streams])
    if gains is not None:
        xs = (xs*np.asarray(gains, np.float32)[None, :, None]).astype(np.float32)
    xp = np.stack([perturbed(x, 1 + i) for i, x in enumerate(xs)])
    worst = dict(spectrum=0.0, spectrum_self=0.0, spectrum_trimmed=0.0, samples=0.0, samples_self=0.0, analysis=0.0, analysis_self=0.0,
                 ring=0.0, ring_self=0.0, spectrum_outside_ties=0.0, magnitude_inside_ties=0.0, spectrum_clean_hops=0.0, spectrum_self_clean_hops=0.0,
                 ring_clean_hops=0.0, ring_self_clean_hops=0.0, spectrum_outside_all_ties=0.0, spectrum_self_outside_all_ties=0.0, excused_hops=0, excused_bins_max=0, excused_bins_total=0,
                 excused_share_max=0.0, excused_energy_fraction=0.0, argmax_ties=0, hops=S*forced_hops)
    per = [[] for _ in streams]
    flips = 0
    for k in range(total_hops):
        lo, hi = _hop_io(I, stretch, k)
        forced = k >= warm_hops
        if forced:
            for i, r in enumerate(refs):
                _inject(b, i, r)
                twins[i].copy_state_from(r)
        y = b.process(xs[:, :, lo:hi] if hi > lo else np.zeros((S, channels, 1), np.float32), I, in_samples=hi - lo)
        outs = [r.process(xs[i][:, lo:hi], I) for i, r in enumerate(refs)]
        outs_p = [t.process(xp[i][:, lo:hi], I) for i, t in enumerate(twins)]
        if not forced:
            continue
        for i, r in enumerate(refs):
            ro, po = r.bands_complex(2), b.debug_state(i, 2)
            s_samp, s_spec = rel_rms(outs_p[i], outs[i]), _crel(twins[i].bands_complex(2), ro)
            e_samp, e_spec, e_trim = rel_rms(y[i], outs[i]), _crel(po, ro), _crel_trimmed(po, ro)
            e_ana, s_ana = _crel(b.debug_state(i, 0), r.bands_complex(0)), _crel(twins[i].bands_complex(0), r.bands_complex(0))
            # the overlap-add ring AFTER the hop holds the frame this hop synthesised: the sample-domain leg that is live in
            # split mode too (there the emitted interval comes from the injected ring alone, and `samples` compares 0 with 0)
            ring_r, _ = r.output_ring()
            ring_t, _ = twins[i].output_ring()
            ring_p = b.debug_carry(i)[0][:, :ring_r.shape[1]]
            e_ring, s_ring = rel_rms(ring_p, ring_r), rel_rms(ring_t, ring_r)
            for key, v in (("samples", e_samp), ("samples_self", s_samp), ("spectrum", e_spec), ("spectrum_self", s_spec),
                           ("spectrum_trimmed", e_trim), ("analysis", e_ana), ("analysis_self", s_ana), ("ring", e_ring), ("ring_self", s_ring)):
                worst[key] = max(worst[key], v)
            # the discrete decisions of the hop: per BIN for the arg-max channel, per hop for the peak runs of a frequency map
            e_r = r.bands_real(4)
            e_t = twins[i].bands_real(4)
            mask, ties, unexplained = _argmax_mask(b.debug_state(i, 3), e_r, e_t)
            mask_t, ties_t, _ = _argmax_mask(e_t, e_r, e_t)  # the perturbed checker flips the same kind of call: its hop is no measure of a smooth response either
            outside, inside = _masked_distances(po, ro, mask, trim)
            union = mask | mask_t
            out_p, _ = _masked_distances(po, ro, union)
            out_t, _ = _masked_distances(twins[i].bands_complex(2), ro, union)
            share_t = float(np.sum(np.abs(np.asarray(ro, np.complex128)[:, mask_t])**2)/max(np.sum(np.abs(np.asarray(ro, np.complex128))**2), 1e-300))
            # Where the checker's OWN one-hop response is small the comparison is well-conditioned whatever the scenario: with formant
            # processing (near-silent bins scaled by the envelope ratio: the whole-spectrum figure is dominated by bins whose value the
            # checker itself does not hold to 1e-1) this is the leg that asserts something -- bound `cap_well_conditioned`
            wc = _well_conditioned(ro, twins[i].bands_complex(2), union)
            roc, poc = np.asarray(ro, np.complex128), np.asarray(po, np.complex128)
            share_wc = float(np.sum(np.abs(roc[wc])**2)/np.sum(np.abs(roc)**2)) if wc.any() else 0.0
            worst["well_conditioned_energy_share_min"] = min(worst.get("well_conditioned_energy_share_min", 1.0), share_wc)
            if share_wc >= 0.25:  # (a hop in which the checker itself holds less than a quarter of the spectrum's energy to 1e-3 has no such leg)
                d_wc = float(np.sqrt(np.sum(np.abs(poc[wc] - roc[wc])**2)/np.sum(np.abs(roc[wc])**2)))
                worst["spectrum_well_conditioned"] = max(worst.get("spectrum_well_conditioned", 0.0), d_wc)
                worst["well_conditioned_hops"] = worst.get("well_conditioned_hops", 0) + 1
                if cap_well_conditioned is not None:
                    assert d_wc <= cap_well_conditioned, "%s: stream %d, forced hop %d: Band.output over the well-conditioned bins (%.0f %% of the energy) %.3e > %.1e" % (
                        label, streams[i], k - warm_hops, 100*share_wc, d_wc, cap_well_conditioned)
            # ... and the phase-free leg over EVERY bin (|Band.output| = sqrt(Prediction.energy), :596-603: the formant ratio, the map and the
            # interpolation of the energies are all in it, the phase of near-silent bins scaled up by the ratio is not)
            d_mag = float(np.sqrt(np.sum((np.abs(poc) - np.abs(roc))**2)/np.sum(np.abs(roc)**2)))
            s_mag = float(np.sqrt(np.sum((np.abs(np.asarray(twins[i].bands_complex(2), np.complex128)) - np.abs(roc))**2)/np.sum(np.abs(roc)**2)))
            worst["magnitude_all_bins"] = max(worst.get("magnitude_all_bins", 0.0), d_mag)
            worst["magnitude_all_bins_self"] = max(worst.get("magnitude_all_bins_self", 0.0), s_mag)
            worst["excused_bins_total"] += int(mask.sum())
            share = float(np.sum(np.abs(np.asarray(ro, np.complex128)[:, mask])**2)/max(np.sum(np.abs(np.asarray(ro, np.complex128))**2), 1e-300))
            margin = _flip_margin(b, i, r)
            flipped = margin is not None and margin < MARGIN_FLIP  # the output maps differ and the checker's run boundary there was a near-tie
            flips += int(flipped)
            worst["argmax_ties"] += ties
            worst["excused_hops"] += int(share > EXCUSED_SHARE)  # (near-ties among bins at the noise floor excuse nothing that carries energy)
            worst["excused_share_max"] = max(worst["excused_share_max"], share)
            worst["excused_energy_fraction"] += share/(S*forced_hops)  # (mean over the hops of the energy share inside excused regions)
            worst["excused_bins_max"] = max(worst["excused_bins_max"], int(mask.sum()))
            where = "%s: stream %d, forced hop %d" % (label, streams[i], k - warm_hops)
            if not flipped:
                assert unexplained == 0, "%s: %d arg-max channel calls differ where the checker's call was no near-tie" % (where, unexplained)
                assert outside <= cap, "%s: Band.output outside the near-tie regions %.3e > the fixed ceiling %.1e (%d bins excused)" % (where, outside, cap, int(mask.sum()))
                worst["spectrum_outside_ties"] = max(worst["spectrum_outside_ties"], outside)
            assert inside <= TOL_EXCUSED_MAGNITUDE or flipped, "%s: |Band.output| inside the near-tie regions %.3e > %.1e" % (where, inside, TOL_EXCUSED_MAGNITUDE)
            worst["magnitude_inside_ties"] = max(worst["magnitude_inside_ties"], 0.0 if flipped else inside)
            clean = share <= EXCUSED_SHARE and share_t <= EXCUSED_SHARE and not flipped
            if clean:
                for key, v in (("spectrum_clean_hops", e_spec), ("spectrum_self_clean_hops", s_spec), ("ring_clean_hops", e_ring), ("ring_self_clean_hops", s_ring)):
                    worst[key] = max(worst[key], v)
            if not flipped:
                per[i].append((e_samp, s_samp, out_p, out_t, e_ring, s_ring, float(clean)))
                worst["spectrum_outside_all_ties"] = max(worst["spectrum_outside_all_ties"], out_p)
                worst["spectrum_self_outside_all_ties"] = max(worst["spectrum_self_outside_all_ties"], out_t)
    b.close()
    assert flips <= max(1, (S*forced_hops)//8), (label, "too many flipped peak-run decisions", flips)
    worst["flips"] = flips
    worst["excused_bin_fraction"] = worst["excused_bins_total"]/float(S*forced_hops*b_bands)
    # per stream: every un-flipped hop's spectrum outside the near-tie regions, and the clean hops' samples and ring, against SELF_FACTOR x
    # the checker's own one-hop sensitivity -- typical hop against typical hop as well (one pathological hop of the checker must not widen
    # everything: the chirp stream under formant compensation has one-hop sensitivities from 4e-3 to 1.7, amplified near-silent bins)
    clean_total = 0
    for i in range(S):
        a = np.array(per[i]).reshape(-1, 7)
        for rows, cols in ((a, ((2, "spectrum outside the near-tie regions", TOL_FORCED_SPECTRUM),)),
                           (a[a[:, 6] > 0], ((0, "samples", TOL_FORCED_SAMPLES), (4, "ring", TOL_FORCED_SAMPLES)))):
            if len(rows) == 0:
                continue
            for col, name, tol in cols:
                assert np.median(rows[:, col]) <= max(tol, SELF_FACTOR*np.median(rows[:, col + 1])), (label, streams[i], "median " + name, np.median(rows[:, col]), np.median(rows[:, col + 1]))
                bound = max(tol, SELF_FACTOR*rows[:, col + 1].max())
                assert rows[:, col].max() <= bound, "%s: stream %d: %s rel-RMS %.3e > %.1e (checker's one-hop sensitivity %.1e)" % (label, streams[i], name, rows[:, col].max(), bound, rows[:, col + 1].max())
        clean_total += int((a[:, 6] > 0).sum())
    worst["clean_hops"] = clean_total
    assert worst["ring"] > 0, (label, "the ring comparison is empty")
    return worst


def case_hop_magnitudes(lib, ref, cfg, channels, stretch, label, setup=None, hops=40, streams=(2, 5), tol=TOL_MAGNITUDE,
                        min_argmax_agreement=0.999, min_map_agreement=0.98, semitones=None):
    """D.2 (iv) + (v), free-running (no state injection): after every hop compare the phase-free quantities, which stay
    comparable after the phases have decorrelated (noise streams): |Band.output| per bin (= sqrt(Prediction.energy) by
    stretch.h:596-603), the arg-max channel per bin derived from Prediction.energy (:729-737), and -- with a frequency
    map -- the output map the peak list produces (:859-917).  These are functions of the INPUT only, so nothing grows over
    time; but with a frequency map they are not smooth functions of it (a peak run that gains or loses a bin moves a whole
    segment of the map).  Per hop: |output| within max(tol, SELF_FACTOR * the checker's own response to an input perturbed
    by PERTURBATION at that hop); a hop beyond that is accepted only as an EXPLAINED FLIP -- the checker's own energy /
    smoothed-energy comparison at a run boundary in the differing region was a near-tie (_flip_margin < MARGIN_FLIP) -- and
    at most 3 % of the hops may be such flips (measured on the MI355X: 1 in 228 on config 4b, bin 902 of the chirp stream
    decided by 1.9e-4 where the product's energy differs by 3.5e-4 and a 1e-6-perturbed checker's by 1.8e-4).
    Returns the figures it measured."""
    pkg = package()
    sr = int(cfg.get("sample_rate", 48000))
    S = len(streams)
    kw = dict(preset=cfg["preset"], sample_rate=cfg.get("sample_rate", 48000.0)) if cfg.get("preset") in ("default", "cheaper") else \
        dict(block=cfg["block"], interval=cfg["interval"], split=cfg.get("split", False))
    b = pkg.StretchBatch(S, channels, lib=lib, **kw)
    # `stretch` may be one factor or one per stream, `semitones` (optional) one transposition per stream: BASELINE config 5 names
    # "per-stream random stretch 0.75-1.5x and +-12 st"
    per_stream = semitones is not None or not np.isscalar(stretch)
    stretches = [float(stretch)]*S if np.isscalar(stretch) else [float(v) for v in stretch]

    def setup_of(i):
        def f(o):
            if setup:
                setup(o)
            if semitones is not None:
                o.setTransposeSemitones(float(semitones[i]), 0.0)
        return f if (setup or semitones is not None) else None
    refs = [make("ref", lib, ref, channels, cfg, setup_of(i), seed=i) for i, _ in enumerate(streams)]  # stream i of a batch = the instance seeded seed + i
    twins = [make("ref", lib, ref, channels, cfg, setup_of(i), seed=i) for i, _ in enumerate(streams)] if (setup or semitones is not None) else None
    if setup:
        setup(b)
    if semitones is not None:
        for i in range(S):
            b.setTransposeSemitones(float(semitones[i]), 0.0, stream=i)
    I = b.intervalSamples()
    n_in = max(_hop_io(I, v, hops)[1] for v in stretches) + 8
    xs = np.stack([synth_input(s, channels, n_in, sr) for s in streams])
    xp = np.stack([perturbed(x, 1 + i) for i, x in enumerate(xs)])
    agree, cells, map_ok, map_cells, flips = 0, 0, 0, 0, 0
    errs, owns = np.zeros((S, hops)), np.zeros((S, hops))
    for k in range(hops):
        spans = [_hop_io(I, v, k) for v in stretches]
        if per_stream:  # every stream its own input span of this hop, one batched call (ragged input lengths)
            width = max(1, max(hi_ - lo_ for lo_, hi_ in spans))
            chunk = np.zeros((S, channels, width), np.float32)
            for i, (lo_, hi_) in enumerate(spans):
                chunk[i, :, :hi_ - lo_] = xs[i][:, lo_:hi_]
            b.process(chunk, I, in_samples=np.array([hi_ - lo_ for lo_, hi_ in spans], np.int32))
        else:
            lo, hi = spans[0]
            b.process(xs[:, :, lo:hi] if hi > lo else np.zeros((S, channels, 1), np.float32), I, in_samples=hi - lo)
        for i, r in enumerate(refs):
            lo, hi = spans[i]
            r.process(xs[i][:, lo:hi], I)
            mo, mr = np.abs(b.debug_state(i, 2)), np.abs(r.bands_complex(2))
            if twins:
                twins[i].process(xp[i][:, lo:hi], I)
                owns[i, k] = rel_rms(np.abs(twins[i].bands_complex(2)), mr)
            if k >= 4:  # the first hops ramp up from the zero state: |output| is tiny and dominated by the window edge
                errs[i, k] = rel_rms(mo, mr)
                bound = max(tol, SELF_FACTOR*owns[i, k])
                if errs[i, k] > bound:
                    # beyond the smooth bound: only a flipped near-tie may do that (and it must be rare, below)
                    margin = _flip_margin(b, i, r)
                    assert margin is not None and margin < MARGIN_FLIP, "%s: stream %d hop %d: |output| rel-RMS %.3e > %.1e (checker's own %.1e) and no near-tie explains it (margin %s)" % (
                        label, streams[i], k, errs[i, k], bound, owns[i, k], margin)
                    flips += 1
            if channels > 1:
                eo, er = b.debug_state(i, 3), r.bands_real(4)
                loud = er.max(axis=0) > 1e-12*max(float(er.max()), 1e-30)  # ties between silent channels are not decisions
                agree += int(np.sum((np.argmax(eo, axis=0) == np.argmax(er, axis=0)) & loud))
                cells += int(np.sum(loud))
            m = b.debug_map(i)
            if m is not None:
                mr2 = r.output_map()
                map_ok += int(np.sum(np.abs(m[:, 0] - mr2[:, 0]) <= MAP_AGREE))
                map_cells += m.shape[0]
    b.close()
    worst_mag, worst_self = float(errs.max()), float(owns.max())
    within = 1.0 - flips/float(S*max(1, hops - 4))
    assert within >= 0.97, (label, "too many flipped decisions", flips)
    rates = dict(magnitude=worst_mag, magnitude_self=worst_self, hops_within_bound=within, flips=flips, argmax=(agree/cells if cells else 1.0), map=(map_ok/map_cells if map_cells else 1.0))
    assert rates["argmax"] >= min_argmax_agreement, (label, rates)
    assert rates["map"] >= min_map_agreement, (label, rates)
    return rates


def complex_helper_expectation(v):
    """The roundings csrc/smst_complex.h documents, in float32 with single-rounded fused multiply-adds (float64 products of
    float32 operands are exact, one rounding on the way back)."""
    f32, f64 = np.float32, np.float64
    ax, ay, bx, by, cx, cy, fr = (v[:, i].astype(f32) for i in range(7))

    def fma(x, y, z):
        return (x.astype(f64)*y.astype(f64) + z.astype(f64)).astype(f32)

    def mul(x, y):
        return (x.astype(f64)*y.astype(f64)).astype(f32)
    out = np.zeros((v.shape[0], 8), f32)
    out[:, 0], out[:, 1] = fma(ay, -by, mul(ax, bx)), fma(ay, bx, mul(ax, by))
    out[:, 2], out[:, 3] = fma(ay, by, mul(ax, bx)), fma(ay, bx, -mul(ax, by))
    out[:, 4], out[:, 5] = fma(ay, -by, fma(ax, bx, cx)), fma(ay, bx, fma(ax, by, cy))
    out[:, 6], out[:, 7] = fma(bx - ax, fr, ax), fma(by - ay, fr, ay)
    return out


def case_complex_helpers(lib):
    g = np.random.Generator(np.random.PCG64(77))
    v = g.standard_normal((4096, 7)).astype(np.float32)
    v[:, 6] = g.uniform(0, 1, 4096)
    v[:16, :6] = 0.0
    got = package().complex_selftest(v, lib=lib)
    want = complex_helper_expectation(v)
    # the float64 emulation of a float32 fma double-rounds in rare cases: an ulp there, exact agreement on the rest
    close = np.abs(got - want) <= np.spacing(np.abs(want).astype(np.float32))
    assert close.all(), (got[~close][:4], want[~close][:4])
    assert (got == want).mean() > 0.999


def case_fast_fft_close_to_generic(lib, monkeypatch, presets=(("cheaper", 48000), ("default", 96000), ("default", 48000), ("cheaper", 96000)), seconds=0.5):
    """The register-blocked FFT (16 x 16 x R3; R3 = 10, 12, 20, 24: every preset geometry of signalsmith-stretch.h:63-68 up to
    96 kHz) against the generic radix-4/2/3/5 ladder (SMST_NO_FAST_FFT=1) on the same input: two correct FFTs of the same data,
    1.0x / 0 st so that no phase-vocoder feedback amplifies their rounding difference; both are the identity with delay."""
    pkg = package()
    worst = {}
    for preset, sr in presets:
        C, n = 2, int(seconds*sr)
        x = np.stack([synth_input(s, C, n, sr) for s in (0, 1)])
        outs = []
        for generic in (False, True):
            if generic:
                monkeypatch.setenv("SMST_NO_FAST_FFT", "1")
            else:
                monkeypatch.delenv("SMST_NO_FAST_FFT", raising=False)
            b = pkg.StretchBatch(2, C, preset=preset, sample_rate=sr, lib=lib)
            outs.append(np.asarray(b.process(x, n)))
            lat, skip = b.inputLatency() + b.outputLatency(), 2*b.blockSamples()
            b.close()
        monkeypatch.delenv("SMST_NO_FAST_FFT", raising=False)
        d = rel_rms(outs[0], outs[1])
        assert n - lat - skip > 2000
        ident = rel_rms(outs[0][:, :, lat + skip:n], x[:, :, skip:n - lat])  # past the first hop (random time factors at 1.0x, see DESIGN)
        worst["%s@%d" % (preset, sr)] = (d, ident)
        assert d < 5e-6, (preset, sr, d)
        assert ident < 2e-6, (preset, sr, ident)
    return worst


def case_fft_teams_equals_per_frame(lib, monkeypatch, presets=(("cheaper", 48000), ("default", 48000)), seconds=0.6, streams=3, channels=2):
    """Persistent analysis / synthesis workgroups (tables loaded once per workgroup, frame after frame) against one frame per
    workgroup (SMST_FFT_TEAMS=0): the same table values and the same operations in the same order, so bit-identical -- at 1.4x with a
    pitch shift on one stream, over two calls (the second call's first hops reach into the carried history: frames that the persistent
    kernel leaves to the per-frame kernel), with ragged lengths."""
    pkg = package()
    for preset, sr in presets:
        n = int(seconds*sr)
        x = np.stack([synth_input(s, channels, n, sr) for s in range(streams)])
        outs = []
        for teams in (True, False):
            # "2": the team kernels even where a tile has only a few frames per team (on a 256-CU part this case is far below the
            # launcher's threshold of six frames per team, and BOTH runs would take the per-frame kernels)
            monkeypatch.setenv("SMST_FFT_TEAMS", "2" if teams else "0")
            counts = [pkg.launch_count(k, lib) for k in ("analyse_teams", "synth_teams", "analyse_fast")]
            b = pkg.StretchBatch(streams, channels, preset=preset, sample_rate=sr, lib=lib)
            b.setTransposeSemitones(4.0, 0.2, stream=1)
            cut = n*2//3
            n_in = np.array([cut - 17*s for s in range(streams)], np.int32)
            y1 = np.array(b.process(np.ascontiguousarray(x[:, :, :cut]), (n_in*1.4).astype(np.int32), in_samples=n_in), copy=True)
            y2 = np.array(b.process(np.ascontiguousarray(x[:, :, cut:]), int((n - cut)*1.4)), copy=True)
            b.close()
            outs.append(np.concatenate([y1, y2], axis=2))
            grew = [pkg.launch_count(k, lib) - c for k, c in zip(("analyse_teams", "synth_teams", "analyse_fast"), counts)]
            # teams: both team kernels ran, and the per-frame kernel took the frames that reach into the history (mixed tile)
            assert (grew[0] > 0 and grew[1] > 0 and grew[2] > 0) if teams else (grew[0] == 0 and grew[1] == 0 and grew[2] > 0), (teams, grew)
        monkeypatch.delenv("SMST_FFT_TEAMS", raising=False)
        assert np.abs(outs[0]).max() > 0.05
        assert np.array_equal(outs[0], outs[1]), (preset, sr, float(np.abs(outs[0] - outs[1]).max()))


def case_synth_emit_equals_two_kernels(lib, monkeypatch, presets=(("cheaper", 48000), ("default", 48000)), seconds=0.6, streams=3, channels=2,
                                       splits=(False, True), half_state=False):
    """Synthesis + overlap-add + emission in one kernel (kSynthEmitTeams: the overlap-add ring in a team's registers, no frames in
    HBM) against kSynthTeams + kEmit: the same values added in the same order, so output AND carried sums are bit-identical -- over
    three calls with ragged lengths (the second and third start from the carry the first left; the last one is short, so a stream
    ends inside a tile), with and without split computation (the frame lands one interval later), then a flush of what the ring holds."""
    pkg = package()
    for preset, sr in presets:
        n = int(seconds*sr)
        x = np.stack([synth_input(s, channels, n, sr) for s in range(streams)])
        for split in splits:
            outs = []
            for fusedKernel in (True, False):
                monkeypatch.setenv("SMST_FFT_TEAMS", "2")
                monkeypatch.setenv("SMST_SYNTH_EMIT", "2" if fusedKernel else "0")
                names = ("synth_emit", "synth_teams")
                counts = [pkg.launch_count(k, lib) for k in names]
                b = pkg.StretchBatch(streams, channels, preset=preset, sample_rate=sr, split=split, lib=lib, **({"half_state": True} if half_state else {}))
                b.setTransposeSemitones(4.0, 0.2, stream=1)
                c1, c2 = n//2, n*5//6
                n_in = np.array([c1 - 17*s for s in range(streams)], np.int32)
                y1 = np.array(b.process(np.ascontiguousarray(x[:, :, :c1]), (n_in*1.4).astype(np.int32), in_samples=n_in), copy=True)
                n_in2 = np.full(streams, c2 - c1, np.int32)
                n_in2[1 % streams] = 0  # a stream that sits this call out: no hops, no samples -- its carry has to pass through the tile untouched
                y2 = np.array(b.process(np.ascontiguousarray(x[:, :, c1:c2]), (n_in2*0.8).astype(np.int32), in_samples=n_in2), copy=True)
                n_in3 = np.array([max(64, (n - c2) - 301*s) for s in range(streams)], np.int32)
                y3 = np.array(b.process(np.ascontiguousarray(x[:, :, c2:]), (n_in3*1.1).astype(np.int32), in_samples=n_in3), copy=True)
                tail = np.array(b.flush(b.outputLatency() + 100), copy=True)
                b.close()
                outs.append(np.concatenate([y1, y2, y3, tail], axis=2))
                grew = [pkg.launch_count(k, lib) - c for k, c in zip(names, counts)]
                # (split computation: a call's first tile, which begins with the block the call before left in flight, goes through kSynthTeams + kEmit)
                assert (grew[0] > 0 and (grew[1] == 0 or split)) if fusedKernel else (grew[0] == 0 and grew[1] > 0), (fusedKernel, split, grew)
            monkeypatch.delenv("SMST_FFT_TEAMS", raising=False)
            monkeypatch.delenv("SMST_SYNTH_EMIT", raising=False)
            assert np.abs(outs[0]).max() > 0.05
            assert np.array_equal(outs[0], outs[1]), (preset, sr, split, float(np.abs(outs[0] - outs[1]).max()))


def case_continuous_equals_tiled(lib, monkeypatch, geometry=dict(block=1920, interval=480), channel_counts=(2, 1), streams=3, ratios=(1.5, 1.0),
                                 seconds=None):
    """The recurrence as ONE wavefront through all tiles of a call (kVocoderCont: lane r takes hops r, r + 64, ... without draining,
    a launch finishes tile t-1 and begins tile t; SMST_CONTINUOUS=1) against the tile-by-tile kernel (the default): same records, same
    arithmetic, so outputs AND the state a call leaves behind must be bit-identical -- over three calls (the second and third start
    from the state the continuous form handed over; the third is short: one tile, the tile form on both sides), with ragged stream
    lengths (a stream whose last tile is not the call's last), at 1.5x (every hop re-analyses its previous spectrum) and at 1.0x
    (Band.prevInput = the hop before's input, across the tile boundary).  The launch counters prove which form ran."""
    pkg = package()
    geometry = dict(geometry)
    sr = geometry.pop("sr", 48000)
    I = geometry["interval"]
    if "preset" in geometry:
        geometry.pop("interval")
    results = {}
    for C in channel_counts:
        for ratio in ratios:
            hops = [200, 150, 70][:streams] + [130]*max(0, streams - 3)    # output hops of the first call per stream: 4 / 3 / 2 tiles
            n_out1 = np.array([h*I + 17*(i + 1) for i, h in enumerate(hops)], np.int32)
            n_in1 = np.array([int(round(n/ratio)) for n in n_out1], np.int32)
            n_out2 = np.array([100*I + 5, 66*I, 129*I + 1][:streams] + [80*I]*max(0, streams - 3), np.int32)
            n_in2 = np.array([int(round(n/ratio)) for n in n_out2], np.int32)
            n_out3 = np.full(streams, 20*I, np.int32)
            n_in3 = np.array([int(round(n/ratio)) for n in n_out3], np.int32)
            total = int(n_in1.max() + n_in2.max() + n_in3.max())
            x = np.stack([synth_input(s, C, total, sr) for s in range(streams)])
            outs, states = [], []
            for tiled in (False, True):
                if tiled:
                    monkeypatch.delenv("SMST_CONTINUOUS", raising=False)
                else:
                    monkeypatch.setenv("SMST_CONTINUOUS", "1")
                before = pkg.launch_count("vocoder_continuous", lib), pkg.launch_count("vocoder_aligned", lib)
                b = pkg.StretchBatch(streams, C, lib=lib, **(geometry if "block" in geometry else dict(sample_rate=sr, **geometry)))
                assert b.intervalSamples() == I
                y1 = np.array(b.process(np.ascontiguousarray(x[:, :, :n_in1.max()]), n_out1, in_samples=n_in1), copy=True)
                st1 = [np.concatenate([b.debug_state(s, w).ravel() for w in (0, 1, 2, 3)]) for s in range(streams)]
                pos = int(n_in1.max())
                y2 = np.array(b.process(np.ascontiguousarray(x[:, :, pos:pos + n_in2.max()]), n_out2, in_samples=n_in2), copy=True)
                pos += int(n_in2.max())
                y3 = np.array(b.process(np.ascontiguousarray(x[:, :, pos:pos + n_in3.max()]), n_out3, in_samples=n_in3), copy=True)
                st3 = [np.concatenate([b.debug_state(s, w).ravel() for w in (0, 1, 2, 3)]) for s in range(streams)]
                b.close()
                grew = pkg.launch_count("vocoder_continuous", lib) - before[0], pkg.launch_count("vocoder_aligned", lib) - before[1]
                assert (grew[0] == 0 and grew[1] > 0) if tiled else (grew[0] >= 3 + 3), (tiled, grew)  # (the first call's first tile holds the hop after the reset, with random time factors: tile by tile)
                outs.append((y1, y2, y3))
                states.append((st1, st3))
            monkeypatch.delenv("SMST_CONTINUOUS", raising=False)
            assert float(np.abs(outs[0][0]).max()) > 0.05
            for i, (p, q) in enumerate(zip(outs[0], outs[1])):
                assert np.array_equal(p, q), (C, ratio, "call", i, float(np.abs(p - q).max()), np.argwhere(p != q)[:4].tolist())
            for which in range(2):
                for s in range(streams):
                    assert np.array_equal(states[0][which][s], states[1][which][s]), (C, ratio, "state after call", 1 + 2*which, s)
            results["%dch %.2fx" % (C, ratio)] = "bit-identical"
    return results


TOL_FORMANT_STAGE = 1e-4


def case_formant_stages(lib, ref, monkeypatch, cfg, channels=2, stretch=0.75, hops=16, streams=(0, 1, 2), variants=None):
    """A phase-free, per-stage instrument for updateFormants (signalsmith-stretch.h:972-1036): after EVERY hop the
    product's formant envelope (formantMetric after its eight max-decay / min-grow passes, :984-1006), the pitch estimate it was built
    with (:980-981; estimateFrequency :929-966 when no base frequency is set) and the per-bin energy ratio applied to inputEnergy
    (:1018-1033) against oracle/_ref's private members.  These are feed-forward quantities -- functions of the input spectra and
    the parameters, not of the recurrence's state -- so a free-running comparison does not drift and the bound is tight: 1e-4
    (energy-weighted relative RMS of envelope and ratio, relative error of the estimate) or -- where the checker's own stage is less
    well-conditioned than that -- five times its response to an input perturbed by PERTURBATION; every hop, every stream.  The teacher-forced
    Band.output leg under formant processing is ill-conditioned (the checker's own one-hop response reaches 1.75 there); this leg is
    not.  The product's figures come from the separate envelope kernel (SMST_NO_FEED_FUSION=1: smst_batch_debug_get_formants); the
    default form keeps them in LDS and is bit-identical to it (case_feed_fusion_equals_separate)."""
    pkg = package()
    monkeypatch.setenv("SMST_NO_FEED_FUSION", "1")
    sr = int(cfg.get("sample_rate", 48000))
    kw = dict(preset=cfg["preset"], sample_rate=cfg.get("sample_rate", 48000.0)) if cfg.get("preset") in ("default", "cheaper") else \
        dict(block=cfg["block"], interval=cfg["interval"], split=cfg.get("split", False))
    if variants is None:
        variants = {
            "config 4b (base 200 Hz)": lambda o: (o.setTransposeSemitones(4, 8000/48000), o.setFormantFactor(1, True), o.setFormantBase(200/48000)),
            "estimated base": lambda o: (o.setTransposeSemitones(4, 8000/48000), o.setFormantFactor(1, True), o.setFormantBase(0)),
            "formant shift +3 st": lambda o: (o.setFormantSemitones(3, False), o.setFormantBase(150/48000)),
        }
    figures = {}
    for name, setup in variants.items():
        S = len(streams)
        b = pkg.StretchBatch(S, channels, lib=lib, **kw)
        setup(b)
        refs = [make("ref", lib, ref, channels, cfg, setup, seed=i) for i in range(S)]
        twins = [make("ref", lib, ref, channels, cfg, setup, seed=i) for i in range(S)]  # the checker on an input perturbed by PERTURBATION
        I, M = b.intervalSamples(), b.bands()
        n_in = _hop_io(I, stretch, hops)[1] + 8
        xs = np.stack([synth_input(s, channels, n_in, sr) for s in streams])
        xp = np.stack([perturbed(x, 1 + i) for i, x in enumerate(xs)])
        worst = dict(envelope=0.0, ratio=0.0, estimate=0.0, envelope_self=0.0, ratio_self=0.0, ratio_over_self=0.0, compared=0)

        def stage(r):  # (envelope[M], ratio[M] as applied, weights, estimate) of a checker instance after a hop
            metric, est = r.formant_metric()
            e_in = np.abs(np.asarray(r.bands_complex(0), np.complex128)[0])**2
            e_scaled = np.asarray(r.bands_real(3), np.float64)[0]
            ok = e_in > 1e-12*float(max(e_in.max(), 1e-300))
            return np.asarray(metric[:M], np.float64), np.where(ok, e_scaled/np.where(ok, e_in, 1.0), 0.0), np.where(ok, e_in, 0.0), est
        for k in range(hops):
            lo, hi = _hop_io(I, stretch, k)
            b.process(xs[:, :, lo:hi] if hi > lo else np.zeros((S, channels, 1), np.float32), I, in_samples=hi - lo)
            for i, r in enumerate(refs):
                r.process(xs[i][:, lo:hi], I)
                twins[i].process(xp[i][:, lo:hi], I)
                got = b.debug_formants(i)
                assert got is not None, (name, "no formant stage reported", k, i)
                ratio_p, env_p, est_p = got
                env_r, ratio_r, w, est_r = stage(r)
                if float(env_r.max()) < 1e-12:
                    continue  # (the stream's first hops: nothing but the window's leading zeros has been analysed)
                env_t, ratio_t, _, _ = stage(twins[i])
                # the ratio as the checker applied it: inputEnergy = |input|^2 * ratio (:1030-1032); energy weights: it matters where there is energy to scale
                norm_e, norm_r = float(np.sum(env_r**2)), max(float(np.sum(w*ratio_r**2)), 1e-300)
                d_env, s_env = float(np.sqrt(np.sum((env_p - env_r)**2)/norm_e)), float(np.sqrt(np.sum((env_t - env_r)**2)/norm_e))
                d_ratio, s_ratio = float(np.sqrt(np.sum(w*(ratio_p - ratio_r)**2)/norm_r)), float(np.sqrt(np.sum(w*(ratio_t - ratio_r)**2)/norm_r))
                d_est = abs(est_p - est_r)/max(1.0, abs(est_r))
                where = "%s: stream %d, hop %d" % (name, streams[i], k)
                # 1e-4, or five times what the CHECKER's own stage moves by under an input perturbation of PERTURBATION (a chirp's energy sits in a few
                # bins whose target falls on the -80 dB skirt of the same peak, where two FFTs' rounding noise is 1e-3 of the local energy: the checker
                # itself does not hold that ratio to 1e-4 -- measured on the MI355X: 1.5e-4 there against 3e-7 on sine and noise streams)
                assert d_env <= max(TOL_FORMANT_STAGE, SELF_FACTOR*s_env), "%s: formant envelope rel-RMS %.2e (checker's own %.2e)" % (where, d_env, s_env)
                assert d_est <= TOL_FORMANT_STAGE, "%s: pitch estimate %.6f vs %.6f bins" % (where, est_p, est_r)
                assert d_ratio <= max(TOL_FORMANT_STAGE, SELF_FACTOR*s_ratio), "%s: energy ratio (energy-weighted rel-RMS) %.2e (checker's own %.2e)" % (where, d_ratio, s_ratio)
                for key, v in (("envelope", d_env), ("ratio", d_ratio), ("estimate", d_est), ("envelope_self", s_env), ("ratio_self", s_ratio)):
                    worst[key] = max(worst[key], v)
                worst["ratio_over_self"] = max(worst["ratio_over_self"], d_ratio/max(s_ratio, 1e-12) if d_ratio > TOL_FORMANT_STAGE else 0.0)
                worst["compared"] += 1
        b.close()
        assert worst["compared"] >= S*hops//2, (name, worst)
        figures[name] = {k: (float("%.2e" % v) if isinstance(v, float) else v) for k, v in worst.items()}
    return figures


def case_carried_emit_equals_copy(lib, monkeypatch, streams=5, channels=2, splits=(False, True), half_state=False):
    """A call in which no stream fires a hop emits the front of the overlap-add carry and leaves the rest where it is (kEmitCarried: the window's
    beginning moves); SMST_CARRIED_EMIT=0 sends such calls through kEmit, which copies the carry.  Same quotients: output and carry are
    bit-identical -- over short calls (37 .. 61 samples; every third stream idle now and then, one a sample short each time, one silent at first: pass-through), with everything
    that indexes the carry from the front of its rows in between: flush() of some streams (inside a split interval too), outputSeek(), reset(),
    the debug accessors, a clone."""
    pkg = package()
    sr, quanta = 48000, 70
    for split in splits:
        outs, carries = [], []
        for carried in (True, False):
            monkeypatch.setenv("SMST_CARRIED_EMIT", "1" if carried else "0")
            count = pkg.launch_count("emit_carried", lib)
            b = pkg.StretchBatch(streams, channels, block=512, interval=128, split=split, lib=lib, **({"half_state": True} if half_state else {}))
            b.setTransposeSemitones(3.0, 0.2, stream=1)
            x = np.stack([synth_input(s, channels, 61*quanta + 2000, sr) for s in range(streams)])
            x[2, :, :1500] = 0.0   # silent at first: the pass-through streams take nothing from the carry
            parts, pos = [], 0
            parts.append(np.array(b.process(np.ascontiguousarray(x[:, :, :700]), 700), copy=True)); pos = 700
            for k in range(quanta):
                n = np.array([0 if (s % 3 == 0 and k % 7 == 0) else 37 + (3*k) % 25 - (s == 4) for s in range(streams)], np.int32)
                y = np.array(b.process(np.ascontiguousarray(x[:, :, pos:pos + 61]), n, in_samples=n), copy=True)
                pos += 61
                parts.append(y[:, :, :61] if y.shape[2] >= 61 else np.pad(y, ((0, 0), (0, 0), (0, 61 - y.shape[2]))))
                if k == 20:
                    counts = np.array([90 if s % 2 else -1 for s in range(streams)], np.int32)
                    parts.append(np.array(b.flush(counts), copy=True))
                if k == 33:
                    L = b.outputSeekLength(1.0)
                    b.outputSeek(np.ascontiguousarray(x[:, :, pos:pos + L])); pos += L
                if k == 41:
                    sums, prods = b.debug_carry(3)
                    b.debug_set_carry(3, sums, prods)
                if k == 58:
                    b.reset()
            carries.append([np.concatenate([a.ravel() for a in b.debug_carry(s)]) for s in range(streams)])
            parts.append(np.array(b.flush(b.outputLatency() + 60), copy=True))
            b.close()
            one = pkg.SignalsmithStretch(seed=3, lib=lib)   # ... and a copy made while the window sits inside its rows
            one.configure(channels, 512, 128, split)
            parts.append(np.array(one.process(x[0][:, :700], 700), copy=True))
            for k in range(3):
                parts.append(np.array(one.process(x[0][:, 700 + 40*k:740 + 40*k], 40), copy=True))
            twin = one.clone()
            ya, yb = one.process(x[0][:, 820:1200], 380), twin.process(x[0][:, 820:1200], 380)
            assert np.array_equal(ya, yb), "the copy diverges"
            parts.append(np.array(ya, copy=True))
            one.close(); twin.close()
            outs.append(parts)
            grew = pkg.launch_count("emit_carried", lib) - count
            assert (grew > quanta//4) if carried else (grew == 0), (carried, split, grew)
        monkeypatch.delenv("SMST_CARRIED_EMIT", raising=False)
        assert max(float(np.abs(p).max()) for p in outs[0]) > 0.05
        for i, (p, q) in enumerate(zip(outs[0], outs[1])):
            assert np.array_equal(p, q), (split, i, float(np.abs(p - q).max()))
        for s in range(streams):
            assert np.array_equal(carries[0][s], carries[1][s]), (split, s)


def case_clone(lib):
    """smst_clone (the drop-in's copy constructor): the copy continues exactly as the original does, and independently of it."""
    pkg = package()
    C, sr = 2, 48000
    x = synth_input(0, C, 30000, sr) + 0.4*synth_input(2, C, 30000, sr)
    a = pkg.SignalsmithStretch(seed=3, lib=lib)
    a.configure(C, 512, 128, False)
    a.setTransposeSemitones(3, 8000/48000)
    a.setFormantFactor(1.1, True)
    a.process(x[:, :6000], 7500)
    b = a.clone()
    assert (b.blockSamples(), b.intervalSamples(), b.inputLatency()) == (a.blockSamples(), a.intervalSamples(), a.inputLatency())
    ya, yb = a.process(x[:, 6000:9000], 3600), b.process(x[:, 6000:9000], 3600)
    assert np.array_equal(ya, yb), "the copy diverges from the original on the same input"
    yb2 = b.process(x[:, 9000:12000], 3000)          # the copy goes its own way ...
    ya2 = a.process(x[:, 9000:12000], 3600)          # ... without disturbing the original
    c = a.clone()
    assert np.array_equal(a.flush(300), c.flush(300))
    assert yb2.shape[1] == 3000 and ya2.shape[1] == 3600
    unconfigured = pkg.SignalsmithStretch(seed=1, lib=lib).clone()
    unconfigured.presetDefault(1, 48000.0)
    assert unconfigured.blockSamples() == 5760


def case_map_table_lengths(lib):
    """Frequency-map tables of different lengths in one batch: every stream interpolates ITS OWN knots (re-evaluating a
    stored row on a longer table's grid cut the corner at every old knot), so a long table given AFTER a short one keeps its detail
    and each stream equals, bit for bit, a single-stream object given the same table."""
    pkg = package()
    C, sr, n = 1, 48000, 128*60
    x = np.stack([synth_input(s, C, n, sr) for s in (0, 1)])
    short = np.array([1.2*0.125, 1.2*0.375], np.float32)                        # two points (at f = 1/8 and 3/8): f -> 1.2 f
    freqs = (np.arange(1024) + 0.5)/2048
    long_ = (freqs*np.where(freqs < 0.1, 1.5, 1.0) + np.where(freqs < 0.1, 0.0, 0.05)).astype(np.float32)  # a kink at 0.1: lost at 2 points
    b = pkg.StretchBatch(2, C, block=512, interval=128, lib=lib)
    b.setFreqMapTable(short, stream=0)
    b.setFreqMapTable(long_, stream=1)
    y = np.asarray(b.process(x, n))
    for s, table in ((0, short), (1, long_)):
        one = pkg.SignalsmithStretch(lib=lib)
        one.configure(C, 512, 128, False)
        one.setFreqMapTable(table)
        o = one.process(x[s], n)
        assert np.array_equal(y[s], np.asarray(o)), (s, rel_rms(y[s][:, 1500:], o[:, 1500:]))  # its own knots, bit for bit
    b.setFreqMapTable(None)                                                      # all tables gone: the next one starts afresh
    b.setFreqMapTable(short, stream=1)
    b.reset()
    y2 = np.asarray(b.process(x, n))
    one = pkg.SignalsmithStretch(lib=lib)
    one.configure(C, 512, 128, False)
    one.setFreqMapTable(short)
    assert np.array_equal(y2[1], np.asarray(one.process(x[1], n)))


def case_debug_map_is_of_the_last_call(lib):
    """smst_batch_debug_get_map reports the newest hop of the LAST process() call: after a call without a hop it reports none
    (the slot it pointed into may have been reused)."""
    pkg = package()
    C, sr = 1, 48000
    x = synth_input(0, C, 4000, sr)[None]
    b = pkg.StretchBatch(1, C, block=512, interval=128, lib=lib)
    b.setTransposeSemitones(5, 0)
    b.process(x[:, :, :2000], 2000)
    assert b.debug_map(0) is not None
    b.process(x[:, :, 2000:2010], 10)   # 10 output samples: no hop fires
    assert b.debug_map(0) is None
    b.process(x[:, :, 2010:3000], 990)
    assert b.debug_map(0) is not None
    b.reset()
    assert b.debug_map(0) is None


def case_split_mid_interval_flush(lib, ref):
    """Split computation (signalsmith-stretch.h:294-297,321-325,442-455): the reference spreads a block's steps over the interval and
    flush() reads the real output ring, one interval ahead of the stashed copy that process() emits from -- so a flush between two
    interval boundaries returns the newest frame only as far as its synthesis steps have run, resets the real ring, and leaves the
    rest of the interval to the stashed ring.  The product analyses the block in flight when it starts and runs the rest when the
    interval is complete, or as far as a flush() needs it (smst_engine.h: PendingBlock): every offset agrees with the reference,
    in the flushed tail AND in what process() returns afterwards.  (Rounds 1-4 completed the block at the interval's first sample
    and pinned the resulting deviation here: 0.05 .. 1.0 of the tail at interior offsets.)"""
    return _split_mid_interval_flush(lib, ref, 2, SMALL_SPLIT, (0, 1, 5, 32, 54, 55, 64, 100, 115, 116, 121, 122, 127))


def case_split_mid_interval_flush_wide(lib, ref, channels=2, block=768, offsets=(54, 100, 104, 108, 112, 116, 121)):
    """The same where the single-hop kernel does not apply (a vertical step round(fft/interval) outside 2..5 --
    block 768 / 896 / 1024 at interval 128 give 6 / 7 / 8 -- or SMST_NO_SINGLE_HOP): the block in flight then runs through the
    wavefront kernels, whose records honour HopDesc.startBin (computeRecord) -- until round 6 only kVocoderOne did, and a flush
    between two chunks of the main prediction (offsets 100..116 here) left 0.47 of the next process() wrong."""
    # (a lead-in of 16 hops instead of 55 and bounds ten times wider: at these block sizes the free-running recurrence has drifted by
    # 6e-4 after 55 hops on the MI355X -- the checker's own sensitivity, nothing to do with the flush; the defect this case pins was 0.47)
    return _split_mid_interval_flush(lib, ref, channels, dict(preset="configure", block=block, interval=128, split=True), offsets, long_flush=False,
                                     lead_hops=16, tol_process=1e-4, tol=1e-3)


def _split_mid_interval_flush(lib, ref, C, cfg, offsets, long_flush=True, lead_hops=55, tol_process=1e-5, tol=1e-4):
    sr = 48000
    SMALL_SPLIT = cfg
    x = synth_input(0, C, 12000, sr)
    I = 128
    figures = {}
    for offset in offsets:
        g, r = make("product", lib, ref, C, SMALL_SPLIT), make("ref", lib, ref, C, SMALL_SPLIT)
        nout = lead_hops*I + offset
        nin = 6000 if lead_hops == 55 else int(nout/1.19)  # (the same stretch ratio with a shorter lead-in)
        a, b = g.process(x[:, :nin], nout), r.process(x[:, :nin], nout)
        assert rel_rms(a, b) < tol_process, (offset, "process", rel_rms(a, b))
        fa, fb = g.flush(I), r.flush(I)
        pa, pb = g.process(x[:, 6000:6700], 700), r.process(x[:, 6000:6700], 700)
        level = float(np.sqrt(np.mean(np.square(b, dtype=np.float64))))
        figures[offset] = (rel_rms(fa, fb), float(np.sqrt(np.mean(np.square(np.asarray(pa, np.float64) - pb))))/level)
    assert all(f < tol and after < tol for f, after in figures.values()), figures
    if not long_flush:
        return {k: ("%.1e" % v[0], "%.1e" % v[1]) for k, v in figures.items()}
    # a flush longer than one interval finishes the block first (it runs process() on silence, :439-440), from any offset
    g, r = make("product", lib, ref, C, SMALL_SPLIT), make("ref", lib, ref, C, SMALL_SPLIT)
    g.process(x[:, :6000], 55*I + 40), r.process(x[:, :6000], 55*I + 40)
    fa, fb = g.flush(300), r.flush(300)
    figures["long flush"] = (rel_rms(fa[:, :I], fb[:, :I]), rel_rms(fa, fb))
    assert figures["long flush"][0] < 1e-4 and figures["long flush"][1] < 1e-3, figures  # (its later intervals ran blocks at a huge time factor: their own sensitivity)
    return {k: ("%.1e" % v[0], "%.1e" % v[1]) for k, v in figures.items()}


def case_split_events_golden(lib, ref, geometry):
    """The product against what the reference's shipped WASM build does when a flush / parameter change / reset / seek falls BETWEEN
    two interval boundaries in split-computation mode (tests/golden/split_events, make_golden.py split): every offset of the
    fixtures, among them the single samples at which a step of the block in flight moves to the other side of the event."""
    x, cfg, scen = scenarios.load_split_events(geometry)
    worst = {}
    for name, ops, y in scen:
        g = make("product", lib, ref, x.shape[0], cfg)
        out = np.asarray(scenarios.replay(g, x, ops))
        assert out.shape == y.shape
        e = scenarios.split_event_errors(out, y, ops, g.intervalSamples())
        tol_seg, tol_after = scenarios.split_event_tolerance(name)
        if max(v for _, v in e["segments"]) > tol_seg or max(e["after"]) > tol_after:
            # a transposed scenario drifts from the WASM within a few hops (so does the checker): the checker's own distance decides
            r = make("ref", lib, ref, x.shape[0], cfg)
            er = scenarios.split_event_errors(scenarios.replay(r, x, ops), y, ops, r.intervalSamples())
            assert all(v <= max(tol_seg, 2*w + FLOOR) for (_, v), (_, w) in zip(e["segments"], er["segments"])), (geometry, name, e, er)
            assert all(v <= max(tol_after, 2*w + 1e-5) for v, w in zip(e["after"], er["after"])), (geometry, name, e, er)
        kind = name.rsplit("_", 1)[0]
        worst[kind] = max(worst.get(kind, 0.0), max(e["after"]))
    return {k: "%.1e" % v for k, v in worst.items()}


def case_split_freq_map_mid_interval(lib, ref, channels=2, cfg=SMALL_SPLIT, offsets=(3, 40, 46, 47, 52, 56, 57, 61, 70, 100)):
    """setFreqMap between two interval boundaries in split-computation mode (signalsmith-stretch.h:120-122 replaces the std::function as a
    whole): findPeaks (:874) of the block in flight evaluates the OLD map if it ran before the call, updateFormants step 2 (:1020, with
    formant compensation) the NEW one if it runs after it.  The product keeps up to three table rows per stream and a step's latched
    parameters name the row it saw (StreamParams.mapSlot) -- until round 6 the knots were shared, and a findPeaks that had run with
    table A was re-run at the interval's end on table B's knots.  The WASM ABI has no setFreqMap: against oracle/_ref, whose step
    partition the split_events fixtures pin.  A second change inside the same interval (a third table) exercises the third row."""
    sr = 48000
    I = make("ref", lib, ref, channels, cfg).intervalSamples()
    x = synth_input(0, channels, 9000, sr) + 0.3*synth_input(3, channels, 9000, sr)
    n = 64
    grid = (np.arange(n) + 0.5)/(2*n)
    table_a = (grid*1.26).astype(np.float32)                                   # +4 semitones
    table_b = np.where(grid < 0.1, grid*0.84, 0.084 + (grid - 0.1)).astype(np.float32)  # -3 semitones below 0.1, then parallel
    table_c = (grid*1.06 + 0.002).astype(np.float32)
    figures = {}
    for off in offsets:
        def play(o, xx=x, off=off):
            o.setFormantFactor(1.1, True)
            o.setFreqMapTable(table_a)
            outs = [o.process(xx[:, :1500], 8*I + off)]
            o.setFreqMapTable(table_b)
            outs.append(o.process(xx[:, 1500:1510], 6))
            o.setFreqMapTable(table_c)
            outs.append(o.process(xx[:, 1510:1510 + 3*I], 3*I))
            return np.concatenate(outs, axis=1)
        g, r = make("product", lib, ref, channels, cfg), make("ref", lib, ref, channels, cfg)
        y, o = np.asarray(play(g)), play(r)
        o2 = [play(make("ref", lib, ref, channels, cfg), xx=perturbed(x, k)) for k in SELF_SEEDS]
        assert_parity(y, o, o2, I, "freq map at %d" % off, cap=CAP_FORMANT)
        figures[off] = rel_rms(y, o)
    return {k: "%.1e" % v for k, v in figures.items()}


def case_split_events_vs_checker(lib, ref, channels=3, cfg=SMALL_SPLIT):
    """What the WASM ABI cannot do (formant parameters: its updateFormants is another revision) and what the
    fixtures do not hold (3 channels, several events inside ONE interval, a flush longer than the interval from an interior offset,
    ragged call sizes) -- against oracle/_ref, whose step partition the fixtures pin.  Formant steps of a mapped 3-channel block:
    29 + 3 steps, updateFormants(0) reads formantBaseFreq (:982), updateFormants(2) the multiplier and the compensation flag (:1020)."""
    sr = 48000
    I = make("ref", lib, ref, channels, cfg).intervalSamples()
    q = I/128.0  # the sample counts below are written for the small geometry (interval 128) and scale with the interval
    x = synth_input(0, channels, int(9000*q), sr) + 0.3*synth_input(3, channels, int(9000*q), sr)
    figures = {}

    def run(label, play, cap=CAP_FORMANT, tol=None):
        g, r = make("product", lib, ref, channels, cfg), make("ref", lib, ref, channels, cfg)
        y, o = np.asarray(play(g)), play(r)
        o2 = [play(make("ref", lib, ref, channels, cfg), xx=perturbed(x, k)) for k in SELF_SEEDS]
        assert_parity(y, o, o2, I, label, cap=cap)
        figures[label] = rel_rms(y, o)
        if tol is not None:
            assert figures[label] < tol, (label, figures[label])

    for off in (3, 40, 47, 52, 53, 56, 57, 61, 70, 100, 127):  # 52|53: updateFormants(0) = step 12 of 31 runs with sample 53; 56|57 (+ 4): updateFormants(2) = step 14 with sample 61
        def formant_change(o, xx=x, off=off):
            o.setTransposeSemitones(3, 0)
            o.setFormantFactor(1.1, True)
            n0, off2 = int(1500*q), int(off*q)
            outs = [o.process(xx[:, :n0], 8*I + off2)]
            o.setFormantBase(180.0/sr)       # updateFormants(0) of the block in flight sees it only if it has not run yet
            outs.append(o.process(xx[:, n0:n0 + 10], 4))
            o.setFormantFactor(0.9, False)   # ... updateFormants(2) likewise, four samples later
            outs.append(o.process(xx[:, n0 + 10:n0 + 10 + 3*I], 3*I))
            return np.concatenate(outs, axis=1)
        run("formant parameters at %d" % off, formant_change)

    def two_events(o, xx=x):  # a parameter change, a short flush and another parameter change inside one interval; then a flush of 2.5 intervals from an interior offset
        o.setTransposeSemitones(-2, 0)
        a, b, c, e = int(2000*q), int(2030*q), int(2800*q), int(4000*q)
        outs = [o.process(xx[:, :a], 10*I + int(20*q))]
        o.setTransposeSemitones(2, 0)
        outs.append(o.process(xx[:, a:b], b - a))
        outs.append(o.flush(int(25*q)))
        o.setTransposeSemitones(5, 0)
        outs.append(o.process(xx[:, b:c], 6*I + int(77*q)))
        outs.append(o.flush(int(2.5*I)))
        outs.append(o.process(xx[:, c:e], e - c))
        return np.concatenate(outs, axis=1)
    run("two events in one interval", two_events, cap=CAP_TONAL)

    def flush_twice(o, xx=x):  # two flushes inside the same interval: the second finds a block that the first one already interrupted
        a, b, c = int(2000*q), int(2030*q), int(3200*q)
        outs = [o.process(xx[:, :a], 12*I + int(60*q))]
        outs.append(o.flush(int(10*q)))
        outs.append(o.process(xx[:, a:b], b - a))
        outs.append(o.flush(int(40*q)))
        outs.append(o.process(xx[:, b:c], c - b))
        return np.concatenate(outs, axis=1)
    run("two flushes in one interval", flush_twice, cap=CAP_TONAL, tol=1e-4 if q == 1 else None)

    def quanta(o, xx=x):  # the real-time pattern with a parameter automation: one setter per 37-sample call
        outs = []
        n = int(37*q)
        for k in range(60):
            o.setTransposeSemitones(-3 + 0.1*k, 0)
            outs.append(o.process(xx[:, n*k:n*(k + 1)], n))
        return np.concatenate(outs, axis=1)
    run("a setter in every 37-sample call", quanta, cap=CAP_TONAL)
    return {k: "%.1e" % v for k, v in figures.items()}


def case_split_dropped_block_random_engine(lib, ref, cfg=SMALL_SPLIT):
    """Beyond 2x a block draws its 2M - 2 time factors from the stream's random engine INSIDE the chunks of its main prediction (stretch.h:749,
    :769), in bin order, when those chunks run.  In split mode a reset() between two interval boundaries drops the block in flight
    (blockProcess = {}, :58): the engine has then moved on by the draws of the chunks that had run -- none early in the interval, all of
    them late -- and every later random factor depends on that.  (Found by the API fuzz, walk 218: the product had advanced the engine by
    the whole block when the block STARTED.)  Offsets before, inside and behind the main prediction of a 20-step stereo block."""
    C, sr = 2, 48000
    x = synth_input(0, C, 9000, sr) + 0.3*synth_input(3, C, 9000, sr)
    I = cfg["interval"]
    figures = {}
    for off in (3, 40, 54, 55, 70, 90, 108, 109, 120, 127):
        def play(o, off=off):
            n1 = int((9*I + off)/2.6)
            outs = [o.process(x[:, :n1], 9*I + off)]           # 2.6x: every hop draws
            o.reset()
            outs.append(o.process(x[:, n1:n1 + 800], 2100))    # ... and so does every hop after the reset
            return np.concatenate([np.asarray(v) for v in outs], axis=1)
        g, r = make("product", lib, ref, C, cfg, seed=5), make("ref", lib, ref, C, cfg, seed=5)
        y, o = play(g), play(r)
        tail = slice(9*I + off, None)
        figures[off] = rel_rms(y[:, tail], o[:, tail])
        assert figures[off] < 1e-4, ("reset with a randomised block in flight at offset %d" % off, figures)
    return {k: "%.1e" % v for k, v in figures.items()}


def case_split_batch_events(lib, monkeypatch=None, streams=5, channels=2, cfg=SMALL_SPLIT):
    """Split computation in a BATCH: every stream sits at another offset inside its interval (ragged per-stream sample counts), parameter
    changes hit single streams, a flush() takes some streams (negative counts leave the others alone) -- and every stream of the batch
    stays bit-identical to the same stream driven alone through the single-stream handle; then a clone taken between two interval
    boundaries (the block in flight and its spectra travel with it) continues exactly as the original."""
    pkg = package()
    sr = 48000
    kw = dict(block=cfg["block"], interval=cfg["interval"], split=True)
    I = cfg["interval"]
    n_total = 40*I
    xs = np.stack([synth_input(s, channels, n_total, sr) + 0.2*synth_input(s + 3, channels, n_total, sr) for s in range(streams)])
    rng = np.random.default_rng(7)
    # a script of calls: per stream (n_in, n_out) for process, or a flush length (negative: not part of it), or a setter
    script = []
    for call in range(9):
        script.append(("process", [(int(rng.integers(40, 400)), int(rng.integers(40, 500))) for _ in range(streams)]))
        if call == 2:
            script.append(("transpose", 1, 5.0))
        if call == 4:
            script.append(("flush", [int(rng.integers(1, 200)) if s % 2 == 0 else -1 for s in range(streams)]))
        if call == 5:
            script.append(("formant", 3, 1.15))
            script.append(("transpose", 0, -3.0))
        if call == 7:
            script.append(("flush", [int(rng.integers(100, 300)) if s % 2 == 1 else -1 for s in range(streams)]))

    def run_batch():
        b = pkg.StretchBatch(streams, channels, lib=lib, **kw)
        pos = [0]*streams
        outs = [[] for _ in range(streams)]
        for step in script:
            if step[0] == "process":
                nin = np.array([c[0] for c in step[1]], np.int32)
                nout = np.array([c[1] for c in step[1]], np.int32)
                x = np.zeros((streams, channels, int(nin.max())), np.float32)
                for s in range(streams):
                    x[s, :, :nin[s]] = xs[s][:, pos[s]:pos[s] + nin[s]]
                    pos[s] += int(nin[s])
                y = np.array(b.process(x, nout, in_samples=nin), copy=True)
                for s in range(streams):
                    outs[s].append(y[s][:, :nout[s]])
            elif step[0] == "flush":
                y = np.array(b.flush(np.array(step[1], np.int32)), copy=True)
                for s in range(streams):
                    if step[1][s] >= 0:
                        outs[s].append(y[s][:, :step[1][s]])
            elif step[0] == "transpose":
                b.setTransposeSemitones(step[2], 0.0, stream=step[1])
            else:
                b.setFormantFactor(step[2], True, stream=step[1])
        b.close()
        return [np.concatenate(o, axis=1) for o in outs]

    def run_single(s):
        g = pkg.SignalsmithStretch(seed=s, lib=lib)  # stream s of a batch carries the engine seeded seed + s
        g.configure(channels, cfg["block"], cfg["interval"], True)
        pos, outs = 0, []
        for step in script:
            if step[0] == "process":
                nin, nout = step[1][s]
                outs.append(np.asarray(g.process(xs[s][:, pos:pos + nin], nout)))
                pos += nin
            elif step[0] == "flush":
                if step[1][s] >= 0:
                    outs.append(np.asarray(g.flush(step[1][s])))
            elif step[0] == "transpose":
                if step[1] == s:
                    g.setTransposeSemitones(step[2], 0.0)
            elif step[1] == s:
                g.setFormantFactor(step[2], True)
        return np.concatenate(outs, axis=1)

    batch = run_batch()
    for s in range(streams):
        single = run_single(s)
        assert batch[s].shape == single.shape and np.abs(single).max() > 0.01
        assert np.array_equal(batch[s], single), ("stream %d of the batch differs from the same stream alone" % s, float(np.abs(batch[s] - single).max()))
    if monkeypatch is not None:  # two streams per sub-batch: the blocks in flight of a call run per sub-batch
        monkeypatch.setenv("SMST_SUB_STREAMS", "2")
        again = run_batch()
        monkeypatch.delenv("SMST_SUB_STREAMS", raising=False)
        assert all(np.array_equal(a, b) for a, b in zip(again, batch))
    # clone between two interval boundaries
    g = pkg.SignalsmithStretch(seed=0, lib=lib)
    g.configure(channels, cfg["block"], cfg["interval"], True)
    g.setTransposeSemitones(4.0, 0.0)
    g.process(xs[0][:, :1500], 11*I + 37)
    twin = g.clone()
    a = np.concatenate([np.asarray(g.process(xs[0][:, 1500:1600], 50)), np.asarray(g.flush(60)), np.asarray(g.process(xs[0][:, 1600:2600], 1000))], axis=1)
    b = np.concatenate([np.asarray(twin.process(xs[0][:, 1500:1600], 50)), np.asarray(twin.flush(60)), np.asarray(twin.process(xs[0][:, 1600:2600], 1000))], axis=1)
    assert np.array_equal(a, b) and np.abs(a).max() > 0.01


def case_across_equals_single_hop(lib, monkeypatch, streams=21, channel_counts=(1, 2), setup=None):
    """Single-hop tiles, mono / stereo: the recurrence runs with its lanes across STREAMS (kVocoder ACROSS); SMST_NO_ACROSS=1 runs
    one chain per stream (kVocoderOne).  Same records, same arithmetic: bit-identical, for a stream count that is no multiple of
    the rows per workgroup and with streams that take no hop in a call (ragged sample counts)."""
    pkg = package()
    geometry = dict(block=512, interval=128, split=False)
    for C in channel_counts:
        outs = []
        for no_across in (False, True):
            if no_across:
                monkeypatch.setenv("SMST_NO_ACROSS", "1")
            else:
                monkeypatch.delenv("SMST_NO_ACROSS", raising=False)
            b = pkg.StretchBatch(streams, C, lib=lib, **geometry)
            if setup:
                setup(b)
            xs = np.stack([synth_input(s, C, 128*24, 48000) for s in range(streams)])
            parts = []
            for k in range(22):
                nin = np.array([128 if (s + k) % 5 else 0 for s in range(streams)], np.int32)   # every fifth stream idles in a call
                nout = np.array([128 if (s + k) % 5 else 0 for s in range(streams)], np.int32)
                y = np.array(b.process(xs[:, :, 128*k:128*(k + 1)], nout, in_samples=nin), copy=True)
                parts.append(y)
            b.close()
            outs.append(np.concatenate(parts, axis=2))
        monkeypatch.delenv("SMST_NO_ACROSS", raising=False)
        assert np.abs(outs[0]).max() > 0.05
        assert np.array_equal(outs[0], outs[1]), (C, float(np.abs(outs[0] - outs[1]).max()))


def case_reconfigure_keeps_random_engine(lib, ref, seed=11):
    """configure() does not reseed the reference's random engine (it is seeded in the constructor only, :38-39, and advanced by the
    draws of randomised hops, :749/:769): configure -> 2.5x -> configure again -> 2.5x must draw the SAME factors as the checker on
    the second run too (the product rebuilt its batch from the seed, rewinding the engine)."""
    C, sr = 2, 48000
    x = synth_input(1, C, 6000, sr) + 0.3*synth_input(5, C, 6000, sr)

    def play(o):
        outs = []
        for _ in range(2):
            o.configure(C, 512, 128, False)
            outs.append(o.process(x[:, :1200], 3000))  # 2.5x: every hop draws 2M - 2 time factors
        return np.concatenate(outs, axis=1)
    g, r = package().SignalsmithStretch(seed=seed, lib=lib), ref.RefStretch(seed)
    y, o = np.asarray(play(g)), play(r)
    o2 = []
    for k in SELF_SEEDS:
        t = ref.RefStretch(seed)
        outs = []
        for _ in range(2):
            t.configure(C, 512, 128, False)
            outs.append(t.process(perturbed(x, k)[:, :1200], 3000))
        o2.append(np.concatenate(outs, axis=1))
    assert_parity(y[:, 3000:], o[:, 3000:], [v[:, 3000:] for v in o2], 128, "second run after a second configure()")
    # and the second run is NOT a replay of the first (it would be if the engine had been rewound)
    assert rel_rms(y[:, 3000:], y[:, :3000]) > 1e-3
