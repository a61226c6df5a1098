"""Parity tests proper: the hand-written gfx950 kernels, called through the C ABI, against the checker (oracle/_ref =
the unmodified reference header) and the WASM golden vectors.  Run on a real MI355X: pytest -m gpu."""
import numpy as np
import pytest

from conftest import package, rel_rms, synth_input
import parity_cases as pc
import scenarios

pytestmark = pytest.mark.gpu

D48 = dict(preset="default", sample_rate=48000.0)


@pytest.mark.parametrize("name", scenarios.golden_names())
def test_golden(hip, ref, name):
    pc.case_golden(hip, ref, name)


def test_api_surface_small(hip, ref):
    pc.case_api_surface(hip, ref)


def test_split_mode(hip, ref):
    pc.case_split_mode(hip, ref)


def test_pitch_and_formants_small(hip, ref):
    pc.case_pitch_and_formants(hip, ref)


def test_pitch_and_formants_preset_default(hip, ref):
    pc.case_pitch_and_formants(hip, ref, cfg=D48, n=20000)


def test_silence(hip, ref):
    pc.case_silence(hip, ref)


def test_channels(hip, ref):
    pc.case_channels(hip, ref, channel_counts=(1, 2, 3, 4, 5, 6, 7, 8))


def test_batch_ragged(hip, ref):
    pc.case_batch_ragged(hip, ref)
    pc.case_batch_ragged(hip, ref, cfg=dict(preset="configure", block=5760, interval=1440, split=False), S=6, n=30000)


def test_random_time_factor(hip, ref):
    pc.case_random_time_factor(hip, ref)


def _ref_run(ref, cfg, C, x, nout, setup=None):
    r = ref.RefStretch()
    scenarios.configure(r, C, cfg)
    if setup:
        setup(r)
    return r.process(x, nout)


def _check_streams(y, refs, I, label):
    """Horizon-aware comparison: first 16 hops rel-RMS <= 1e-3 for every signal type; the whole
    render <= 5e-3 for tonal streams; noise streams (s % 3 == 2) decorrelate, so compare the output energy (1 %)."""
    for s, o in enumerate(refs):
        n = o.shape[1]
        head = min(n, 16*I)
        assert rel_rms(y[s][:, :head], o[:, :head]) < pc.TOL_SHORT, (label, s, "short horizon")
        if s % 3 != 2:
            assert rel_rms(y[s][:, :n], o) < pc.TOL_LONG, (label, s, "long horizon", rel_rms(y[s][:, :n], o))
        else:
            ra, rb = np.sqrt(np.mean(y[s][:, head:n]**2)), np.sqrt(np.mean(o[:, head:]**2))
            assert abs(ra/rb - 1) < 0.01, (label, s, "noise energy", ra, rb)


def test_config2_subset(hip, ref):
    """BASELINE config 2 (256 stereo streams, 48 kHz, presetDefault, 1.5x): parity subset = first 8 streams, 2 s."""
    pkg = package()
    S, C, sr, n = 8, 2, 48000, 96000
    xs = np.stack([synth_input(s, C, n, sr) for s in range(S)])
    b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr, lib=hip)
    y = b.process(xs, int(n*1.5))
    refs = [_ref_run(ref, D48, C, xs[s], int(n*1.5)) for s in range(S)]
    _check_streams(y, refs, 1440, "config2")
    b.close()


def test_config3_subset(hip, ref):
    """config 3: +12 semitones with 8 kHz tonality limit, stretch 1.0."""
    pkg = package()
    S, C, sr, n = 6, 2, 48000, 72000
    xs = np.stack([synth_input(s, C, n, sr) for s in range(S)])
    b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr, lib=hip)
    b.setTransposeSemitones(12, 8000/48000)
    y = b.process(xs, n)
    refs = [_ref_run(ref, D48, C, xs[s], n, lambda r: r.setTransposeSemitones(12, 8000/48000)) for s in range(S)]
    _check_streams(y, refs, 1440, "config3")
    b.close()


def test_config4_subset(hip, ref):
    """config 4 literal (0.75x, formant compensation inert) and 4b (+4 st so the formant kernel runs)."""
    pkg = package()
    S, C, sr, n = 6, 2, 48000, 72000
    xs = np.stack([synth_input(s, C, n, sr) for s in range(S)])
    for semis in (0.0, 4.0):
        def setup(o, semis=semis):
            if semis:
                o.setTransposeSemitones(semis, 8000/48000)
            o.setFormantFactor(1, True)
            o.setFormantBase(200/48000)
        b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr, lib=hip)
        setup(b)
        y = b.process(xs, int(n*0.75))
        refs = [_ref_run(ref, D48, C, xs[s], int(n*0.75), setup) for s in range(S)]
        # with formants the long-horizon bound is 5e-2 (SURVEY App. D.2 iii): check short horizon + energy here
        for s, o in enumerate(refs):
            assert rel_rms(y[s][:, :16*1440], o[:, :16*1440]) < pc.TOL_SHORT, ("config4", semis, s)
            assert rel_rms(y[s][:, :o.shape[1]], o) < (5e-2 if s % 3 != 2 else 2.0), ("config4 long", semis, s)
        b.close()


def test_config5_subset(hip, ref):
    """config 5 flavour: 8-channel streams, 96 kHz, presetCheaper (split), per-stream random stretch and transpose."""
    pkg = package()
    S, C, sr, n = 4, 8, 96000, 96000
    cfg = dict(preset="cheaper", sample_rate=float(sr))
    xs = np.stack([synth_input(s, C, n, sr) for s in range(S)])
    g = np.random.Generator(np.random.PCG64(5))
    stretch = g.uniform(0.75, 1.5, S)
    semis = g.uniform(-12, 12, S)
    nout = [int(round(n*stretch[s])) for s in range(S)]
    b = pkg.StretchBatch(S, C, preset="cheaper", sample_rate=sr, lib=hip)
    for s in range(S):
        b.setTransposeSemitones(float(semis[s]), 0.0, stream=s)
    y = b.process(xs, nout)
    for s in range(S):
        o = _ref_run(ref, cfg, C, xs[s], nout[s], lambda r, s=s: r.setTransposeSemitones(float(semis[s]), 0.0))
        assert rel_rms(y[s][:, :10*3840], o[:, :10*3840]) < pc.TOL_SHORT, ("config5", s)
    b.close()


def test_full_batch_identity_and_determinism(hip):
    """Size-independent known answers at BASELINE config-2 scale (256 stereo streams, presetDefault @ 48 kHz):
    1.0x / 0 st reproduces every input delayed by inputLatency+outputLatency; the same call twice is bit-identical;
    hop-aligned chunking is bit-identical to one call; a stream inside a batch equals the same stream run alone."""
    import torch
    pkg = package()
    S, C, sr, n = 256, 2, 48000, 48000
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(7)
    base = torch.from_numpy(np.stack([synth_input(s, C, n, sr) for s in range(12)]))
    x = base.repeat(22, 1, 1)[:S].contiguous()
    x = (x*(0.5 + 0.5*torch.rand(S, 1, 1, generator=gen))).to(dev)
    b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr, lib=hip)
    lag = b.inputLatency() + b.outputLatency()
    y = b.process(x, n)
    b.synchronize()
    err = torch.sqrt(((y[:, :, lag:] - x[:, :, :-lag])**2).mean(dim=(1, 2))/(x[:, :, :-lag]**2).mean(dim=(1, 2)))
    assert float(err.max()) < 1e-6, float(err.max())
    # determinism
    b.reset()
    y2 = b.process(x, n)
    b.synchronize()
    assert torch.equal(y, y2)
    # chunking invariance (hop-aligned chunks), 1.25x so the phase vocoder is active
    b.reset()
    whole = b.process(x[:, :, :28800], 36000)
    b.synchronize()
    b.reset()
    parts = []
    for k in range(5):
        parts.append(b.process(x[:, :, 5760*k:5760*(k + 1)].contiguous(), 7200))
    b.synchronize()
    chunked = torch.cat(parts, dim=2)
    assert float((whole - chunked).abs().max()) <= 1e-6*float(whole.abs().max())
    b.close()
    # batch == single
    one = pkg.StretchBatch(1, C, preset="default", sample_rate=sr, lib=hip)
    for s in (0, 77, 255):
        one.reset()
        ys = one.process(x[s:s + 1, :, :28800].contiguous(), 36000)
        one.synchronize()
        assert torch.equal(ys[0], whole[s]), s
    one.close()


def test_host_and_device_memory_agree(hip):
    import torch
    pkg = package()
    S, C, sr, n = 3, 2, 48000, 20000
    xs = np.stack([synth_input(s, C, n, sr) for s in range(S)])
    b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr, lib=hip)
    yh = b.process(xs, 25000)
    b.reset()
    yd = b.process(torch.from_numpy(xs).cuda(), 25000)
    b.synchronize()
    assert np.array_equal(yh, yd.cpu().numpy())
    b.close()
