"""Pins the oracle: oracle/_ref (the UNMODIFIED reference header + the L1 restatement in oracle/linear_shim) against
golden vectors produced by the reference's own shipped WASM build (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import rel_rms
import scenarios


@pytest.mark.parametrize("name", scenarios.golden_names())
def test_ref_matches_wasm_golden(ref, name):
    x, y, ops, cfg, info = scenarios.load_golden(name)
    obj = ref.RefStretch()
    scenarios.configure(obj, x.shape[0], cfg)
    assert obj.blockSamples() == info["block"] and obj.intervalSamples() == info["interval"]
    assert obj.inputLatency() == info["inputLatency"] and obj.outputLatency() == info["outputLatency"]
    out = scenarios.replay(obj, x, ops)
    assert out.shape == y.shape
    err = rel_rms(out, y)
    assert err <= scenarios.GOLDEN_TOL[name], (name, err)


def test_identity_known_answer(ref):
    """1.0x / 0 semitones is the input delayed by inputLatency+outputLatency."""
    x, y, ops, cfg, info = scenarios.load_golden("identity_mono_44k")
    lag = info["inputLatency"] + info["outputLatency"]
    assert rel_rms(y[:, lag:], x[:, :-lag]) < 1e-6  # the WASM itself
    obj = ref.RefStretch()
    scenarios.configure(obj, 1, cfg)
    out = scenarios.replay(obj, x, ops)
    assert rel_rms(out[:, lag:], x[:, :-lag]) < 1e-6


def test_window_perfect_reconstruction(ref):
    """sum_m w[j + m*interval]^2 == 1 (SURVEY.md 8c KAT 5) and the published window values (App. A.3)."""
    obj = ref.RefStretch()
    obj.presetDefault(1, 48000.0)
    w = obj.window().astype(np.float64)
    B, I = obj.blockSamples(), obj.intervalSamples()
    assert B == 5760 and I == 1440 and obj.fftSamples() == 6144
    acc = (w.reshape(B//I, I)**2).sum(axis=0)
    assert np.abs(acc - 1).max() < 1e-6
    assert abs(w[0] - 0.0154853) < 1e-6 and abs(w[B//2] - 0.8161286) < 1e-6 and abs(w.sum() - 2400.955) < 2e-2


def test_fft_sizes(ref):
    """fftSamples per preset (probe-verified on the WASM's twiddle tables)."""
    for preset, sr, n in (("default", 48000, 6144), ("default", 44100, 6144), ("cheaper", 96000, 10240), ("cheaper", 48000, 5120)):
        obj = ref.RefStretch()
        (obj.presetDefault if preset == "default" else obj.presetCheaper)(1, float(sr))
        assert obj.fftSamples() == n


def test_modified_spectrum_definition(ref):
    """X[k] = sum_n x[n] w[n] exp(-2 pi i (k+1/2) n / N), centre origin, unnormalised vs a direct DFT."""
    obj = ref.RefStretch()
    obj.configure(1, 120, 30)
    B, N = 120, obj.fftSamples()
    rng = np.random.default_rng(3)
    block = rng.standard_normal(B).astype(np.float32)
    X = obj.analyse_block(block)
    w = obj.window().astype(np.float64)
    n = np.arange(B) - B//2
    k = np.arange(N//2)
    direct = (block*w)[None, :] @ np.exp(-2j*np.pi*np.outer(n, k + 0.5)/N)
    assert np.abs(X - direct[0]).max() < 2e-5*np.abs(direct).max()


@pytest.mark.parametrize("name", scenarios.golden_names())
def test_port_matches_wasm_golden_and_ref(ref, name):
    """oracle/stretch_port.cpp (the plain C++ restatement) against the WASM golden vectors and, sample for sample,
    against oracle/_ref (the unmodified reference header): same L1, same order of operations -> tight bound."""
    import port_oracle
    x, y, ops, cfg, info = scenarios.load_golden(name)
    p, r = port_oracle.PortStretch(), ref.RefStretch()
    scenarios.configure(p, x.shape[0], cfg)
    scenarios.configure(r, x.shape[0], cfg)
    assert (p.blockSamples(), p.intervalSamples(), p.inputLatency(), p.outputLatency()) == \
        (info["block"], info["interval"], info["inputLatency"], info["outputLatency"])
    a, b = scenarios.replay(p, x, ops), scenarios.replay(r, x, ops)
    assert rel_rms(a, y) <= scenarios.GOLDEN_TOL[name]
    assert rel_rms(a, b) <= 1e-5, rel_rms(a, b)


def test_port_formants_match_ref(ref):
    import port_oracle
    from conftest import synth_input
    x = synth_input(0, 2, 20000, 48000) + 0.5*synth_input(4, 2, 20000, 48000)
    for setup in (lambda o: (o.setTransposeSemitones(4, 8000/48000), o.setFormantFactor(1, True), o.setFormantBase(200/48000)),
                  lambda o: (o.setFormantSemitones(3, False), o.setFormantBase(0))):
        p, r = port_oracle.PortStretch(), ref.RefStretch()
        p.presetDefault(2, 48000.0)
        r.presetDefault(2, 48000.0)
        setup(p)
        setup(r)
        assert rel_rms(p.process(x, 15000), r.process(x, 15000)) <= 1e-5
