"""Parity cases shared by the CPU-emulated run (host logic, `-m "not gpu"`) and the real gfx950 run (`-m gpu`).
Every case drives the product through its C ABI (python ctypes mirror of the reference API) and the checker
(oracle/_ref: the unmodified reference header) with the same seeded inputs, and compares.

Tolerances (fp32 path, chaotic recurrence -- SURVEY.md App. D): rel-RMS <= TOL_SHORT over short horizons
(<= 16 hops), TOL_LONG over long tonal horizons; identity and pure-bookkeeping cases <= TOL_EXACT."""
import numpy as np

from conftest import package, rel_rms, synth_input
import scenarios

TOL_EXACT = 2e-6   # no phase-vocoder feedback involved (identity, carry/ring bookkeeping)
TOL_SHORT = 1e-3
TOL_LONG = 5e-3    # SURVEY App. D.2 (iii), tonal input


def make_pair(lib, ref, channels, cfg, seed=0):
    pkg = package()
    g = pkg.SignalsmithStretch(seed=seed, lib=lib)
    r = ref.RefStretch(seed)
    scenarios.configure(g, channels, cfg)
    scenarios.configure(r, channels, cfg)
    return g, r


def case_golden(lib, ref, name):
    """Product vs. the WASM golden vector AND vs. the checker."""
    x, y, ops, cfg, info = scenarios.load_golden(name)
    g, r = make_pair(lib, ref, x.shape[0], cfg)
    assert (g.blockSamples(), g.intervalSamples(), g.inputLatency(), g.outputLatency()) == \
        (info["block"], info["interval"], info["inputLatency"], info["outputLatency"])
    out = scenarios.replay(g, x, ops)
    chk = scenarios.replay(r, x, ops)
    assert out.shape == y.shape
    assert rel_rms(out, y) <= scenarios.GOLDEN_TOL[name], ("vs wasm", name, rel_rms(out, y))
    assert rel_rms(out, chk) <= scenarios.GOLDEN_TOL[name], ("vs ref", name, rel_rms(out, chk))


SMALL = dict(preset="configure", block=512, interval=128, split=False)
SMALL_SPLIT = dict(preset="configure", block=512, interval=128, split=True)


def case_api_surface(lib, ref, cfg=SMALL):
    """seek / process in ragged chunks / flush / process-after-flush / outputSeek / exact, small geometry."""
    C, sr = 2, 48000
    x = synth_input(0, C, 128*150, sr) + 0.3*synth_input(1, C, 128*150, sr)
    g, r = make_pair(lib, ref, C, cfg)
    # many hops in one call (crosses the 64-hop tile boundary twice)
    a, b = g.process(x, int(x.shape[1]*1.25)), r.process(x, int(x.shape[1]*1.25))
    assert rel_rms(a, b) < TOL_SHORT
    # ragged chunk sizes, ratio 1.3 (0..5 hops per call)
    g, r = make_pair(lib, ref, C, cfg)
    rng = np.random.default_rng(0)
    pos = 0
    while pos < x.shape[1] - 700:
        ni = int(rng.integers(1, 600))
        no = int(ni*1.3)
        a, b = g.process(x[:, pos:pos + ni], no), r.process(x[:, pos:pos + ni], no)
        assert rel_rms(a, b) < TOL_SHORT or np.abs(b).max() < 1e-6, pos
        pos += ni
    # seek, then process, then flush (short), process again, flush (exactly one interval)
    g, r = make_pair(lib, ref, C, cfg)
    g.seek(x[:, :640], 0.8)
    r.seek(x[:, :640], 0.8)
    assert rel_rms(g.process(x[:, 640:4640], 5000), r.process(x[:, 640:4640], 5000)) < TOL_SHORT
    assert rel_rms(g.flush(100), r.flush(100)) < TOL_SHORT
    assert rel_rms(g.process(x[:, 5000:7000], 2000), r.process(x[:, 5000:7000], 2000)) < TOL_SHORT
    assert rel_rms(g.flush(128), r.flush(128)) < TOL_SHORT
    # outputSeek + process
    g, r = make_pair(lib, ref, C, cfg)
    n = r.outputSeekLength(0.8)
    assert g.outputSeekLength(0.8) == n and g.seekLength() == r.seekLength()
    g.outputSeek(x[:, :n])
    r.outputSeek(x[:, :n])
    assert rel_rms(g.process(x[:, n:n + 4000], 5000), r.process(x[:, n:n + 4000], 5000)) < TOL_SHORT
    # exact(): whole buffer; and the too-short input case (returns false, zeroes the output)
    g, r = make_pair(lib, ref, C, cfg)
    (a, ok_a), (b, ok_b) = g.exact(x[:, :6000], 7000), r.exact(x[:, :6000], 7000)
    assert ok_a and ok_b and rel_rms(a, b) < TOL_SHORT
    (a, ok_a), (b, ok_b) = g.exact(x[:, :200], 300), r.exact(x[:, :200], 300)
    assert not ok_a and not ok_b and np.abs(a).max() == 0


def case_split_mode(lib, ref):
    C, sr = 2, 48000
    x = synth_input(0, C, 12000, sr)
    g, r = make_pair(lib, ref, C, SMALL_SPLIT)
    assert g.outputLatency() == r.outputLatency() == 256 + 128
    assert rel_rms(g.process(x[:, :6000], 7040), r.process(x[:, :6000], 7040)) < TOL_SHORT
    # interval-aligned flush in split mode (mid-interval flushes differ by design, see DESIGN.md "deviations")
    assert rel_rms(g.flush(90), r.flush(90)) < TOL_SHORT


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
        g, r = make_pair(lib, ref, C, cfg)
        setup(g)
        setup(r)
        no = int(n*stretch)
        assert rel_rms(g.process(x, no), r.process(x, no)) < TOL_SHORT, label


def case_silence(lib, ref):
    sr = 48000
    g, r = make_pair(lib, ref, 1, SMALL)
    x = synth_input(0, 1, 4000, sr)
    z = np.zeros((1, 700), np.float32)
    for chunk in (x[:, :2000], z, z, z, z, x[:, 2000:3000], z, x[:, 3000:]):
        a, b = g.process(chunk, chunk.shape[1] + 50), r.process(chunk, chunk.shape[1] + 50)
        if np.abs(b).max() > 0:
            assert rel_rms(a, b) < TOL_SHORT
        else:
            assert np.abs(a).max() == 0


def case_channels(lib, ref, channel_counts=(1, 3, 8)):
    sr = 48000
    for C in channel_counts:
        x = synth_input(3, C, 6000, sr)
        x *= (1 + 0.3*np.arange(C))[:, None].astype(np.float32)  # different energies: exercises the max-channel hand-over
        g, r = make_pair(lib, ref, C, SMALL)
        g.setTransposeSemitones(2, 0.2)
        r.setTransposeSemitones(2, 0.2)
        assert rel_rms(g.process(x, 7000), r.process(x, 7000)) < TOL_SHORT, C


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
        r = ref.RefStretch()
        scenarios.configure(r, C, cfg)
        r.setTransposeSemitones(float(semis[s]), 0.0)
        o = r.process(xs[s][:, :nin[s]], nout[s])
        assert rel_rms(y[s][:, :nout[s]], o) < TOL_SHORT, s
    b.close()


def case_random_time_factor(lib, ref):
    """stretch > 2x randomises the vertical time factors (signalsmith-stretch.h:639-640); the RNG is
    implementation-defined in the reference, so only the energy is comparable."""
    C, sr = 1, 48000
    x = synth_input(0, C, 3000, sr)
    g, r = make_pair(lib, ref, C, SMALL)
    a, b = g.process(x, 9000), r.process(x, 9000)
    ra, rb = np.sqrt(np.mean(a[:, 2000:]**2)), np.sqrt(np.mean(b[:, 2000:]**2))
    assert abs(ra/rb - 1) < 0.1
