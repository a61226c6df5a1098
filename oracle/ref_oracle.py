"""TEST INFRASTRUCTURE ONLY (oracle).

ctypes front-end for oracle/_ref/libsmst_ref.so = the UNMODIFIED reference header
(/root/reference/signalsmith-stretch.h) compiled against the L1 restatement oracle/linear_shim/.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libsmst_ref.so")
# the formant-envelope revision of the shipped WASM build (oracle/Makefile, target wasmrev): explains the WASM fixtures under
# tests/golden/wasm_revision/, never checks the product
WASMREV_LIB_PATH = os.path.join(_HERE, "_ref", "libsmst_ref_wasmrev.so")
_libs = {}
_fp = C.POINTER(C.c_float)


def available():
    return os.path.exists(LIB_PATH)


def lib(path=None):
    path = path or LIB_PATH
    if path not in _libs:
        L = C.CDLL(path)
        L.smst_ref_create.restype = C.c_void_p
        L.smst_ref_create.argtypes = [C.c_long]
        for name, args in dict(
            smst_ref_destroy=[], smst_ref_reset=[],
            smst_ref_preset_default=[C.c_int, C.c_float, C.c_int], smst_ref_preset_cheaper=[C.c_int, C.c_float, C.c_int],
            smst_ref_configure=[C.c_int, C.c_int, C.c_int, C.c_int],
            smst_ref_set_transpose_factor=[C.c_float, C.c_float], smst_ref_set_transpose_semitones=[C.c_float, C.c_float],
            smst_ref_set_formant_factor=[C.c_float, C.c_int], smst_ref_set_formant_semitones=[C.c_float, C.c_int],
            smst_ref_set_formant_base=[C.c_float], smst_ref_set_freq_map_table=[_fp, C.c_int],
            smst_ref_seek=[_fp, C.c_long, C.c_int, C.c_double], smst_ref_output_seek=[_fp, C.c_long, C.c_int],
            smst_ref_process=[_fp, C.c_long, C.c_int, _fp, C.c_long, C.c_int],
            smst_ref_flush=[_fp, C.c_long, C.c_int, C.c_float],
            smst_ref_get_bands=[C.c_int, _fp], smst_ref_get_output_map=[_fp], smst_ref_get_window=[_fp],
            smst_ref_get_output_ring=[_fp, _fp], smst_ref_analyse_block=[_fp, _fp],
            smst_ref_set_bands=[C.c_int, _fp], smst_ref_set_output_ring=[_fp, _fp], smst_ref_get_energy=[_fp, _fp],
        ).items():
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [C.c_void_p] + args
        for name, args in dict(
            smst_ref_block_samples=[], smst_ref_interval_samples=[], smst_ref_input_latency=[], smst_ref_output_latency=[],
            smst_ref_fft_samples=[], smst_ref_bands=[], smst_ref_seek_length=[], smst_ref_output_seek_length=[C.c_float],
            smst_ref_exact=[_fp, C.c_long, C.c_int, _fp, C.c_long, C.c_int], smst_ref_get_peaks=[_fp, C.c_int],
        ).items():
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p] + args
        L.smst_ref_get_formant_metric.restype = C.c_float
        L.smst_ref_get_formant_metric.argtypes = [C.c_void_p, _fp]
        _libs[path] = L
    return _libs[path]


def _p(a):
    return a.ctypes.data_as(_fp)


class RefStretch:
    """Same method names as the reference class (signalsmith-stretch.h:38-491)."""

    def __init__(self, seed=0, library=None):
        self.L = lib(library)
        self.h = self.L.smst_ref_create(seed)
        self.channels = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.L.smst_ref_destroy(self.h)
            self.h = None

    def presetDefault(self, channels, sample_rate, split=False):
        self.channels = channels
        self.L.smst_ref_preset_default(self.h, channels, sample_rate, int(split))

    def presetCheaper(self, channels, sample_rate, split=True):
        self.channels = channels
        self.L.smst_ref_preset_cheaper(self.h, channels, sample_rate, int(split))

    def configure(self, channels, block, interval, split=False):
        self.channels = channels
        self.L.smst_ref_configure(self.h, channels, block, interval, int(split))

    def reset(self): self.L.smst_ref_reset(self.h)
    def blockSamples(self): return self.L.smst_ref_block_samples(self.h)
    def intervalSamples(self): return self.L.smst_ref_interval_samples(self.h)
    def inputLatency(self): return self.L.smst_ref_input_latency(self.h)
    def outputLatency(self): return self.L.smst_ref_output_latency(self.h)
    def fftSamples(self): return self.L.smst_ref_fft_samples(self.h)
    def bands(self): return self.L.smst_ref_bands(self.h)
    def seekLength(self): return self.L.smst_ref_seek_length(self.h)
    def outputSeekLength(self, rate): return self.L.smst_ref_output_seek_length(self.h, rate)
    def setTransposeFactor(self, m, tonality=0.0): self.L.smst_ref_set_transpose_factor(self.h, m, tonality)
    def setTransposeSemitones(self, s, tonality=0.0): self.L.smst_ref_set_transpose_semitones(self.h, s, tonality)
    def setFormantFactor(self, m, comp=False): self.L.smst_ref_set_formant_factor(self.h, m, int(comp))
    def setFormantSemitones(self, s, comp=False): self.L.smst_ref_set_formant_semitones(self.h, s, int(comp))
    def setFormantBase(self, f=0.0): self.L.smst_ref_set_formant_base(self.h, f)

    def setFreqMapTable(self, table):
        if table is None:
            self.L.smst_ref_set_freq_map_table(self.h, None, 0)
        else:
            t = np.ascontiguousarray(table, np.float32)
            self.L.smst_ref_set_freq_map_table(self.h, _p(t), len(t))

    def seek(self, x, rate):
        x = np.ascontiguousarray(x, np.float32)
        self.L.smst_ref_seek(self.h, _p(x), x.shape[1], x.shape[1], rate)

    def outputSeek(self, x):
        x = np.ascontiguousarray(x, np.float32)
        self.L.smst_ref_output_seek(self.h, _p(x), x.shape[1], x.shape[1])

    def process(self, x, out_samples):
        x = np.ascontiguousarray(x, np.float32).reshape(self.channels, -1)
        out = np.zeros((self.channels, out_samples), np.float32)
        self.L.smst_ref_process(self.h, _p(x), x.shape[1], x.shape[1], _p(out), out_samples, out_samples)
        return out

    def flush(self, out_samples, rate=0.0):
        out = np.zeros((self.channels, out_samples), np.float32)
        self.L.smst_ref_flush(self.h, _p(out), out_samples, out_samples, rate)
        return out

    def exact(self, x, out_samples):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros((self.channels, out_samples), np.float32)
        ok = self.L.smst_ref_exact(self.h, _p(x), x.shape[1], x.shape[1], _p(out), out_samples, out_samples)
        return out, bool(ok)

    # --- state inspection
    def bands_complex(self, which):
        a = np.zeros((self.channels, self.bands(), 2), np.float32)
        self.L.smst_ref_get_bands(self.h, which, _p(a))
        return a[..., 0] + 1j*a[..., 1]

    def bands_real(self, which):
        a = np.zeros((self.channels, self.bands()), np.float32)
        self.L.smst_ref_get_bands(self.h, which, _p(a))
        return a

    def set_bands(self, which, values):
        if which in (3, 4):
            a = np.ascontiguousarray(np.asarray(values, np.float32).reshape(self.channels, self.bands()))
        else:
            v = np.asarray(values).reshape(self.channels, self.bands())
            a = np.ascontiguousarray(np.stack([v.real, v.imag], axis=-1).astype(np.float32))
        self.L.smst_ref_set_bands(self.h, which, _p(a))

    def set_output_ring(self, sums, products):
        s = np.ascontiguousarray(sums, np.float32)
        p = np.ascontiguousarray(products, np.float32)
        self.L.smst_ref_set_output_ring(self.h, _p(s), _p(p))

    def copy_state_from(self, other):
        """Teacher forcing between two checker instances: per-bin state and overlap-add ring := other's."""
        for which in (0, 1, 2):
            self.set_bands(which, other.bands_complex(which))
        self.set_bands(4, other.bands_real(4))
        self.set_output_ring(*other.output_ring())

    def formant_metric(self):
        """(envelope[bands + 2], freqEstimate) as updateFormants left them (signalsmith-stretch.h:968-1006)."""
        a = np.zeros(self.bands() + 2, np.float32)
        est = self.L.smst_ref_get_formant_metric(self.h, _p(a))
        return a, float(est)

    def energy(self):
        """(channel-summed energy, smoothed energy) as findPeaks saw them (signalsmith-stretch.h:818-880)."""
        e, sm = np.zeros(self.bands(), np.float32), np.zeros(self.bands(), np.float32)
        self.L.smst_ref_get_energy(self.h, _p(e), _p(sm))
        return e, sm

    def output_map(self):
        a = np.zeros((self.bands(), 2), np.float32)
        self.L.smst_ref_get_output_map(self.h, _p(a))
        return a

    def peaks(self):
        a = np.zeros((self.bands(), 2), np.float32)
        n = self.L.smst_ref_get_peaks(self.h, _p(a), self.bands())
        return a[:n]

    def window(self):
        a = np.zeros(self.blockSamples(), np.float32)
        self.L.smst_ref_get_window(self.h, _p(a))
        return a

    def output_ring(self):
        B = self.blockSamples()
        s = np.zeros((self.channels, B), np.float32)
        p = np.zeros(B, np.float32)
        self.L.smst_ref_get_output_ring(self.h, _p(s), _p(p))
        return s, p

    def analyse_block(self, block):
        b = np.ascontiguousarray(block, np.float32)
        out = np.zeros((self.bands(), 2), np.float32)
        self.L.smst_ref_analyse_block(self.h, _p(b), _p(out))
        return out[:, 0] + 1j*out[:, 1]
