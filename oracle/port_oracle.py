"""TEST INFRASTRUCTURE ONLY (oracle): ctypes front-end for oracle/libsmst_port.so (oracle/stretch_port.cpp, the plain
C++ restatement of the reference's process() path).  Same method names as the reference class."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsmst_port.so")
_fp = C.POINTER(C.c_float)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.run(["make", "-C", _HERE, "port"], check=True, capture_output=True)
        L = C.CDLL(LIB_PATH)
        L.smst_port_create.restype = C.c_void_p
        for name, args in dict(
            smst_port_destroy=[], smst_port_reset=[], smst_port_configure=[C.c_int]*4,
            smst_port_preset_default=[C.c_int, C.c_float, C.c_int], smst_port_preset_cheaper=[C.c_int, C.c_float, C.c_int],
            smst_port_set_transpose_factor=[C.c_float, C.c_float], smst_port_set_transpose_semitones=[C.c_float, C.c_float],
            smst_port_set_formant_factor=[C.c_float, C.c_int], smst_port_set_formant_semitones=[C.c_float, C.c_int],
            smst_port_set_formant_base=[C.c_float], smst_port_seek=[_fp, C.c_long, C.c_int, C.c_double],
            smst_port_process=[_fp, C.c_long, C.c_int, _fp, C.c_long, C.c_int], smst_port_flush_short=[_fp, C.c_long, C.c_int],
        ).items():
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [C.c_void_p] + args
        for name in ("smst_port_block_samples", "smst_port_interval_samples", "smst_port_input_latency", "smst_port_output_latency"):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_fp)


def available():
    return True


class PortStretch:
    def __init__(self, seed=0):
        self.L = lib()
        self.h = self.L.smst_port_create()
        self.is_port = True
        self.channels = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.L.smst_port_destroy(self.h)
            self.h = None

    def presetDefault(self, channels, sr, split=False):
        self.channels = channels
        self.L.smst_port_preset_default(self.h, channels, sr, int(split))

    def presetCheaper(self, channels, sr, split=True):
        self.channels = channels
        self.L.smst_port_preset_cheaper(self.h, channels, sr, int(split))

    def configure(self, channels, block, interval, split=False):
        self.channels = channels
        self.L.smst_port_configure(self.h, channels, block, interval, int(split))

    def reset(self): self.L.smst_port_reset(self.h)
    def blockSamples(self): return self.L.smst_port_block_samples(self.h)
    def intervalSamples(self): return self.L.smst_port_interval_samples(self.h)
    def inputLatency(self): return self.L.smst_port_input_latency(self.h)
    def outputLatency(self): return self.L.smst_port_output_latency(self.h)
    def setTransposeFactor(self, m, t=0.0): self.L.smst_port_set_transpose_factor(self.h, m, t)
    def setTransposeSemitones(self, s, t=0.0): self.L.smst_port_set_transpose_semitones(self.h, s, t)
    def setFormantFactor(self, m, comp=False): self.L.smst_port_set_formant_factor(self.h, m, int(comp))
    def setFormantSemitones(self, s, comp=False): self.L.smst_port_set_formant_semitones(self.h, s, int(comp))
    def setFormantBase(self, f=0.0): self.L.smst_port_set_formant_base(self.h, f)

    def seek(self, x, rate):
        x = np.ascontiguousarray(x, np.float32)
        self.L.smst_port_seek(self.h, _p(x), x.shape[1], x.shape[1], rate)

    def process(self, x, out_samples):
        x = np.ascontiguousarray(x, np.float32).reshape(self.channels, -1)
        out = np.zeros((self.channels, out_samples), np.float32)
        self.L.smst_port_process(self.h, _p(x), x.shape[1], x.shape[1], _p(out), out_samples, out_samples)
        return out

    def seekLength(self): return self.blockSamples() + self.intervalSamples()   # signalsmith-stretch.h:166-168
    def outputSeekLength(self, rate): return int(self.inputLatency() + rate*self.outputLatency())  # :205-207

    def _unsupported(self, what):
        import pytest
        pytest.skip("oracle/_ref is not available on this machine and the plain C++ port does not restate %s" % what)

    def setFreqMapTable(self, table): self._unsupported("setFreqMap")
    def outputSeek(self, x): self._unsupported("outputSeek")
    def exact(self, x, n): self._unsupported("exact")

    def flush(self, out_samples, rate=0.0):
        if out_samples > self.intervalSamples():
            self._unsupported("flush longer than one interval")
        out = np.zeros((self.channels, out_samples), np.float32)
        self.L.smst_port_flush_short(self.h, _p(out), out_samples, out_samples)
        return out
