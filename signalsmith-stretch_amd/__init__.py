"""signalsmith-stretch_amd -- MI355X (gfx950) implementation of the Signalsmith Stretch spectral hot path.

Python is plumbing only: this module binds the C ABI of ``libsmst_hip.so`` (``include/smst.h``) with ctypes and
mirrors the reference class' method names (``signalsmith-stretch.h:38-491``) so tests read like calls on the
reference.  All arithmetic runs in the hand-written HIP kernels under ``csrc/``; there is NO CPU fallback -- if
the shared library is missing or no GPU is visible, construction raises.

The directory name contains a hyphen (it follows the reference project's name), so import it with::

    import importlib; smst = importlib.import_module("signalsmith-stretch_amd")
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (The engine's three pipeline streams should not share a hardware queue: INTEGRATION.md "Hardware queues" asks the HOST APPLICATION to
# export GPU_MAX_HW_QUEUES=8 before the HIP runtime starts.  Importing this package does not touch the environment -- bench.py and tools/
# set the variable themselves; load_library() warns once when it finds the runtime already started with fewer queues.)
# SMST_LIBRARY: measurement hook of bench.py / tools/ -- another BUILD of the same library (an A/B variant, an instrumented trace
# build under variants/), never another implementation; unset in every product use, and announced on stderr when set.  A build that
# lacks entry points of include/smst.h is refused at load unless SMST_LIBRARY_ALLOW_MISSING=1 (A/B against an older revision).
LIBRARY_PATH = os.environ.get("SMST_LIBRARY") or os.path.join(_HERE, "libsmst_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

MEM_HOST, MEM_DEVICE = 0, 1
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)
_ll = C.c_longlong

# name -> (restype, argtypes); every symbol include/smst.h declares
_SIGNATURES = {
    "smst_last_error": (C.c_char_p, []),
    "smst_reference_version": (None, [_ip]),
    "smst_device_count": (C.c_int, []),
    # single-stream
    "smst_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_long, C.c_int]),
    "smst_destroy": (None, [C.c_void_p]),
    "smst_clone": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    "smst_default_device": (C.c_int, []),
    "smst_set_default_device": (C.c_int, [C.c_int]),
    "smst_preset_default": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int]),
    "smst_preset_cheaper": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int]),
    "smst_configure": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "smst_block_samples": (C.c_int, [C.c_void_p]),
    "smst_interval_samples": (C.c_int, [C.c_void_p]),
    "smst_input_latency": (C.c_int, [C.c_void_p]),
    "smst_output_latency": (C.c_int, [C.c_void_p]),
    "smst_split_computation": (C.c_int, [C.c_void_p]),
    "smst_block_steps": (C.c_int, [C.c_void_p]),
    "smst_blocks_started": (C.c_int, [C.c_void_p]),
    "smst_seek_length": (C.c_int, [C.c_void_p]),
    "smst_output_seek_length": (C.c_int, [C.c_void_p, C.c_float]),
    "smst_reset": (C.c_int, [C.c_void_p]),
    "smst_set_transpose_factor": (C.c_int, [C.c_void_p, C.c_float, C.c_float]),
    "smst_set_transpose_semitones": (C.c_int, [C.c_void_p, C.c_float, C.c_float]),
    "smst_set_formant_factor": (C.c_int, [C.c_void_p, C.c_float, C.c_int]),
    "smst_set_formant_semitones": (C.c_int, [C.c_void_p, C.c_float, C.c_int]),
    "smst_set_formant_base": (C.c_int, [C.c_void_p, C.c_float]),
    "smst_set_freq_map_table": (C.c_int, [C.c_void_p, _fp, C.c_int]),
    "smst_seek": (C.c_int, [C.c_void_p, C.POINTER(_fp), C.c_int, C.c_double]),
    "smst_process": (C.c_int, [C.c_void_p, C.POINTER(_fp), C.c_int, C.POINTER(_fp), C.c_int]),
    "smst_flush": (C.c_int, [C.c_void_p, C.POINTER(_fp), C.c_int, C.c_float]),
    "smst_output_seek": (C.c_int, [C.c_void_p, C.POINTER(_fp), C.c_int]),
    "smst_exact": (C.c_int, [C.c_void_p, C.POINTER(_fp), C.c_int, C.POINTER(_fp), C.c_int]),
    # batch
    "smst_batch_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long]),
    "smst_batch_create_preset": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_long]),
    "smst_batch_create_ex": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_uint]),
    "smst_batch_create_preset_ex": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_long, C.c_uint]),
    "smst_batch_destroy": (None, [C.c_void_p]),
    "smst_batch_streams": (C.c_int, [C.c_void_p]),
    "smst_batch_channels": (C.c_int, [C.c_void_p]),
    "smst_batch_block_samples": (C.c_int, [C.c_void_p]),
    "smst_batch_interval_samples": (C.c_int, [C.c_void_p]),
    "smst_batch_fft_samples": (C.c_int, [C.c_void_p]),
    "smst_batch_bands": (C.c_int, [C.c_void_p]),
    "smst_batch_input_latency": (C.c_int, [C.c_void_p]),
    "smst_batch_output_latency": (C.c_int, [C.c_void_p]),
    "smst_batch_seek_length": (C.c_int, [C.c_void_p]),
    "smst_batch_half_state": (C.c_int, [C.c_void_p]),
    "smst_batch_output_seek_length": (C.c_int, [C.c_void_p, C.c_float]),
    "smst_batch_workspace_bytes": (_ll, [C.c_void_p]),
    "smst_batch_reset": (C.c_int, [C.c_void_p]),
    "smst_batch_set_transpose_factor": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float]),
    "smst_batch_set_transpose_semitones": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float]),
    "smst_batch_set_formant_factor": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int]),
    "smst_batch_set_formant_semitones": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int]),
    "smst_batch_set_formant_base": (C.c_int, [C.c_void_p, C.c_int, C.c_float]),
    "smst_batch_set_freq_map_table": (C.c_int, [C.c_void_p, C.c_int, _fp, C.c_int]),
    "smst_batch_seek": (C.c_int, [C.c_void_p, C.c_void_p, _ll, _ll, _ip, _dp, C.c_int]),
    "smst_batch_process": (C.c_int, [C.c_void_p, C.c_void_p, _ll, _ll, _ip, C.c_void_p, _ll, _ll, _ip, C.c_int]),
    "smst_batch_flush": (C.c_int, [C.c_void_p, C.c_void_p, _ll, _ll, _ip, _fp, C.c_int]),
    "smst_batch_output_seek": (C.c_int, [C.c_void_p, C.c_void_p, _ll, _ll, _ip, C.c_int]),
    "smst_batch_synchronize": (C.c_int, [C.c_void_p]),
    "smst_batch_hip_stream": (C.c_void_p, [C.c_void_p]),
    "smst_batch_enable_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "smst_batch_take_timings": (C.c_int, [C.c_void_p, _dp, C.POINTER(_ll)]),
    "smst_batch_take_host_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(_ll)]),
    "smst_batch_debug_get_state": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _fp]),
    "smst_batch_debug_get_carry": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp]),
    "smst_batch_debug_set_state": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _fp]),
    "smst_batch_debug_set_carry": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp]),
    "smst_batch_debug_get_map": (C.c_int, [C.c_void_p, C.c_int, _fp]),
    "smst_batch_debug_get_formants": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp, _fp]),
    "smst_batch_debug_allocation_events": (_ll, [C.c_void_p]),
    "smst_batch_wait_for_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "smst_batch_signal_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "smst_debug_complex_selftest": (C.c_int, [C.c_int, _fp, _fp, C.c_int]),
    "smst_debug_launch_count": (_ll, [C.c_char_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class StretchError(RuntimeError):
    pass


def bind(cdll):
    """Attach the include/smst.h prototypes to a loaded library object."""
    missing = [name for name in _SIGNATURES if not hasattr(cdll, name)]
    if missing and not (os.environ.get("SMST_LIBRARY") and os.environ.get("SMST_LIBRARY_ALLOW_MISSING") == "1"):
        raise StretchError("the library lacks entry points of include/smst.h: %s (a stale build? rebuild with __graft_entry__.build(); "
                           "an A/B build of an older revision needs SMST_LIBRARY_ALLOW_MISSING=1)" % ", ".join(missing))
    for name, (res, args) in _SIGNATURES.items():
        if name in missing:
            continue
        f = getattr(cdll, name)
        f.restype = res
        f.argtypes = args
    return cdll


def library_path():
    return LIBRARY_PATH


def load_library():
    """Load libsmst_hip.so (built by ``__graft_entry__.build()`` / ``csrc/Makefile``).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBRARY_PATH):
            raise StretchError(
                "libsmst_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIBRARY_PATH)
        # PyTorch ships its own HIP runtime; when both live in one process it has to be the first one loaded
        # (torch is the plumbing for device tensors/streams here, so load it first whenever it is installed).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        import sys
        try:  # advisory only: with 4 hardware queues (HIP's default) a step of the pipelined engine takes ~12 % longer next to an RCCL communicator
            queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
        except ValueError:
            queues = 4
        torch_mod = sys.modules.get("torch")
        if queues < 8 and torch_mod is not None and torch_mod.cuda.is_available() and torch_mod.cuda.is_initialized():
            print("signalsmith-stretch_amd: the HIP runtime is already running with GPU_MAX_HW_QUEUES=%d; export GPU_MAX_HW_QUEUES=8 before the "
                  "first HIP call so that the engine's pipeline streams get hardware queues of their own (INTEGRATION.md)" % queues, file=sys.stderr)
        if os.environ.get("SMST_LIBRARY"):
            print("signalsmith-stretch_amd: SMST_LIBRARY is set -- loading %s instead of the in-tree library (measurement hook)" % LIBRARY_PATH, file=sys.stderr)
        _lib = bind(C.CDLL(LIBRARY_PATH))
    return _lib


def build(verbose=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess
    res = subprocess.run(["make", "-C", CSRC_DIR], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise StretchError("hipcc build failed")
    return LIBRARY_PATH


def _check(lib, rc):
    if rc != 0:
        raise StretchError("smst error %d: %s" % (rc, (lib.smst_last_error() or b"").decode()))


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _int_array(values, n):
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(values, dtype=np.int32), (n,)))
    return a, a.ctypes.data_as(_ip)


def complex_selftest(values, device=0, lib=None):
    """values: [n, 7] float32 (a, b, c complex + a fraction) -> [n, 8] (a*b, a*conj(b), a*b + c, lerp) from the device helpers."""
    lib = lib if lib is not None else load_library()
    v = np.ascontiguousarray(values, np.float32).reshape(-1, 7)
    out = np.zeros((v.shape[0], 8), np.float32)
    _check(lib, lib.smst_debug_complex_selftest(device, v.ctypes.data_as(_fp), out.ctypes.data_as(_fp), v.shape[0]))
    return out


def launch_count(name, lib=None):
    """Launches of one kernel variant since the library was loaded (test hook, smst_debug_launch_count)."""
    lib = lib if lib is not None else load_library()
    return int(lib.smst_debug_launch_count(name.encode()))


class StretchBatch:
    """S independent streams with one configuration on one GPU (C ABI group 2 of include/smst.h).

    Buffers are [S, C, n] float32: numpy arrays (host memory, staged by the library) or CUDA/HIP torch tensors
    (device memory, zero-copy).  Each stream behaves like one reference ``SignalsmithStretch<float>`` instance.
    """

    def __init__(self, streams, channels, block=None, interval=None, split=None, preset=None, sample_rate=None,
                 device=0, seed=0, lib=None, half_state=False):
        """half_state: store the carried per-bin state and the overlap-add sums in fp16 (SMST_FLAG_HALF_STATE, include/smst.h)."""
        self.lib = lib if lib is not None else load_library()
        h = C.c_void_p()
        flags = 1 if half_state else 0
        if preset is not None:
            code = {"default": 0, "cheaper": 1}[preset]
            rc = self.lib.smst_batch_create_preset_ex(C.byref(h), streams, channels, code, float(sample_rate),
                                                      -1 if split is None else int(split), device, seed, flags)
        else:
            rc = self.lib.smst_batch_create_ex(C.byref(h), streams, channels, int(block), int(interval), int(bool(split)), device, seed, flags)
        _check(self.lib, rc)
        self.h = h
        self.streams, self.channels, self.device = streams, channels, device

    def close(self):
        if getattr(self, "h", None):
            self.lib.smst_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- queries (reference names)
    def blockSamples(self): return self.lib.smst_batch_block_samples(self.h)
    def intervalSamples(self): return self.lib.smst_batch_interval_samples(self.h)
    def fftSamples(self): return self.lib.smst_batch_fft_samples(self.h)
    def bands(self): return self.lib.smst_batch_bands(self.h)
    def inputLatency(self): return self.lib.smst_batch_input_latency(self.h)
    def outputLatency(self): return self.lib.smst_batch_output_latency(self.h)
    def seekLength(self): return self.lib.smst_batch_seek_length(self.h)
    def outputSeekLength(self, rate): return self.lib.smst_batch_output_seek_length(self.h, rate)
    def workspaceBytes(self): return self.lib.smst_batch_workspace_bytes(self.h)

    def reset(self): _check(self.lib, self.lib.smst_batch_reset(self.h))
    def setTransposeFactor(self, m, tonality=0.0, stream=-1): _check(self.lib, self.lib.smst_batch_set_transpose_factor(self.h, stream, m, tonality))
    def setTransposeSemitones(self, s, tonality=0.0, stream=-1): _check(self.lib, self.lib.smst_batch_set_transpose_semitones(self.h, stream, s, tonality))
    def setFormantFactor(self, m, comp=False, stream=-1): _check(self.lib, self.lib.smst_batch_set_formant_factor(self.h, stream, m, int(comp)))
    def setFormantSemitones(self, s, comp=False, stream=-1): _check(self.lib, self.lib.smst_batch_set_formant_semitones(self.h, stream, s, int(comp)))
    def setFormantBase(self, f=0.0, stream=-1): _check(self.lib, self.lib.smst_batch_set_formant_base(self.h, stream, f))

    def setFreqMapTable(self, table, stream=-1):
        if table is None:
            _check(self.lib, self.lib.smst_batch_set_freq_map_table(self.h, stream, None, 0))
        else:
            t = np.ascontiguousarray(table, np.float32)
            _check(self.lib, self.lib.smst_batch_set_freq_map_table(self.h, stream, t.ctypes.data_as(_fp), len(t)))

    def synchronize(self):
        _check(self.lib, self.lib.smst_batch_synchronize(self.h))
        self._inflight = []

    def _order_after_torch(self, *tensors, wait=True):
        """Device-memory calls are asynchronous on the batch's own HIP streams.  Order them after the producer (torch's
        current stream) with an event -- no host synchronisation -- and keep the tensors referenced until the batch's
        stream has drained, so torch's caching allocator cannot recycle their memory while our kernels still use it."""
        import torch
        if wait:
            _check(self.lib, self.lib.smst_batch_wait_for_stream(self.h, C.c_void_p(torch.cuda.current_stream(tensors[0].device).cuda_stream)))
        inflight = getattr(self, "_inflight", [])
        if len(inflight) >= 16:
            self.synchronize()
            inflight = []
        inflight.append(tensors)
        self._inflight = inflight

    def enableProfiling(self, mode=1): _check(self.lib, self.lib.smst_batch_enable_profiling(self.h, int(mode)))

    def takeHostTimes(self):
        """Host time of process() since the last take: dict(call_ms, wait_tables_ms, wait_gate_ms, work_ms, calls) -- see include/smst.h"""
        ms = (C.c_double*3)()
        n = _ll(0)
        _check(self.lib, self.lib.smst_batch_take_host_times(self.h, ms, C.byref(n)))
        return dict(call_ms=ms[0], wait_tables_ms=ms[1], wait_gate_ms=ms[2], work_ms=ms[0] - ms[1] - ms[2], calls=int(n.value))

    def takeTimings(self):
        ms = (C.c_double*8)()
        n = (_ll*6)()
        _check(self.lib, self.lib.smst_batch_take_timings(self.h, ms, n))
        keys = ["analyse", "feed", "predict", "chain", "synth", "emit", "other", "chain_live"]
        lk = ["analyse", "predict", "chain", "synth", "emit", "chain_live"]
        return dict(zip(keys, list(ms))), dict(zip(lk, list(n)))

    # --- buffers
    def _describe(self, x, what):
        """-> (pointer, streamStride, channelStride, length, memory, keepalive)"""
        if _is_torch(x):
            if x.dim() != 3 or x.shape[0] != self.streams or x.shape[1] != self.channels:
                raise StretchError("%s must be [S, C, n]" % what)
            if str(x.dtype) != "torch.float32" or not x.is_cuda or x.stride(2) != 1:
                raise StretchError("%s: need a float32 GPU tensor with contiguous samples" % what)
            return C.c_void_p(x.data_ptr()), x.stride(0), x.stride(1), x.shape[2], MEM_DEVICE, x
        a = np.asarray(x, dtype=np.float32)
        if a.ndim != 3 or a.shape[0] != self.streams or a.shape[1] != self.channels:
            raise StretchError("%s must be [S, C, n]" % what)
        a = np.ascontiguousarray(a)
        return C.c_void_p(a.ctypes.data), a.shape[1]*a.shape[2], a.shape[2], a.shape[2], MEM_HOST, a

    def process(self, x, out_samples, in_samples=None, out=None, ordered=True):
        """process(inputs, inputSamples, outputs, outputSamples) for every stream (signalsmith-stretch.h:210).

        Device tensors: with ``ordered`` (default) the call is ordered after torch's current stream (the producer of ``x``) and
        torch's current stream is ordered after it (consumers of the result) -- by events, without a host synchronisation.
        That makes consecutive calls wait for each other through torch's stream.  A caller whose inputs are already
        complete and who synchronises the batch itself before touching the outputs (``bench.py``) passes
        ``ordered=False`` and keeps the overlap of call n+1's host scheduling with call n's kernels."""
        S, Cn = self.streams, self.channels
        ptr, ss, cs, n, mem, keep = self._describe(x, "input")
        nin, pin = _int_array(n if in_samples is None else in_samples, S)
        nout, pout = _int_array(out_samples, S)
        max_out = max(int(nout.max()), 1)
        if out is None:
            if mem == MEM_DEVICE:
                import torch
                out = torch.zeros((S, Cn, max_out), dtype=torch.float32, device=x.device)
            else:
                out = np.zeros((S, Cn, max_out), np.float32)
        optr, oss, ocs, on, omem, okeep = self._describe(out, "output")
        if omem != mem:
            raise StretchError("input and output must live in the same memory space")
        if on < max_out or int(nin.max()) > n:
            raise StretchError("buffer shorter than the requested sample count")
        if mem == MEM_DEVICE and ordered:
            self._order_after_torch(x, out)
        # (ordered=False: the caller's contract -- inputs complete, outputs untouched and both tensors ALIVE until it synchronises the batch -- so
        # nothing is tracked here.  Until round 6 the tensors were still put on the in-flight list, whose every 17th entry synchronises the batch:
        # one pipeline drain per 16 calls, 2.4 ms of the bench's 17th step -- bench.py's per-step periods showed it, roofline.step_ms.in_order)
        _check(self.lib, self.lib.smst_batch_process(self.h, ptr, ss, cs, pin, optr, oss, ocs, pout, mem))
        if mem == MEM_DEVICE and ordered:  # torch ops on `out` issued from here on are ordered after our kernels (no host sync either)
            import torch
            _check(self.lib, self.lib.smst_batch_signal_stream(self.h, C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)))
        return out

    def seek(self, x, rates, in_samples=None):
        S = self.streams
        ptr, ss, cs, n, mem, keep = self._describe(x, "input")
        nin, pin = _int_array(n if in_samples is None else in_samples, S)
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(rates, dtype=np.float64), (S,)))
        if mem == MEM_DEVICE:
            self._order_after_torch(x)
        _check(self.lib, self.lib.smst_batch_seek(self.h, ptr, ss, cs, pin, r.ctypes.data_as(_dp), mem))

    def flush(self, out_samples, rates=0.0, like=None):
        """flush() of every stream with a non-negative count; a negative count leaves that stream alone (include/smst.h)"""
        S, Cn = self.streams, self.channels
        nout, pout = _int_array(out_samples, S)
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(rates, dtype=np.float32), (S,)))
        max_out = max(int(nout.max()), 1)
        if like is not None and _is_torch(like):
            import torch
            out = torch.zeros((S, Cn, max_out), dtype=torch.float32, device=like.device)
        else:
            out = np.zeros((S, Cn, max_out), np.float32)
        optr, oss, ocs, on, omem, okeep = self._describe(out, "output")
        _check(self.lib, self.lib.smst_batch_flush(self.h, optr, oss, ocs, pout, r.ctypes.data_as(_fp), omem))
        if omem == MEM_DEVICE:
            self.synchronize()
        return out

    def outputSeek(self, x, input_lengths=None):
        S = self.streams
        ptr, ss, cs, n, mem, keep = self._describe(x, "input")
        nin, pin = _int_array(n if input_lengths is None else input_lengths, S)
        if mem == MEM_DEVICE:
            self._order_after_torch(x)
        _check(self.lib, self.lib.smst_batch_output_seek(self.h, ptr, ss, cs, pin, mem))

    # --- test hooks
    def debug_state(self, stream, which):
        Cn, M = self.channels, self.bands()
        if which == 3:
            a = np.zeros((Cn, M), np.float32)
            _check(self.lib, self.lib.smst_batch_debug_get_state(self.h, stream, which, a.ctypes.data_as(_fp)))
            return a
        a = np.zeros((Cn, M, 2), np.float32)
        _check(self.lib, self.lib.smst_batch_debug_get_state(self.h, stream, which, a.ctypes.data_as(_fp)))
        return a[..., 0] + 1j*a[..., 1]

    def debug_set_state(self, stream, which, values):
        """Teacher forcing: overwrite Band.input / .prevInput / .output (complex [C, M]) or Prediction.energy ([C, M])."""
        Cn, M = self.channels, self.bands()
        if which == 3:
            a = np.ascontiguousarray(np.asarray(values, np.float32).reshape(Cn, M))
        else:
            v = np.asarray(values).reshape(Cn, M)
            a = np.ascontiguousarray(np.stack([v.real, v.imag], axis=-1).astype(np.float32))
        _check(self.lib, self.lib.smst_batch_debug_set_state(self.h, stream, which, a.ctypes.data_as(_fp)))

    def debug_set_carry(self, stream, sums, products):
        n = self.blockSamples() + self.intervalSamples()
        s = np.ascontiguousarray(np.asarray(sums, np.float32).reshape(self.channels, n))
        p = np.ascontiguousarray(np.asarray(products, np.float32).reshape(n))
        _check(self.lib, self.lib.smst_batch_debug_set_carry(self.h, stream, s.ctypes.data_as(_fp), p.ctypes.data_as(_fp)))

    def debug_map(self, stream):
        """(inputBin, freqGrad) per bin of the stream's newest hop, or None if that hop had no frequency map."""
        a = np.zeros((self.bands(), 2), np.float32)
        rc = self.lib.smst_batch_debug_get_map(self.h, stream, a.ctypes.data_as(_fp))
        if rc < 0:
            _check(self.lib, rc)
        return a if rc == 1 else None

    def debug_formants(self, stream):
        """(ratio[bands], envelope[bands], freqEstimate in bins) of the stream's newest hop, or None (no formant processing in that hop, or a
        batch that was not created with SMST_NO_FEED_FUSION=1)."""
        ratio, env, fe = np.zeros(self.bands(), np.float32), np.zeros(self.bands(), np.float32), np.zeros(1, np.float32)
        rc = self.lib.smst_batch_debug_get_formants(self.h, stream, ratio.ctypes.data_as(_fp), env.ctypes.data_as(_fp), fe.ctypes.data_as(_fp))
        if rc < 0:
            _check(self.lib, rc)
        return (ratio, env, float(fe[0])) if rc == 1 else None

    def allocation_events(self):
        return int(self.lib.smst_batch_debug_allocation_events(self.h))

    def debug_carry(self, stream):
        n = self.blockSamples() + self.intervalSamples()
        s = np.zeros((self.channels, n), np.float32)
        p = np.zeros(n, np.float32)
        _check(self.lib, self.lib.smst_batch_debug_get_carry(self.h, stream, s.ctypes.data_as(_fp), p.ctypes.data_as(_fp)))
        return s, p


class SignalsmithStretch:
    """Single-stream mirror of ``signalsmith::stretch::SignalsmithStretch<float>`` (same method names;
    buffers are [C, n] float32 numpy arrays) over the single-stream C ABI (group 1 of include/smst.h)."""

    def __init__(self, seed=0, device=0, lib=None):
        self.lib = lib if lib is not None else load_library()
        h = C.c_void_p()
        _check(self.lib, self.lib.smst_create(C.byref(h), seed, device))
        self.h = h
        self.channels = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.smst_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clone(self):
        """A copy of the object as the reference's (implicit) copy constructor makes one: configuration, parameters and the
        complete processing state; both continue identically and independently (smst_clone)."""
        other = SignalsmithStretch.__new__(SignalsmithStretch)
        other.lib, other.channels = self.lib, self.channels
        h = C.c_void_p()
        _check(self.lib, self.lib.smst_clone(C.byref(h), self.h))
        other.h = h
        return other

    def presetDefault(self, channels, sample_rate, split=False):
        self.channels = channels
        _check(self.lib, self.lib.smst_preset_default(self.h, channels, sample_rate, int(split)))

    def presetCheaper(self, channels, sample_rate, split=True):
        self.channels = channels
        _check(self.lib, self.lib.smst_preset_cheaper(self.h, channels, sample_rate, int(split)))

    def configure(self, channels, block, interval, split=False):
        self.channels = channels
        _check(self.lib, self.lib.smst_configure(self.h, channels, block, interval, int(split)))

    def blockSamples(self): return self.lib.smst_block_samples(self.h)
    def intervalSamples(self): return self.lib.smst_interval_samples(self.h)
    def inputLatency(self): return self.lib.smst_input_latency(self.h)
    def outputLatency(self): return self.lib.smst_output_latency(self.h)
    def splitComputation(self): return bool(self.lib.smst_split_computation(self.h))
    def seekLength(self): return self.lib.smst_seek_length(self.h)
    def outputSeekLength(self, rate): return self.lib.smst_output_seek_length(self.h, rate)
    def reset(self): _check(self.lib, self.lib.smst_reset(self.h))
    def setTransposeFactor(self, m, tonality=0.0): _check(self.lib, self.lib.smst_set_transpose_factor(self.h, m, tonality))
    def setTransposeSemitones(self, s, tonality=0.0): _check(self.lib, self.lib.smst_set_transpose_semitones(self.h, s, tonality))
    def setFormantFactor(self, m, comp=False): _check(self.lib, self.lib.smst_set_formant_factor(self.h, m, int(comp)))
    def setFormantSemitones(self, s, comp=False): _check(self.lib, self.lib.smst_set_formant_semitones(self.h, s, int(comp)))
    def setFormantBase(self, f=0.0): _check(self.lib, self.lib.smst_set_formant_base(self.h, f))

    def setFreqMapTable(self, table):
        if table is None:
            _check(self.lib, self.lib.smst_set_freq_map_table(self.h, None, 0))
        else:
            t = np.ascontiguousarray(table, np.float32)
            _check(self.lib, self.lib.smst_set_freq_map_table(self.h, t.ctypes.data_as(_fp), len(t)))

    def _planes(self, a):
        ptrs = (_fp*self.channels)()
        for c in range(self.channels):
            ptrs[c] = a[c].ctypes.data_as(_fp)
        return ptrs

    def _in(self, x):
        a = np.ascontiguousarray(np.asarray(x, np.float32).reshape(self.channels, -1))
        return a, self._planes(a)

    def seek(self, x, rate):
        a, p = self._in(x)
        _check(self.lib, self.lib.smst_seek(self.h, p, a.shape[1], rate))

    def process(self, x, out_samples):
        a, p = self._in(x)
        out = np.zeros((self.channels, max(out_samples, 1)), np.float32)
        _check(self.lib, self.lib.smst_process(self.h, p, a.shape[1], self._planes(out), out_samples))
        return out[:, :out_samples]

    def flush(self, out_samples, rate=0.0):
        out = np.zeros((self.channels, max(out_samples, 1)), np.float32)
        _check(self.lib, self.lib.smst_flush(self.h, self._planes(out), out_samples, rate))
        return out[:, :out_samples]

    def outputSeek(self, x):
        a, p = self._in(x)
        _check(self.lib, self.lib.smst_output_seek(self.h, p, a.shape[1]))

    def exact(self, x, out_samples):
        a, p = self._in(x)
        out = np.zeros((self.channels, max(out_samples, 1)), np.float32)
        rc = self.lib.smst_exact(self.h, p, a.shape[1], self._planes(out), out_samples)
        if rc == -3:
            return out[:, :out_samples], False
        _check(self.lib, rc)
        return out[:, :out_samples], True
