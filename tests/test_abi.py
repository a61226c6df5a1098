"""The C-ABI library loads and exports every symbol include/smst.h declares (no compute calls: runs without a GPU)."""
import ctypes
import os
import re

from conftest import ROOT, package


def header_symbols():
    text = open(os.path.join(ROOT, "include", "smst.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smst_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    pkg = package()
    if not os.path.exists(pkg.LIBRARY_PATH):
        pkg.build()
    lib = ctypes.CDLL(pkg.LIBRARY_PATH)
    names = header_symbols()
    assert len(names) >= 55
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_header():
    pkg = package()
    assert sorted(pkg.EXPORTED_SYMBOLS) == header_symbols()


def test_no_gpu_fails_loudly():
    """Without a visible GPU the product must refuse to run (no CPU fallback)."""
    pkg = package()
    lib = pkg.load_library()
    if lib.smst_device_count() > 0:
        return
    h = ctypes.c_void_p()
    rc = lib.smst_batch_create(ctypes.byref(h), 1, 1, 512, 128, 0, 0, 0)
    assert rc != 0 and b"no HIP device" in lib.smst_last_error()
    rc = lib.smst_create(ctypes.byref(h), 0, 0)
    assert rc != 0


def _steady_state_allocations(lib, geometry, S=3, C=2, calls=6):
    """process() must not allocate once it has seen the call pattern (the reference asserts the same of itself:
    cmd/main-dev.cpp:158-163 "allocated during process()"): device / pinned allocations and host-table growth are counted by
    the engine; after ONE warm-up call (the two double-buffered table sets grow together) the count has to stand still."""
    import numpy as np
    from conftest import synth_input
    pkg = package()
    b = pkg.StretchBatch(S, C, lib=lib, **geometry)
    I = b.intervalSamples()
    n = I*10
    x = np.stack([synth_input(s, C, n*(calls + 2), 48000) for s in range(S)])
    b.process(x[:, :, :n], int(n*1.25))
    before = b.allocation_events()
    for k in range(1, 1 + calls):
        b.process(x[:, :, k*n:(k + 1)*n], int(n*1.25))
    after = b.allocation_events()
    b.close()
    assert after == before, (before, after)
    assert before > 0


def test_process_does_not_allocate_in_steady_state(emu):
    _steady_state_allocations(emu, dict(block=512, interval=128, split=False))


def test_error_codes_distinguish_device_from_argument_errors(emu):
    """SMST_ERR_INVALID for bad arguments, SMST_ERR_DEVICE for HIP failures -- decided by the error's own kind, not by the text."""
    h = ctypes.c_void_p()
    assert emu.smst_batch_create(ctypes.byref(h), 1, 99, 512, 128, 0, 0, 0) == -1   # 99 channels: invalid argument
    assert emu.smst_batch_create(ctypes.byref(h), 1, 1, 512, 1, 0, 0, 0) == -1      # interval too small for the FFT size
    assert emu.smst_batch_create(ctypes.byref(h), 1, 2, 512, 128, 0, 0, 0) == 0
    assert emu.smst_batch_set_transpose_factor(h, 5, 1.0, 0.0) == -1               # stream index out of range
    emu.smst_batch_destroy(h)


def test_process_does_not_allocate_in_steady_state_split(emu):
    """... nor in split-computation mode, where every call leaves a block in flight (its tables were sized at construction)"""
    _steady_state_allocations(emu, dict(block=512, interval=128, split=True))


def test_invalid_default_device_is_an_error(emu):
    """SMST_DEVICE that names no device of the process: smst_default_device() says so and returns -1, a handle on the default device is
    refused with the value and the device count in the message -- not silently created on device 0."""
    import subprocess
    import sys
    code = (
        "import ctypes, sys\n"
        "sys.path.insert(0, %r)\n"
        "import importlib\n"
        "pkg = importlib.import_module('signalsmith-stretch_amd')\n"
        "lib = pkg.bind(ctypes.CDLL(%r))\n"
        "d = lib.smst_default_device()\n"
        "h = ctypes.c_void_p()\n"
        "rc = lib.smst_create(ctypes.byref(h), 0, d)\n"
        "print(d, rc, lib.smst_last_error().decode())\n"
    ) % (ROOT, os.path.join(ROOT, "tests", "emu", "libsmst_emu.so"))
    for value, ok in (("7", False), ("gpu1", False), ("0", True)):
        res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SMST_DEVICE=value), capture_output=True, text=True, check=True)
        d, rc, msg = res.stdout.strip().split(" ", 2) if res.stdout.strip().count(" ") >= 2 else (res.stdout.strip().split(" ") + [""])[:3]
        if ok:
            assert (int(d), int(rc)) == (0, 0), res.stdout
        else:
            assert int(d) == -1 and int(rc) != 0 and "SMST_DEVICE" in msg and value in msg and "1 device" in msg, (res.stdout, res.stderr)
            assert "SMST_DEVICE" in res.stderr
