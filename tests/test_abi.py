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
