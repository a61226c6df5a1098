"""The command-line tool (tools/stretch_cli.cpp, WAV in -> WAV out over the C ABI) against the reference's own CLI
(cmd/main.cpp, compiled UNMODIFIED into oracle/_ref/ref_cli with stand-ins for its absent util submodule): the on-disk
format either side of the hot path and BASELINE config 1 end to end."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, package, synth_input

REF_CLI = os.path.join(ROOT, "oracle", "_ref", "ref_cli")


def write_wav16(path, x, sr):
    data = np.clip(np.round(x.T*32768.0), -32768, 32767).astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, x.shape[0], sr, sr*x.shape[0]*2, x.shape[0]*2, 16))
        f.write(b"data" + struct.pack("<I", len(data)) + data)


def read_wav16(path):
    raw = open(path, "rb").read()
    pos = 12
    channels = 1
    while pos + 8 <= len(raw):
        tag, size = raw[pos:pos + 4], struct.unpack("<I", raw[pos + 4:pos + 8])[0]
        if tag == b"fmt ":
            channels = struct.unpack("<H", raw[pos + 10:pos + 12])[0]
        if tag == b"data":
            return np.frombuffer(raw[pos + 8:pos + 8 + size], "<i2").reshape(-1, channels).T.astype(np.float32)/32768.0
        pos += 8 + size + (size & 1)
    raise ValueError("no data chunk")


def run_both(cli, tmp_path, cases, flags):
    args = []
    outs = []
    for i, (x, sr) in enumerate(cases):
        src, dst, ref = (str(tmp_path/("%s%d.wav" % (n, i))) for n in ("in", "out", "ref"))
        write_wav16(src, x, sr)
        subprocess.run([REF_CLI, src, ref] + flags, check=True, capture_output=True)
        args += [src, dst]
        outs.append((dst, ref))
    res = subprocess.run([cli] + flags + args, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    for dst, ref in outs:
        a, b = read_wav16(dst), read_wav16(ref)
        assert a.shape == b.shape
        yield a, b


def cli_cases(cli, tmp_path):
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/ref_cli not built")
    sr = 44100
    x = 0.8*synth_input(0, 1, sr, sr)
    # config 1: 1 mono stream, 44.1 kHz, presetDefault, 1.0x / 0 st
    for a, b in run_both(cli, tmp_path, [(x, sr)], ["--time=1", "--semitones=0"]):
        assert np.abs(a - b).max() <= 2/32768.0
        assert np.abs(a[:, 1323:-2646] - np.round(x[:, 1323:a.shape[1] - 2646]*32768)/32768).max() <= 2/32768.0  # the input itself
    # a batch of two stereo files of different length, stretched and transposed
    sr = 48000
    files = [(0.7*synth_input(0, 2, 30000, sr), sr), (0.7*synth_input(3, 2, 23000, sr), sr)]
    for a, b in run_both(cli, tmp_path, files, ["--time=1.25", "--semitones=3", "--tonality=8000"]):
        err = np.sqrt(np.mean((a - b)**2)/np.mean(b**2))
        assert err < 2e-3, err


def test_cli_matches_reference_cli_emulated(emu, tmp_path):
    """CPU-emulated product library behind the same CLI source (host logic; the real run is test_cli_gpu)."""
    exe = str(tmp_path/"stretch_cli_emu")
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["g++", "-std=c++11", "-O2", os.path.join(ROOT, "tools", "stretch_cli.cpp"), "-o", exe, "-L" + emu_dir,
                    "-l:libsmst_emu.so", "-Wl,-rpath," + emu_dir], check=True)
    cli_cases(exe, tmp_path)


@pytest.mark.gpu
def test_cli_gpu(tmp_path):
    pkg = package()
    exe = os.path.join(os.path.dirname(pkg.LIBRARY_PATH), "stretch_cli")
    assert os.path.exists(exe), "stretch_cli not built (csrc/Makefile)"
    cli_cases(exe, tmp_path)


# ---- the reference's OWN caller through the drop-in header ------------------------------------------------------------
# oracle/Makefile (target dropin) compiles /root/reference/cmd/main.cpp UNMODIFIED, where it lies, with -I<repo>/include, so
# that its `#include "signalsmith-stretch/signalsmith-stretch.h"` (cmd/main.cpp:5) resolves to the product's drop-in header,
# and links it against libsmst_hip.so.  The binary travels to the GPU box like oracle/_ref/ref_cli.
def reference_main_cases(exe, tmp_path):
    """cmd/main.cpp:44-82 (presetDefault, setTransposeSemitones, setFormant*, outputSeek, process, flush) executed by the
    reference's own main() on the product, against the same main() on the reference header (oracle/_ref/ref_cli)."""
    if not os.path.exists(REF_CLI) or not os.path.exists(exe):
        pytest.skip("oracle/_ref binaries not built (need the reference tree at build time)")
    cases = [
        ("config 1", 0.8*synth_input(0, 1, 44100, 44100), 44100, ["--time=1", "--semitones=0"], None),
        ("stereo 1.25x +3 st", 0.7*synth_input(0, 2, 30000, 48000), 48000, ["--time=1.25", "--semitones=3", "--tonality=8000"], 2e-3),
        ("stereo 0.8x formants", 0.7*synth_input(3, 2, 30000, 48000), 48000,
         ["--time=0.8", "--semitones=-2", "--formant=2", "--formant-comp", "--formant-base=150"], 2e-2),
    ]
    for i, (label, x, sr, flags, tol) in enumerate(cases):
        src, dst, ref = (str(tmp_path/("%s_main%d.wav" % (n, i))) for n in ("in", "out", "ref"))
        write_wav16(src, x, sr)
        subprocess.run([REF_CLI, src, ref] + flags, check=True, capture_output=True)
        res = subprocess.run([exe, src, dst] + flags, capture_output=True, text=True)
        assert res.returncode == 0, (label, res.stderr[-2000:])
        a, b = read_wav16(dst), read_wav16(ref)
        assert a.shape == b.shape, label
        if tol is None:  # 1.0x / 0 st: sample-exact up to the 16-bit rounding, and the input itself
            assert np.abs(a - b).max() <= 2/32768.0, label
            assert np.abs(a[:, 1323:-2646] - np.round(x[:, 1323:a.shape[1] - 2646]*32768)/32768).max() <= 2/32768.0
        else:
            err = np.sqrt(np.mean((a - b)**2)/np.mean(b**2))
            assert err < tol, (label, err)


def test_reference_main_through_dropin_header_emulated(emu, tmp_path):
    reference_main_cases(os.path.join(ROOT, "oracle", "_ref", "main_dropin_emu"), tmp_path)


@pytest.mark.gpu
def test_reference_main_through_dropin_header_gpu(tmp_path):
    reference_main_cases(os.path.join(ROOT, "oracle", "_ref", "main_dropin"), tmp_path)
