"""Diagnostic (not a test): WHICH bins carry the one-hop Band.output error of the 8-channel teacher-forced case (BASELINE config 5's
geometry: presetCheaper @ 96 kHz, 8 channels, split mode, -5 st, 1.2x)?  For every forced hop and stream it lists the bins with the largest
|product - checker| next to the perturbed-input checker's own deviation at the same bins, with the quantities that decide them: the
arg-max channel on both sides, Prediction.energy, the output map entry, |phase prediction| regime (noise-floor fallback or not).
usage: python tools/diag/diag_config5_bins.py [emu|hip] [channels] [forced_hops]       (SMST_NO_FUSE=1 compares the un-fused pair)"""
import ctypes
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import parity_cases as pc  # noqa: E402
from conftest import synth_input  # noqa: E402
import ref_oracle  # noqa: E402

pkg = importlib.import_module("signalsmith-stretch_amd")
which = sys.argv[1] if len(sys.argv) > 1 else "hip"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 8
forced_hops = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lib = pkg.load_library() if which == "hip" else pkg.bind(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libsmst_emu.so")))
cfg = dict(preset="cheaper", sample_rate=96000.0)
setup = lambda o: o.setTransposeSemitones(-5, 0)  # noqa: E731
streams, stretch, warm = (0, 1, 2), 1.2, 10
sr = 96000
b = pkg.StretchBatch(len(streams), C, lib=lib, preset="cheaper", sample_rate=96000.0)
refs = [pc.make("ref", lib, ref_oracle, C, cfg, setup, seed=i) for i in range(len(streams))]
twins = [pc.make("ref", lib, ref_oracle, C, cfg, setup, seed=i) for i in range(len(streams))]
setup(b)
I = b.intervalSamples()
M = b.bands()
total = warm + forced_hops
n_in = pc._hop_io(I, stretch, total)[1] + 8
xs = np.stack([synth_input(s, C, n_in, sr) for s in streams])
xp = np.stack([pc.perturbed(x, 1 + i) for i, x in enumerate(xs)])
rows = []
for k in range(total):
    lo, hi = pc._hop_io(I, stretch, k)
    forced = k >= warm
    if forced:
        for i, r in enumerate(refs):
            pc._inject(b, i, r)
            twins[i].copy_state_from(r)
    b.process(xs[:, :, lo:hi] if hi > lo else np.zeros((len(streams), C, 1), np.float32), I, in_samples=hi - lo)
    for i, r in enumerate(refs):
        r.process(xs[i][:, lo:hi], I)
        twins[i].process(xp[i][:, lo:hi], I)
    if not forced:
        continue
    for i, r in enumerate(refs):
        zr, zp, zt = r.bands_complex(2), np.asarray(b.debug_state(i, 2)), twins[i].bands_complex(2)  # complex [C][M]
        norm = np.sqrt(np.mean(np.abs(zr)**2))
        ep, et = np.abs(zp - zr), np.abs(zt - zr)      # [C][M]
        e_r = np.asarray(r.bands_real(4)).reshape(C, M)  # Prediction.energy (checker)
        e_p = np.asarray(b.debug_state(i, 3)).reshape(C, M)
        tot_p, tot_t = float(np.sqrt(np.mean(ep**2))/norm), float(np.sqrt(np.mean(et**2))/norm)
        per_bin_p, per_bin_t = (ep**2).sum(axis=0), (et**2).sum(axis=0)
        order = np.argsort(per_bin_p)[::-1]
        share = np.cumsum(per_bin_p[order])/max(per_bin_p.sum(), 1e-300)
        n90 = int(np.searchsorted(share, 0.9)) + 1
        top = []
        for bn in order[:8]:
            am_r, am_p = int(np.argmax(e_r[:, bn])), int(np.argmax(e_p[:, bn]))
            srt = np.sort(e_r[:, bn])
            top.append(dict(bin=int(bn), err=float(np.sqrt(per_bin_p[bn])/norm), twin_err=float(np.sqrt(per_bin_t[bn])/norm),
                            argmax_ref=am_r, argmax_prod=am_p, energy_max=float(srt[-1]), energy_2nd=float(srt[-2]) if C > 1 else 0.0,
                            mag_ref=float(np.abs(zr[:, bn]).max()), mag_err=float(np.abs(np.abs(zp[:, bn]) - np.abs(zr[:, bn])).max()),
                            phase_err_max=float(np.abs(np.angle(zp[:, bn]*np.conj(zr[:, bn]))[np.abs(zr[:, bn]) > 1e-12]).max(initial=0.0))))
        rows.append(dict(hop=k, stream=streams[i], spectrum_err=tot_p, spectrum_self=tot_t, bins_for_90pct_of_error=n90, top=top))
        print("hop %2d stream %d: product %.2e  checker-self %.2e  | 90%% of the squared error in %d of %d bins" % (k, streams[i], tot_p, tot_t, n90, M))
        for t in top[:5]:
            print("     bin %4d err %.2e (twin %.2e) argmax ref/prod %d/%d  E max %.3e 2nd %.3e  |mag err| %.1e  phase err %.2e rad" % (
                t["bin"], t["err"], t["twin_err"], t["argmax_ref"], t["argmax_prod"], t["energy_max"], t["energy_2nd"], t["mag_err"], t["phase_err_max"]))
b.close()
out = os.path.join(ROOT, "gpurun_out", "diag_config5_bins_%s.json" % which)
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rows, open(out, "w"), indent=1)
print("wrote", out)
