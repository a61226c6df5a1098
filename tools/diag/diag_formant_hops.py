"""Diagnostic (GPU box): per-hop |output| distance product-vs-checker next to the checker's own response to a 1e-6 input
perturbation, config 4b (pitch map + formants), free-running.  python tools/diag/diag_formant_hops.py [hops]"""
import json
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(_ROOT, "tests"), os.path.join(_ROOT, "oracle")]
import conftest  # noqa: E402
import parity_cases as pc  # noqa: E402
import ref_oracle  # noqa: E402


def main():
    hops = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    pkg = conftest.package()
    import ctypes
    lib = pkg.bind(ctypes.CDLL(os.environ["SMST_DIAG_LIB"])) if os.environ.get("SMST_DIAG_LIB") else pkg.load_library()
    cfg = dict(preset="default", sample_rate=48000.0)

    def setup(o):
        o.setTransposeSemitones(4, 8000/48000)
        o.setFormantFactor(1, True)
        o.setFormantBase(200/48000)
    streams, C, stretch = (0, 1, 2), 2, 0.75
    S = len(streams)
    b = pkg.StretchBatch(S, C, lib=lib, preset="default", sample_rate=48000.0)
    setup(b)
    refs = [pc.make("ref", lib, ref_oracle, C, cfg, setup) for _ in streams]
    twins = [[pc.make("ref", lib, ref_oracle, C, cfg, setup) for _ in streams] for _ in range(3)]
    I = b.intervalSamples()
    n_in = pc._hop_io(I, stretch, hops)[1] + 8
    xs = np.stack([conftest.synth_input(s, C, n_in, 48000) for s in streams])
    xps = [np.stack([pc.perturbed(x, 10*j + 1 + i) for i, x in enumerate(xs)]) for j in range(3)]
    rows = []
    for k in range(hops):
        lo, hi = pc._hop_io(I, stretch, k)
        b.process(xs[:, :, lo:hi], I, in_samples=hi - lo)
        for i, r in enumerate(refs):
            r.process(xs[i][:, lo:hi], I)
            mr = np.abs(r.bands_complex(2))
            e = conftest.rel_rms(np.abs(b.debug_state(i, 2)), mr)
            own = []
            for j in range(3):
                twins[j][i].process(xps[j][i][:, lo:hi], I)
                own.append(conftest.rel_rms(np.abs(twins[j][i].bands_complex(2)), mr))
            m, mr2 = b.debug_map(i), r.output_map()
            mapdiff = float(np.abs(m[:, 0] - mr2[:, 0]).max()) if m is not None else -1.0
            mapown = float(np.abs(twins[0][i].output_map()[:, 0] - mr2[:, 0]).max())
            if m is not None and e > 5*max(own) and e > 1e-4:
                d = np.abs(m[:, 0] - mr2[:, 0])
                bad = np.nonzero(d > 1e-3)[0]
                en, sm = r.energy()
                pk = r.peaks()
                print("FLIP stream %d hop %d: %d map bins differ, range %d..%d, max %.3g" % (streams[i], k, len(bad), bad.min(), bad.max(), d.max()))
                near = [tuple(p) for p in pk if bad.min() - 30 <= p[1] <= bad.max() + 30]
                print("   checker peaks (in, out) near:", near)
                for p in near:
                    b0 = int(p[0])
                    lo_b, hi_b = max(0, b0 - 8), min(len(en), b0 + 9)
                    print("   around in-bin %d: energy-smoothed (rel):" % b0, ["%+.1e" % ((en[q] - sm[q])/sm[q]) for q in range(lo_b, hi_b)])
                Xp = b.debug_state(i, 0)
                enp = (np.abs(Xp)**2).sum(axis=0)
                Xt = twins[0][i].bands_complex(0)
                ent = (np.abs(Xt)**2).sum(axis=0)
                Xr = r.bands_complex(0)
                print("   spectrum norm %.3g, |X| at 895..905: %s" % (np.sqrt(np.mean(np.abs(Xr)**2)), ["%.2g" % v for v in np.abs(Xr[0, 895:906])]))
                print("   energy rel diff product-vs-checker 895..905:", ["%+.1e" % ((enp[q] - en[q])/en[q]) for q in range(895, 906)])
                print("   energy rel diff twin-vs-checker    895..905:", ["%+.1e" % ((ent[q] - en[q])/en[q]) for q in range(895, 906)])
                print("   |dX| product 895..905:", ["%.1e" % v for v in np.abs(Xp[0, 895:906] - Xr[0, 895:906])], " twin:", ["%.1e" % v for v in np.abs(Xt[0, 895:906] - Xr[0, 895:906])])
                print("   product map at", bad[:6], m[bad[:6], 0], "checker", mr2[bad[:6], 0])
            rows.append(dict(hop=k, stream=streams[i], err=e, own=own, map_maxdiff=mapdiff, map_maxdiff_own=mapown, npeaks=len(r.peaks())))
    b.close()
    for s in streams:
        rs = [r for r in rows if r["stream"] == s and r["hop"] >= 4]
        errs = np.array([r["err"] for r in rs])
        owns = np.array([r["own"] for r in rs])
        print("stream %d: product worst %.2e median %.2e | checker own worst %.2e median %.2e (3 seeds x %d hops)" % (
            s, errs.max(), np.median(errs), owns.max(), np.median(owns), len(rs)))
        for r in rs:
            if r["err"] > 5*max(r["own"]) and r["err"] > 1e-4:
                print("   hop %d err %.2e own %s map diff %.3g (own %.3g) peaks %d" % (r["hop"], r["err"], ["%.1e" % v for v in r["own"]], r["map_maxdiff"], r["map_maxdiff_own"], r["npeaks"]))
    out = os.path.join(_ROOT, "gpurun_out", "diag_formant_hops.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rows, open(out, "w"))


if __name__ == "__main__":
    main()
