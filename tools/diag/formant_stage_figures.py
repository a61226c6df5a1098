"""The figures of tests/parity_cases.case_formant_stages WITHOUT its assertions, one stream at a time: envelope / pitch estimate / energy ratio of
the product against oracle/_ref per hop, next to the checker's own response to an input perturbation of 1e-6 (EXPERIMENTS.md 6.5: the chirp
streams' ratio differs by 1.5e-4 where the checker's own moves by 5e-4).  Runs on the GPU box.  usage: python tools/diag/formant_stage_figures.py [streams ...]"""
import json
import os
import sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p_ in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p_)
import importlib
pkg = importlib.import_module("signalsmith-stretch_amd")
import ref_oracle
import parity_cases as pc


class Env:  # what the case needs of pytest's monkeypatch
    def setenv(self, key, value):
        os.environ[key] = value


def main():
    pc.TOL_FORMANT_STAGE = 1.0  # report, do not assert
    lib = pkg.load_library()
    d48 = dict(preset="default", sample_rate=48000.0)
    for s in [int(a) for a in sys.argv[1:]] or range(6):
        print((s,), json.dumps(pc.case_formant_stages(lib, ref_oracle, Env(), d48, hops=30, streams=(s,))))


if __name__ == "__main__":
    main()
