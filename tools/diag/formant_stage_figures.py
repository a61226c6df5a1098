import sys, os, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p_ in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")): sys.path.insert(0, p_)
import importlib
pkg = importlib.import_module("signalsmith-stretch_amd")
import ref_oracle, parity_cases as pc
class MP:
    def setenv(self, k, v): os.environ[k] = v
pc.TOL_FORMANT_STAGE = 1.0
lib = pkg.load_library()
D48 = dict(preset="default", sample_rate=48000.0)
for streams in ((0,), (1,), (2,), (3,), (4,), (5,)):
    print(streams, json.dumps(pc.case_formant_stages(lib, ref_oracle, MP(), D48, hops=30, streams=streams)))
