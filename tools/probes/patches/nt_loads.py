"""Experiment: read-once inputs of the FFT kernels (the samples of an analysis frame, the spectrum of a synthesis frame) with
NON-TEMPORAL loads, so that they do not push the window / twiddle tables out of the 32-KB L1.
SMST_PATCH_NT = comma list of {analyse, synth} (default both)."""
import os
import sys
which = os.environ.get("SMST_PATCH_NT", "analyse,synth").split(",")
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
helper = """
typedef float smstV2L __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ntLoad(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float2 ntLoad(const float2 *p) { const smstV2L t = __builtin_nontemporal_load(reinterpret_cast<const smstV2L *>(p)); return make_float2(t.x, t.y); }
"""
anchor = "struct BlockCoord { int x, y, s; };"
assert s.count(anchor) == 1
s = s.replace(anchor, helper + anchor)
def rep(old, new, cnt):
    global s
    assert s.count(old) == cnt, (s.count(old), old)
    s = s.replace(old, new)
if "analyse" in which:
    rep("const float xr = x0[m];", "const float xr = ntLoad(x0 + m);", 1)
    rep("const float xi = x1[m];", "const float xi = ntLoad(x1 + m);", 1)
if "synth" in which:
    rep("float2 v = X[upper ? N - 1 - kk : kk];", "float2 v = ntLoad(X + (upper ? N - 1 - kk : kk));", 2)
open(p, 'w').write(s)
