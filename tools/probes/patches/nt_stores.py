"""Experiment: the tile-sized intermediates (analysis spectra, recurrence results, synthesised frames) written with NON-TEMPORAL
stores, so that 25 GB of write-once data per step do not evict the tables and the overlapping sample windows from L2.
SMST_PATCH_NT = comma list of {analyse, vocoder, synth} (default all)."""
import os
import sys
which = os.environ.get("SMST_PATCH_NT", "analyse,vocoder,synth").split(",")
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
helper = """
typedef float smstV2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ntStore(float2 *p, float2 v) { smstV2 t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, reinterpret_cast<smstV2 *>(p)); }
__device__ __forceinline__ void ntStore(float *p, float v) { __builtin_nontemporal_store(v, p); }
"""
anchor = "struct BlockCoord { int x, y, s; };"
assert s.count(anchor) == 1
s = s.replace(anchor, helper + anchor)
def rep(old, new, cnt):
    global s
    assert s.count(old) == cnt, (s.count(old), old)
    s = s.replace(old, new)
if "analyse" in which:
    rep("		if (kk < H) dst[kk] = u;\n		else dst[N - 1 - kk] = cconj(u);", "		if (kk < H) ntStore(dst + kk, u);\n		else ntStore(dst + (N - 1 - kk), cconj(u));", 2)
if "vocoder" in which:
    rep("							dst[0] = v0;\n							dst[1] = v1;", "							ntStore(dst, v0);\n							ntStore(dst + 1, v1);", 2)
if "synth" in which:
    rep("if (m < B - halfB) frame[m + halfB] = (2*v.x)*w.x;", "if (m < B - halfB) ntStore(frame + m + halfB, (2*v.x)*w.x);", 1)
    rep("if (m >= H - halfB) frame[m - H + halfB] = (2*v.y)*w.y;", "if (m >= H - halfB) ntStore(frame + (m - H + halfB), (2*v.y)*w.y);", 1)
    rep("if (m < B - halfB) frame[m + halfB] = (2*v.x)*r.z;", "if (m < B - halfB) ntStore(frame + m + halfB, (2*v.x)*r.z);", 1)
    rep("if (m >= H - halfB) frame[m - H + halfB] = (2*v.y)*r.w;", "if (m >= H - halfB) ntStore(frame + (m - H + halfB), (2*v.y)*r.w);", 1)
open(p, 'w').write(s)
