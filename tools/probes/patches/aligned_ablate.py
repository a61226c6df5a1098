"""Timing-only ablations of the line-aligned record producers (results wrong by construction): which ingredient bounds the block period.
usage (through tools/probes/build_variant.sh): ABLATE=<mode> python aligned_ablate.py <csrc copy>
modes: nolines (no line loads), nosmall (no row-above / rotation / carried loads), nopark (no LDS parking), nocompute (all-zero records),
       noloads (nolines + nosmall)"""
import os
import sys
SRC = sys.argv[1]
mode = os.environ["ABLATE"]
p = SRC + '/smst_kernels.hip'
s = open(p).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:60])
    s = s.replace(old, new)


if mode in ("nolines", "noloads"):
    rep("			v[i] = *reinterpret_cast<const float4 *>(lsrc[par][i] + 16*jc);", "			v[i] = make_float4(float(jc), 0.f, 0.f, 0.f);")
if mode in ("nosmall", "noloads"):
    rep("""		if (it > 0) xv = *reinterpret_cast<const float4 *>(xsrc + xcl);
		else xe = loadEnergyPair(d, xenergy + xcl);
		const int b = BS*(n - row) + st;
		rotNext1 = d.rot[min(max(b + 1, 0), M - 1)];
		rotNextL = d.rot[min(max(b + L, 0), M - 1)];""", """		xv = make_float4(float(xcl), 0.f, 1.f, 0.f);
		const int b = BS*(n - row) + st;
		rotNext1 = make_float2(1.f, float(b)*1e-9f);
		rotNextL = make_float2(1.f, float(b)*2e-9f);""")
if mode == "nopark":
    rep("""				*lower = *upper;
				*upper = (j < lines) ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);""", """				if (v[i].x == 12345.678f) *lower = v[i];""")
if mode == "nocompute":
    rep("		if (row < nh && b >= 0 && b < M) {\n			// same arithmetic as computeRecord<CH, true, false, false>, operands from the line buffers.", "		if (row < nh && b >= 0 && b < M && d.M < 0) {\n			// same arithmetic as computeRecord<CH, true, false, false>, operands from the line buffers.")
open(p, 'w').write(s)
