"""Timing experiment (results wrong by construction): kAnalyseFast with its window and twiddle tables replaced by constants in
registers (no table loads at all); samples still loaded, spectra still stored."""
import sys
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
for old, new in (("				const float4 w = win4[m]; // (winA, winB) in one 16-byte load\n", "				const float4 w = make_float4(0.5f, 0.25f, 0.125f, 0.0625f);\n"),
                 ("			for (int i = 0; i < 8; ++i) wA[i] = twA[i*MA + t];\n", "			for (int i = 0; i < 8; ++i) wA[i] = make_float4(0.6f, 0.8f, 0.8f, 0.6f);\n"),
                 ("		for (int i = 0; i < 8; ++i) wB[i] = twB[i*R3 + p];\n", "		for (int i = 0; i < 8; ++i) wB[i] = make_float4(0.6f, 0.8f, 0.8f, 0.6f);\n")):
    assert s.count(old) == 1, old
    s = s.replace(old, new)
open(p, 'w').write(s)
