"""Timing experiment (results wrong by construction): kAnalyseFast computes its spectra and stores almost none of them."""
import sys
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
old = """	auto store = [&](int j, float2 u, int, int) {
		const int kk = 2*j;
		if (kk < H) dst[kk] = u;
		else dst[N - 1 - kk] = cconj(u);
	};"""
new = """	auto store = [&](int j, float2 u, int, int) {
		const int kk = 2*j;
		if (u.x == 1234.5f) { if (kk < H) dst[kk] = u; else dst[N - 1 - kk] = cconj(u); }
	};"""
assert s.count(old) == 1
open(p, 'w').write(s.replace(old, new))
