"""Timing experiment (results wrong by construction): kAnalyseFast stores 64 lanes x 8 bytes CONTIGUOUSLY (dst[j]) instead of
interleaving even and odd bins at a 16-byte stride."""
import sys
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
old = """	auto store = [&](int j, float2 u, int, int) {
		const int kk = 2*j;
		if (kk < H) dst[kk] = u;
		else dst[N - 1 - kk] = cconj(u);
	};"""
new = """	auto store = [&](int j, float2 u, int, int) {
		dst[j] = u;
	};"""
assert s.count(old) == 1
open(p, 'w').write(s.replace(old, new))
