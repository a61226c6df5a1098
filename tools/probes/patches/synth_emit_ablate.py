"""Timing-only ablations of kSynthEmitTeams (outputs are then meaningless: run the bench with --no-self-check).  Which one: environment
SE_ABLATE = noload (spectrum loads replaced by constants) | noacc (no overlap-add reads / sums, no barriers around them) |
nostore (nothing emitted inside the hop loop) | nosync2 (the two extra team barriers removed: a race, timing only)."""
import os, sys
mode = os.environ['SE_ABLATE']
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
i = s.index('void kSynthEmitTeams(')
j = s.index('// K5: per-stream input energy')
k = s[i:j]
def rep(old, new):
    global k
    assert k.count(old) == 1, old
    k = k.replace(old, new)
if mode == 'noload':
    rep("float2 v = X[upper ? N - 1 - kk : kk];", "float2 v = make_float2(1.0f + kk, 0.5f*(float)(size_t)X);")
elif mode == 'noacc':
    rep("if (r < I && i < B) { acc[a + DQ][slot] += ex[i]; wp[a + DQ][slot] += wprodLds[i]; }", "if (r < I && i < B && d.S < 0) { acc[a + DQ][slot] += ex[i]; wp[a + DQ][slot] += wprodLds[i]; }")
    rep("sync(); // the frame is complete", "")
    rep("sync(); // the frame has been read, before the next transform's first-stage writes", "")
elif mode == 'nostore':
    rep("if (r < I) place(n0 + r, acc[0][slot], wp[0][slot]);\n			}\n#pragma unroll\n			for (int u = 0; u + 1 < NI; ++u) {", "if (r < I && d.S < 0) place(n0 + r, acc[0][slot], wp[0][slot]);\n			}\n#pragma unroll\n			for (int u = 0; u + 1 < NI; ++u) {")
elif mode == 'nosync2':
    rep("}, t, sync, sync);", "}, t, sync);")
    rep("sync(); // the frame has been read, before the next transform's first-stage writes", "")
else:
    raise SystemExit('unknown SE_ABLATE')
s = s[:i] + k + s[j:]
open(p, 'w').write(s)
