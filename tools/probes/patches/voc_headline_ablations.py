# timing-only ablations of kVocoder's aligned form (headline; build with -- -DSMST_EXPERIMENTS): SMST_DEBUG_MODE=5 the writer wave posts
# its passes without storing; =3 the producers' line loads all come from the first two lines of their rows (an L1-resident footprint);
# =4 both.  Results garbage.
import sys, os
p = os.path.join(sys.argv[1], "smst_kernels.hip")
s = open(p).read()
old = """						if (ok) {
							float2 *dst = d.OUT + rowOf(d, rowStream(row), rowHop(row), c) + b;
							dst[0] = v0;
							dst[1] = v1;
						}"""
assert s.count(old) == 1, s.count(old)
s = s.replace(old, old.replace("if (ok) {", "if (ok && d.debugMode != 5 && d.debugMode != 4) {"))
old = "			const int jc = min(max(lineOf(n, par, i), 0), lines - 1);\n			asyncLoad16(v[i], lsrc[par][i] + 16*jc);"
assert s.count(old) >= 1, s.count(old)
s = s.replace(old, "			const int jc0 = min(max(lineOf(n, par, i), 0), lines - 1);\n			const int jc = (d.debugMode == 3 || d.debugMode == 4) ? (jc0 & 1) : jc0;\n			asyncLoad16(v[i], lsrc[par][i] + 16*jc);")
open(p, "w").write(s)
