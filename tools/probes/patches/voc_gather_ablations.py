# timing-only ablations of kVocoder's GATHERING form (mapped tiles: configs 3 / 4b; build with -- -DSMST_EXPERIMENTS): SMST_DEBUG_MODE=3 every
# record's operands from 8 rows x 64 bins (an L1-resident footprint), =5 the writer posts without storing, =4 both; 1 / 2 as in the product
# (zero records / the recurrence wave only acknowledges).  Results garbage.
import sys, os
p = os.path.join(sys.argv[1], "smst_kernels.hip")
s = open(p).read()
old = "					computeRecord<CH, PLAIN, false, false, NCH*4, ROTL, true>(d, hopsLds[row], hopsLds[row > 0 ? row - 1 : 0], s, sg, row, b, f, rotLds);"
assert s.count(old) == 1, s.count(old)
new = """					{
						const bool hot = d.debugMode == 3 || d.debugMode == 4;
						const int rowA = hot ? 1 + (row & 7) : row, bA = hot ? 8 + (b & 63) : b;
						computeRecord<CH, PLAIN, false, false, NCH*4, ROTL, true>(d, hopsLds[rowA], hopsLds[rowA > 0 ? rowA - 1 : 0], s, sg, rowA, bA, f, rotLds);
					}"""
s = s.replace(old, new)
old = """						if (ok) {
							float2 *dst = d.OUT + rowOf(d, rowStream(row), rowHop(row), c) + b;
							dst[0] = v0;
							dst[1] = v1;
						}"""
assert s.count(old) == 1, s.count(old)
s = s.replace(old, old.replace("if (ok) {", "if (ok && d.debugMode != 5 && d.debugMode != 4) {"))
open(p, "w").write(s)
