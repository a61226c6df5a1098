# timing-only ablation of kVocoderN's wide producers (build with -- -DSMST_EXPERIMENTS): SMST_DEBUG_MODE=3 every record's operands come
# from 8 rows x 64 bins of the tile (an L1-resident footprint; results garbage), =4 the same with the recurrence wave only acknowledging
import sys, os
p = os.path.join(sys.argv[1], "smst_kernels.hip")
s = open(p).read()
old = "if (row < nh && b >= 0 && b < M && !SMST_SKIP_PRODUCER_MATH(d)) computeRecord<CH, PLAIN, false, false>(d, hopsLds[row], hopsLds[row > 0 ? row - 1 : 0], s, sg, row, b, f);\n#pragma unroll\n\t\t\t\tfor (int h = 0; h < 2; ++h)"
assert s.count(old) == 1
new = """if (row < nh && b >= 0 && b < M && !SMST_SKIP_PRODUCER_MATH(d)) {
					const bool hot = d.debugMode >= 3;
					const int rowA = hot ? 1 + (row & 7) : row, bA = hot ? 8 + (b & 63) : b;
					computeRecord<CH, PLAIN, false, false>(d, hopsLds[rowA], hopsLds[rowA > 0 ? rowA - 1 : 0], s, sg, rowA, bA, f);
				}
#pragma unroll
				for (int h = 0; h < 2; ++h)"""
s = s.replace(old, new)
open(p, "w").write(s)
p = os.path.join(sys.argv[1], "smst_kernels_common.h")
s = open(p).read()
old = "#define SMST_CONSUMER_ONLY_ACKNOWLEDGES(d) ((d).debugMode == 2)"
assert old in s
s = s.replace(old, "#define SMST_CONSUMER_ONLY_ACKNOWLEDGES(d) ((d).debugMode == 2 || (d).debugMode == 4)")
open(p, "w").write(s)
# =5: the writer wave posts its passes without storing (output garbage); =6: that and zero records (the recurrence wave alone)
p = os.path.join(sys.argv[1], "smst_kernels.hip")
s = open(p).read()
old = "*reinterpret_cast<float4 *>(dst) = make_float4(v0.x, v0.y, v1.x, v1.y);"
assert s.count(old) == 1
s = s.replace(old, "if (d.debugMode < 5) " + old)
open(p, "w").write(s)
p = os.path.join(sys.argv[1], "smst_kernels_common.h")
s = open(p).read()
old = "#define SMST_SKIP_PRODUCER_MATH(d) ((d).debugMode == 1)"
assert old in s
s = s.replace(old, "#define SMST_SKIP_PRODUCER_MATH(d) ((d).debugMode == 1 || (d).debugMode == 6)")
open(p, "w").write(s)
