# experiment: kVocoderN's wide producers touch the (P, E) lines of their NEXT pass (one byte per channel row, into a register nobody reads)
# before they compute the current one (SMST_DEBUG_MODE=10; build with -- -DSMST_EXPERIMENTS).  Results unchanged.
import sys, os
p = os.path.join(sys.argv[1], "smst_kernels.hip")
s = open(p).read()
old = "if (row < nh && b >= 0 && b < M && !SMST_SKIP_PRODUCER_MATH(d)) computeRecord<CH, PLAIN, false, false>(d, hopsLds[row], hopsLds[row > 0 ? row - 1 : 0], s, sg, row, b, f);\n#pragma unroll\n\t\t\t\tfor (int h = 0; h < 2; ++h)"
assert s.count(old) == 1
new = """int touched = 0;
				if (d.debugMode == 10 || d.debugMode == 11) {
					const int u2 = u + NP, pair2 = u2/PASSES, row2 = PR*(u2 - pair2*PASSES) + r, b2 = PS*pair2 + st8 - lag*row2;
					if (u2 < (totalBlocks/2)*PASSES && row2 < nh && b2 >= 0 && b2 < M) {
#pragma unroll
						for (int c = 0; c < CH; ++c) {
							const void *a = PLAIN ? (const void *)(inputRow(d, hopsLds[row2], s, sg, c) + b2) : (const void *)(d.PE + rowOf(d, s, row2, c) + b2);
							asm volatile("global_load_ubyte %0, %1, off" : "+v"(touched) : "v"(a) : "memory");
						}
						if (d.debugMode == 11) {
#pragma unroll
							for (int c = 0; c < CH; ++c) {
								const void *a = (const void *)(inputRow(d, hopsLds[row2], s, sg, c) + b2);
								asm volatile("global_load_ubyte %0, %1, off" : "+v"(touched) : "v"(a) : "memory");
							}
						}
					}
				}
				if (row < nh && b >= 0 && b < M && !SMST_SKIP_PRODUCER_MATH(d)) computeRecord<CH, PLAIN, false, false>(d, hopsLds[row], hopsLds[row > 0 ? row - 1 : 0], s, sg, row, b, f);
#pragma unroll
				for (int h = 0; h < 2; ++h)"""
s = s.replace(old, new)
old = "\t\t\t\t\tif (k == 0) ldsCount(&sync[h]); // LDS ops of a wave are in order: data first, then the count\n\t\t\t\t}\n"
assert s.count(old) == 1
s = s.replace(old, old + "\t\t\t\tasm volatile(\"s_waitcnt vmcnt(0)\" :: \"v\"(touched) : \"memory\"); // the pass's own loads are long done; the touches were issued before them\n")
open(p, "w").write(s)
