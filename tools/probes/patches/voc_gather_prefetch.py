# experiment: kVocoder's wide gathering producers touch lines of their NEXT pass (one byte each, into a register nobody reads) before they
# compute the current one: SMST_DEBUG_MODE=10 the (P, E) rows and the map row at the pass's bins, =11 also the input rows.  Results unchanged.
import sys, os
p = os.path.join(sys.argv[1], "smst_kernels.hip")
s = open(p).read()
old = """				if (row < nh && b >= 0 && b < M && !SMST_SKIP_PRODUCER_MATH(d))
					computeRecord<CH, PLAIN, false, false, NCH*4, ROTL, true>(d, hopsLds[row], hopsLds[row > 0 ? row - 1 : 0], s, sg, row, b, f, rotLds);
#pragma unroll
				for (int h = 0; h < 2; ++h) {"""
assert s.count(old) == 1, s.count(old)
new = """				int touched = 0;
				if (!PLAIN && (d.debugMode == 10 || d.debugMode == 11)) {
					const int u2 = u + NP, pair2 = u2 >> 4, row2 = 4*(u2 & 15) + r4, b2 = 2*BS*pair2 + st16 - lag*row2;
					if (u2 < (totalBlocks/2)*16 && row2 < nh && b2 >= 0 && b2 < M) {
#pragma unroll
						for (int c = 0; c < CH; ++c) {
							const void *a = (const void *)(d.PE + rowOf(d, s, row2, c) + b2);
							asm volatile("global_load_ubyte %0, %1, off" : "+v"(touched) : "v"(a) : "memory");
						}
						{
							const void *a = (const void *)(d.map + ((size_t)s*d.T + row2)*M + b2);
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
				if (row < nh && b >= 0 && b < M && !SMST_SKIP_PRODUCER_MATH(d))
					computeRecord<CH, PLAIN, false, false, NCH*4, ROTL, true>(d, hopsLds[row], hopsLds[row > 0 ? row - 1 : 0], s, sg, row, b, f, rotLds);
#pragma unroll
				for (int h = 0; h < 2; ++h) {"""
s = s.replace(old, new)
old = """					if (k == 0) ldsCount(&sync[slot]); // LDS ops of a wave are in order: data first, then the count
				}
			}
			return;
		}
		const int st = k & 7, r = k >> 3; // 8 adjacent lanes"""
assert s.count(old) == 1, s.count(old)
new = """					if (k == 0) ldsCount(&sync[slot]); // LDS ops of a wave are in order: data first, then the count
				}
				asm volatile("s_waitcnt vmcnt(0)" :: "v"(touched) : "memory"); // the pass's own loads are long done; the touches were issued before them
			}
			return;
		}
		const int st = k & 7, r = k >> 3; // 8 adjacent lanes"""
s = s.replace(old, new)
open(p, "w").write(s)
