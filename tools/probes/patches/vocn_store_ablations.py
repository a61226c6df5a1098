# timing-only ablations of kVocoderN's writer (build with -- -DSMST_EXPERIMENTS): SMST_DEBUG_MODE=7 stores channel 0 only (1/8 of the store
# instructions), =8 stores every channel with the non-temporal hint, =9 stores every channel's group into ONE line per row (same instruction
# count, 1/8 of the lines touched)
import sys, os
p = os.path.join(sys.argv[1], "smst_kernels.hip")
s = open(p).read()
old = "*reinterpret_cast<float4 *>(dst) = make_float4(v0.x, v0.y, v1.x, v1.y);"
assert s.count(old) == 1
new = """{
								const float4 val = make_float4(v0.x, v0.y, v1.x, v1.y);
								if (d.debugMode == 7) { if (c == 0) *reinterpret_cast<float4 *>(dst) = val; }
								else if (d.debugMode == 8) { typedef float vf4 __attribute__((ext_vector_type(4))); vf4 q = {val.x, val.y, val.z, val.w}; __builtin_nontemporal_store(q, reinterpret_cast<vf4 *>(dst)); }
								else if (d.debugMode == 9) *reinterpret_cast<float4 *>(d.OUT + rowOf(d, s, row, 0) + (b0 & 3)) = val;
								else if (d.debugMode < 5) *reinterpret_cast<float4 *>(dst) = val;
							}"""
s = s.replace(old, new)
open(p, "w").write(s)
