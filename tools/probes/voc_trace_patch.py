"""Applies the timestamp instrumentation that tools/probes/voc_trace.py reads to a COPY of the kernel / engine sources:\n    cp csrc/smst_kernels.hip csrc/smst_engine.cpp /tmp/keep/ ; python tools/probes/voc_trace_patch.py ; hipcc ... -o variants/trace.so ; restore the sources.\nThe product sources never contain it."""
import sys
SRC = sys.argv[1]  # required: a COPY of csrc/ (tools/probes/build_variant.sh makes one)
assert '/tmp/' in SRC or 'variant' in SRC, 'refusing to patch anything but a temporary copy of the sources'
p=SRC + '/smst_kernels.hip'
s=open(p).read()
anchor="// Staged producers (PLAIN tiles without random time factors, L <= 5)."
assert anchor in s
s=s.replace(anchor,"__device__ unsigned long long gTrace[14*400 + 8];\nvoid traceRead(void *dst) { hipMemcpyFromSymbol(dst, HIP_SYMBOL(gTrace), sizeof(gTrace)); }\n#define TR(slot, n) do { if (s == 0 && k == 0 && (n) < 400) { gTrace[(slot)*400 + (n)] = clock64(); if ((slot) == 5 && ((n) == 100 || (n) == 300)) gTrace[14*400 + ((n) == 300)] = wall_clock64(); } } while (0)\n"+anchor,1)
def rep(old,new):
    global s
    assert old in s, old[:60]
    s=s.replace(old,new)
rep("""	for (; n < totalBlocks; n += NPB) {
		park(n);""","""	for (; n < totalBlocks; n += NPB) {
		if (it == 0) TR(0, n);
		park(n);""")
rep("""		if (n + NPB < totalBlocks) { issue(n + NPB); if (it == 0) issueCarried(n + NPB); }
		const int slot = n%NB;""","""		if (it == 0) TR(1, n);
		if (n + NPB < totalBlocks) { issue(n + NPB); if (it == 0) issueCarried(n + NPB); }
		if (it == 0) TR(12, n);
		const int slot = n%NB;""")
rep("""		while (n - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2); // slot still being read
		asm volatile("" ::: "memory");
		const int b0 = BS*n - lag*row, b = b0 + st;""","""		while (n - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2); // slot still being read
		asm volatile("" ::: "memory");
		if (it == 0) TR(2, n);
		const int b0 = BS*n - lag*row, b = b0 + st;""")
rep("""#pragma unroll
		for (int j = 0; j < NCH; ++j) recs[((slot*BS + st)*NCH + j)*64 + ((row + st) & 63)] = make_float4(f[4*j], f[4*j + 1], f[4*j + 2], f[4*j + 3]);
		asm volatile("" ::: "memory");
		if (k == 0) ldsCount(&sync[slot]);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");""","""		if (it == 0) TR(3, n);
#pragma unroll
		for (int j = 0; j < NCH; ++j) recs[((slot*BS + st)*NCH + j)*64 + ((row + st) & 63)] = make_float4(f[4*j], f[4*j + 1], f[4*j + 2], f[4*j + 3]);
		asm volatile("" ::: "memory");
		if (k == 0) ldsCount(&sync[slot]);
		if (it == 0) TR(4, n);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");""")
rep("""		const int need = 8*(n/NB + 1);
		while (seenProduced < need) { __builtin_amdgcn_s_sleep(1); seenProduced = ldsPeek(&sync[slot]); }
		asm volatile("" ::: "memory");
		const float4 *blockRecs = recs + (size_t)slot*BS*NCH*64;
		while (n - seenWritten >= 2) { __builtin_amdgcn_s_sleep(1); seenWritten = ldsPeek(&sync[NB + 2]); } // the writer still owns this result slot
		asm volatile("" ::: "memory");""","""		const int need = 8*(n/NB + 1);
		TR(5, n);
		while (seenProduced < need) { __builtin_amdgcn_s_sleep(1); seenProduced = ldsPeek(&sync[slot]); }
		asm volatile("" ::: "memory");
		TR(6, n);
		const float4 *blockRecs = recs + (size_t)slot*BS*NCH*64;
		while (n - seenWritten >= 2) { __builtin_amdgcn_s_sleep(1); seenWritten = ldsPeek(&sync[NB + 2]); } // the writer still owns this result slot
		asm volatile("" ::: "memory");
		TR(7, n);""")
rep("""		asm volatile("" ::: "memory");
		if (k == 0) { ldsPost(&sync[NB], n + 1); ldsPost(&sync[NB + 1], n + 1); } // record slot may be refilled; results may be written out""","""		asm volatile("" ::: "memory");
		TR(8, n);
		if (k == 0) { ldsPost(&sync[NB], n + 1); ldsPost(&sync[NB + 1], n + 1); } // record slot may be refilled; results may be written out""")
rep("""			for (int n = 0; n <= totalBlocks + 1; ++n) {
				if (n < totalBlocks) {
					while (ldsPeek(&sync[NB + 1]) <= n) __builtin_amdgcn_s_sleep(2);
				}
				asm volatile("" ::: "memory");""","""			for (int n = 0; n <= totalBlocks + 1; ++n) {
				TR(9, n);
				if (n < totalBlocks) {
					while (ldsPeek(&sync[NB + 1]) <= n) __builtin_amdgcn_s_sleep(2);
				}
				asm volatile("" ::: "memory");
				TR(10, n);""")
rep("""				asm volatile("" ::: "memory");
				if (k == 0) ldsPost(&sync[NB + 2], n + 1);
			}
			return;
		}
		// 0..NP-1 over the producer waves.""","""				asm volatile("" ::: "memory");
				TR(11, n);
				if (k == 0) ldsPost(&sync[NB + 2], n + 1);
			}
			return;
		}
		// 0..NP-1 over the producer waves.""")
open(p,'w').write(s)
p=SRC + '/smst_engine.cpp'
s=open(p).read()
rep2_old="void Batch::debugGetState(int stream, int which, float *dst) {\n	SMST_HIP(hipSetDevice(dev));\n	SMST_HIP(hipStreamSynchronize(st));"
assert rep2_old in s
s=s.replace(rep2_old,"void traceRead(void *dst);\nvoid Batch::debugGetState(int stream, int which, float *dst) {\n	SMST_HIP(hipSetDevice(dev));\n	SMST_HIP(hipStreamSynchronize(st));\n	if (which == 7) { traceRead(dst); return; }")
open(p,'w').write(s)
