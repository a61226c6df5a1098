"""The instrumentation of tools/probes/voc_trace_patch.py for the LINE-ALIGNED producers (vocoderProduceAligned): timestamps of producer 0,
the recurrence wave and the writer of workgroup 0, read by tools/probes/voc_trace.py.  Applied to a COPY of csrc/ by
tools/probes/build_variant.sh; the product sources never contain it.  Slot 13 = after an explicit wait for the producer's loads
(park = 13 -> 1 is then LDS work only)."""
import sys
SRC = sys.argv[1]
p = SRC + '/smst_kernels.hip'
s = open(p).read()
anchor = "// Staged producers (PLAIN tiles without random time factors, L <= 5)."
assert anchor in s
s = s.replace(anchor, "__device__ unsigned long long gTrace[14*400 + 8];\nvoid traceRead(void *dst) { hipMemcpyFromSymbol(dst, HIP_SYMBOL(gTrace), sizeof(gTrace)); }\n#define TRACED 1 /* the producer wave whose timeline is recorded (0 also fetches hop 0's carried taps) */\n#define TR(slot, n) do { if (s == 0 && k == 0 && (n) >= 0 && (n) < 400) { gTrace[(slot)*400 + (n)] = clock64(); if ((slot) == 5 && ((n) == 100 || (n) == 300)) gTrace[14*400 + ((n) == 300)] = wall_clock64(); } } while (0)\n" + anchor, 1)


def rep(old, new, count=1):
    global s
    assert s.count(old) == count, (s.count(old), old[:70])
    s = s.replace(old, new)


# (no stamps in the producer waves: a stamp is a global store, and the producers wait for their loads by COUNT -- smst_async.h)
rep("""		const int need = 8*(n/NB + 1);
		while (seenProduced < need) { __builtin_amdgcn_s_sleep(1); seenProduced = ldsPeek(&sync[slot]); }
		asm volatile("" ::: "memory");
		const float4 *blockRecs = recs + (size_t)slot*BS*NCH*64;
		while (n - seenWritten >= 2) { __builtin_amdgcn_s_sleep(1); seenWritten = ldsPeek(&sync[NB + 2]); } // the writer still owns this result slot
		asm volatile("" ::: "memory");""", """		const int need = 8*(n/NB + 1);
		TR(5, n);
		while (seenProduced < need) { __builtin_amdgcn_s_sleep(1); seenProduced = ldsPeek(&sync[slot]); }
		asm volatile("" ::: "memory");
		TR(6, n);
		const float4 *blockRecs = recs + (size_t)slot*BS*NCH*64;
		while (n - seenWritten >= 2) { __builtin_amdgcn_s_sleep(1); seenWritten = ldsPeek(&sync[NB + 2]); } // the writer still owns this result slot
		asm volatile("" ::: "memory");
		TR(7, n);""")
rep("""		asm volatile("" ::: "memory");
		if (k == 0) { ldsPost(&sync[NB], n + 1); ldsPost(&sync[NB + 1], n + 1); } // record slot may be refilled; results may be written out""", """		asm volatile("" ::: "memory");
		TR(8, n);
		if (k == 0) { ldsPost(&sync[NB], n + 1); ldsPost(&sync[NB + 1], n + 1); } // record slot may be refilled; results may be written out""")
rep("""			for (int n = 0; n <= totalBlocks + 1; ++n) {
				if (n < totalBlocks) {
					while (ldsPeek(&sync[NB + 1]) <= n) __builtin_amdgcn_s_sleep(2);
				}
				asm volatile("" ::: "memory");""", """			for (int n = 0; n <= totalBlocks + 1; ++n) {
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
		// 0..NP-1 over the producer waves.""", """				asm volatile("" ::: "memory");
				TR(11, n);
				if (k == 0) ldsPost(&sync[NB + 2], n + 1);
			}
			return;
		}
		// 0..NP-1 over the producer waves.""")
open(p, 'w').write(s)
p = SRC + '/smst_engine.cpp'
s = open(p).read()
old = "void Batch::debugGetState(int stream, int which, float *dst) {\n	SMST_HIP(hipSetDevice(dev));\n	SMST_HIP(hipStreamSynchronize(st));"
assert old in s
s = s.replace(old, "void traceRead(void *dst);\nvoid Batch::debugGetState(int stream, int which, float *dst) {\n	SMST_HIP(hipSetDevice(dev));\n	SMST_HIP(hipStreamSynchronize(st));\n	if (which == 7) { traceRead(dst); return; }")
open(p, 'w').write(s)
