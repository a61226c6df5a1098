"""Instrumentation for tools/probes/vocN_trace.py (kVocoderN, 8 channels).  ONLY through tools/probes/build_variant.sh, which hands it a
COPY of csrc/ (round 4: run with its old hard-coded default it patched the product sources in place)."""
import sys
SRC = sys.argv[1]  # required: a copy of csrc/
assert "/tmp/" in SRC or "variant" in SRC, "refusing to patch anything but a temporary copy of the sources"
p = SRC + '/smst_kernels.hip'
s=open(p).read()
hp = SRC + '/smst_kernels_common.h'
h = open(hp).read()
tail = "\n} // namespace smst\n"
assert h.rstrip().endswith("} // namespace smst")
h = h.rstrip()[:-len("} // namespace smst")] + "__device__ int gUnit;\n__device__ unsigned long long gTrace[12*400 + 8];\n#define TRP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 64) gTrace[(slot)*400 + (gUnit % 400)] = clock64(); } while (0)\ninline void traceReadImpl(void *dst) { hipMemcpyFromSymbol(dst, HIP_SYMBOL(gTrace), sizeof(gTrace)); }\n#define TR(slot, n) do { if (s == 0 && k == 0 && (n) < 400) { gTrace[(slot)*400 + (n)] = clock64(); if ((slot) == 5 && ((n) == 100 || (n) == 300)) gTrace[12*400 + ((n) == 300)] = wall_clock64(); } } while (0)\n" + tail
open(hp, 'w').write(h)
s += "\nnamespace smst { void traceRead(void *dst) { traceReadImpl(dst); } }\n"
def rep(old,new):
    global s
    assert s.count(old)==1, (s.count(old), old[:60])
    s=s.replace(old,new)
# kVocoderN producer (pIndex 0): per unit u: n = u/UNITS, it
rep("""			const int n = u/UNITS, it = u - n*UNITS;
			const int slot = n%NB;
			const int row = ROWS*it + r;""","""			const int n = u/UNITS, it = u - n*UNITS;
			const int slot = n%NB;
			if (pIndex == 0) TR(0, u/NP);
			if (pIndex == 0) TR(1, u/NP);
			if (pIndex == 0 && s == 0 && k == 0) gUnit = u/NP;
			const int row = ROWS*it + r;""")
rep("""			while (n - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2); // slot still being read
			asm volatile("" ::: "memory");
#pragma unroll
			for (int j = 0; j < NCH; ++j) recs[((slot*BS + st)*NCH + j)*64 + ((row + st) & 63)] = make_float4(f[4*j], f[4*j + 1], f[4*j + 2], f[4*j + 3]);
			asm volatile("" ::: "memory");
			if (k == 0) ldsCount(&sync[slot]); // LDS ops of a wave are in order: data first, then the count
		}
		return;
	}

	// ---------------- consumer (wave 0) ----------------
	__builtin_amdgcn_s_setprio(3);
	const int kLag = lag*k;
	float2 pf[CH], own1[CH];""","""			if (pIndex == 0) TR(2, u/NP);
			while (n - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2); // slot still being read
			asm volatile("" ::: "memory");
			if (pIndex == 0) TR(3, u/NP);
#pragma unroll
			for (int j = 0; j < NCH; ++j) recs[((slot*BS + st)*NCH + j)*64 + ((row + st) & 63)] = make_float4(f[4*j], f[4*j + 1], f[4*j + 2], f[4*j + 3]);
			asm volatile("" ::: "memory");
			if (k == 0) ldsCount(&sync[slot]); // LDS ops of a wave are in order: data first, then the count
		}
		return;
	}

	// ---------------- consumer (wave 0) ----------------
	__builtin_amdgcn_s_setprio(3);
	const int kLag = lag*k;
	float2 pf[CH], own1[CH];""")

rep("""			const int need = UNITS*(n/NB + 1);
			while (ldsPeek(&sync[slot]) < need) __builtin_amdgcn_s_sleep(1);""","""			const int need = UNITS*(n/NB + 1);
			TR(5, n);
			while (ldsPeek(&sync[slot]) < need) __builtin_amdgcn_s_sleep(1);
			TR(6, n);""")
rep("""			while (n - ldsPeek(&sync[NB + 2]) >= R/BS - 1) __builtin_amdgcn_s_sleep(1);
			asm volatile("" ::: "memory");""","""			while (n - ldsPeek(&sync[NB + 2]) >= R/BS - 1) __builtin_amdgcn_s_sleep(1);
			asm volatile("" ::: "memory");
			TR(7, n);""")
rep("""			asm volatile("" ::: "memory");
			if (k == 0) { ldsPost(&sync[NB], n + 1); ldsPost(&sync[NB + 1], n + 1); }
		}
	}
}""","""			asm volatile("" ::: "memory");
			TR(8, n);
			if (k == 0) { ldsPost(&sync[NB], n + 1); ldsPost(&sync[NB + 1], n + 1); }
		}
	}
}""")
open(p,'w').write(s)
p = SRC + '/smst_recurrence.h'
s=open(p).read()
rep("""	int mc = 0; // maximum-energy channel, first maximum wins (:729-737)
	float eMax = e[0];""","""	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	TRP(9);
	int mc = 0; // maximum-energy channel, first maximum wins (:729-737)
	float eMax = e[0];""")
rep("""	f[0] = A.x; f[1] = A.y; f[2] = B.x; f[3] = B.y; f[4] = Cc.x; f[5] = Cc.y; f[6] = Dc.x; f[7] = Dc.y;
	f[8] = __int_as_float(mc);""","""	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	TRP(10);
	f[0] = A.x; f[1] = A.y; f[2] = B.x; f[3] = B.y; f[4] = Cc.x; f[5] = Cc.y; f[6] = Dc.x; f[7] = Dc.y;
	f[8] = __int_as_float(mc);""")

open(p,'w').write(s)
p = SRC + '/smst_engine.cpp'
s=open(p).read()
o="void Batch::debugGetState(int stream, int which, float *dst) {\n	SMST_HIP(hipSetDevice(dev));\n	SMST_HIP(hipStreamSynchronize(st));"
assert o in s
s=s.replace(o,"void traceRead(void *dst);\nvoid Batch::debugGetState(int stream, int which, float *dst) {\n	SMST_HIP(hipSetDevice(dev));\n	SMST_HIP(hipStreamSynchronize(st));\n	if (which == 7) { traceRead(dst); return; }")
open(p,'w').write(s)
