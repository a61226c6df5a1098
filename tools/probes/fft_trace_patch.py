"""Instrumentation for tools/probes/fft_trace.py; applied to a COPY of csrc/ by tools/probes/build_variant.sh:
    tools/probes/build_variant.sh fft_trace tools/probes/fft_trace_patch.py"""
import sys
SRC = sys.argv[1]
p=SRC + '/smst_kernels.hip'
s=open(p).read()
def rep(old,new,count=1):
    global s
    assert s.count(old)==count, (s.count(old), old[:70])
    s=s.replace(old,new)
anchor="template <int SIGN, int R3, bool LEAN, typename Load, typename Prep, typename Store>\n__device__ __forceinline__ void fftFast("
assert anchor in s
s=s.replace(anchor,"__device__ unsigned long long gTrace[12*400 + 8];\nvoid traceRead(void *dst) { hipMemcpyFromSymbol(dst, HIP_SYMBOL(gTrace), sizeof(gTrace)); }\n__device__ int gTraceOn;\n#define TRA(slot) do { if (SIGN < 0 && threadIdx.x == 0) { const unsigned w = blockIdx.x + gridDim.x*(blockIdx.y + gridDim.y*blockIdx.z); if (w % 97 == 0 && (w/97) < 400) gTrace[(slot)*400 + w/97] = clock64(); } } while (0)\n"+anchor,1)
rep("""	float2 v[16];
	// stage A: radix 16, stride 1
	if (t < MA) {
#pragma unroll
		for (int k = 0; k < 16; ++k) v[k] = load(t + MA*k, k);""","""	float2 v[16];
	TRA(0);
	// stage A: radix 16, stride 1
	if (t < MA) {
#pragma unroll
		for (int k = 0; k < 16; ++k) v[k] = load(t + MA*k, k);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		TRA(1);""")
rep("""			lds[17*t + n] = val; // padded index of 16 t + n
		}
	}
	__syncthreads();""","""			lds[17*t + n] = val; // padded index of 16 t + n
		}
	}
	TRA(2);
	__syncthreads();
	TRA(3);""")
rep("""			lds[q0 + 256*p + 16*n] = val;
		}
	}
	__syncthreads();""","""			lds[q0 + 256*p + 16*n] = val;
		}
	}
	TRA(4);
	__syncthreads();
	TRA(5);""")
rep("""				if (pos < R3) store(t + 256*(e + RA*c), u[pos], ready[i], e + RA*c);
			}
		}
	}
}""","""				if (pos < R3) store(t + 256*(e + RA*c), u[pos], ready[i], e + RA*c);
			}
		}
		TRA(6);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		TRA(7);
	}
}""")
open(p,'w').write(s)
p=SRC + '/smst_engine.cpp'
s=open(p).read()
o="void Batch::debugGetState(int stream, int which, float *dst) {\n	SMST_HIP(hipSetDevice(dev));\n	SMST_HIP(hipStreamSynchronize(st));"
assert o in s
s=s.replace(o,"void traceRead(void *dst);\nvoid Batch::debugGetState(int stream, int which, float *dst) {\n	SMST_HIP(hipSetDevice(dev));\n	SMST_HIP(hipStreamSynchronize(st));\n	if (which == 7) { traceRead(dst); return; }")
open(p,'w').write(s)
