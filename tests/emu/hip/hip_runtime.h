// TEST INFRASTRUCTURE ONLY.  A minimal stand-in for <hip/hip_runtime.h> that lets the UNMODIFIED product
// sources (signalsmith-stretch_amd/csrc/*) be compiled with g++ and executed on the CPU, one workgroup at a
// time, with the workgroup's threads as cooperative fibers (so __syncthreads() works).  It exists so the host
// scheduler and the kernels' index logic can be checked in the GPU-less build container
// (`pytest -m "not gpu"`).  The product library (libsmst_hip.so) never uses it and has no CPU path.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__

struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static inline float __int_as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; std::memcpy(&v, &f, 4); return v; }
struct dim3 {
	unsigned x, y, z;
	dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct EmuIdx { unsigned x, y, z; };
extern EmuIdx threadIdx, blockIdx, blockDim, gridDim;

using std::min;
using std::max;

// dynamic LDS of the running workgroup (the kernels declare `extern __shared__ unsigned char smemRaw[]`)
namespace smst { extern unsigned char smemRaw[]; }

void emuSyncThreads();
#define __syncthreads() emuSyncThreads()
#define __builtin_amdgcn_wave_barrier() emuSyncThreads()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a*(uint64_t)b) >> 32); }
static inline float __fmul_rn(float a, float b) { return a*b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f/std::sqrt(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f/x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }

// IEEE binary16 storage type (round to nearest even), standing in for the compiler's _Float16 (g++ 11 has none on x86)
struct EmuHalf {
	uint16_t bits;
	EmuHalf() : bits(0) {}
	explicit EmuHalf(float f) {
		uint32_t x; std::memcpy(&x, &f, 4);
		const uint32_t sign = (x >> 16) & 0x8000u;
		const int32_t exp = int32_t((x >> 23) & 0xff) - 127 + 15;
		uint32_t man = x & 0x7fffffu;
		if (((x >> 23) & 0xff) == 0xff) { bits = uint16_t(sign | 0x7c00u | (man ? 0x200u : 0)); return; }
		if (exp >= 31) { bits = uint16_t(sign | 0x7c00u); return; }
		if (exp <= 0) {
			if (exp < -10) { bits = uint16_t(sign); return; }
			man |= 0x800000u;
			const int shift = 14 - exp;
			uint32_t h = man >> shift;
			const uint32_t rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
			if (rem > halfway || (rem == halfway && (h & 1))) ++h;
			bits = uint16_t(sign | h);
			return;
		}
		uint32_t h = (uint32_t(exp) << 10) | (man >> 13);
		const uint32_t rem = man & 0x1fffu;
		if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
		bits = uint16_t(sign | h);
	}
	operator float() const {
		const uint32_t sign = uint32_t(bits & 0x8000u) << 16;
		uint32_t exp = (bits >> 10) & 0x1f, man = bits & 0x3ffu, x;
		if (exp == 0) {
			if (man == 0) x = sign;
			else { int e = -1; do { ++e; man <<= 1; } while (!(man & 0x400u)); x = sign | uint32_t(127 - 15 - e) << 23 | (man & 0x3ffu) << 13; }
		} else if (exp == 31) x = sign | 0x7f800000u | (man << 13);
		else x = sign | (exp + 127 - 15) << 23 | (man << 13);
		float f; std::memcpy(&f, &x, 4); return f;
	}
};
#define _Float16 EmuHalf

typedef int hipError_t;
#define hipSuccess 0
typedef struct EmuStream *hipStream_t;
typedef struct EmuEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
#define hipStreamNonBlocking 1

static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 8; return 0; } // a small machine: persistent kernels loop over their jobs
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)1; return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return 0; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = (hipStream_t)1; return 0; }
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x) /* used on wave-uniform values only */
#define __builtin_amdgcn_s_sleep(x) emuSyncThreads()
// wave-level votes/shuffles are only used by single-wave kernels whose lanes all reach them; the emulator runs lanes one at a
// time, so these are provided by tests/emu/hip_emu.cpp with a gather-then-yield protocol
bool emuAny(bool v);
float emuShflF(float v, int lane);
int emuShflI(int v, int lane);
#define __any(v) emuAny(v)
static inline float __shfl(float v, int lane) { return emuShflF(v, lane); }
static inline int __shfl(int v, int lane) { return emuShflI(v, lane); }
int emuDppShr1(int old, int v);
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool) { (void)ctrl; return emuDppShr1(old, src); }
static inline int atomicAdd(int *p, int v) { int o = *p; *p = o + v; return o; }
static inline int atomicMax(int *p, int v) { int o = *p; if (v > o) *p = v; return o; }
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __hip_atomic_load(p, order, scope) (*(volatile int *)(p))
#define __hip_atomic_store(p, v, order, scope) (*(volatile int *)(p) = (v))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipMemGetInfo(size_t *freeB, size_t *totalB) { *freeB = *totalB = (size_t)24 << 30; return 0; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = std::calloc(n ? n : 1, 1); return *p ? 0 : 2; }
static inline hipError_t hipFree(void *p) { std::free(p); return 0; }
#define hipHostMallocDefault 0
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::calloc(n ? n : 1, 1); return *p ? 0 : 2; }
static inline hipError_t hipHostFree(void *p) { std::free(p); return 0; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpy2D(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
	for (size_t r = 0; r < height; ++r) std::memcpy(static_cast<char *>(d) + r*dpitch, static_cast<const char *>(s) + r*spitch, width);
	return 0;
}
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return 0; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return 0; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return 0; }
#define hipEventDisableTiming 2
#define hipEventDisableSystemFence 0x20000000
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }

void emuLaunch(dim3 grid, dim3 block, size_t ldsBytes, const std::function<void()> &body);
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
	emuLaunch((grid), (block), (lds), [=]() { kernel(__VA_ARGS__); })
