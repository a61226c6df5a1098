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
static inline float __fmul_rn(float a, float b) { return a*b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f/std::sqrt(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f/x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }

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
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)1; return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return 0; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = (hipStream_t)1; return 0; }
#define __builtin_amdgcn_s_setprio(x) ((void)0)
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
