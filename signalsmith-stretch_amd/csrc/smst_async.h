// Vector-memory loads whose completion the KERNEL tracks, not the compiler (gfx950 inline assembly).
//
// hipcc waits for a load where its result is first used, and with branches or a loop back-edge between the load and the use its
// bookkeeping falls back to `s_waitcnt vmcnt(0)`: every outstanding load of the wave.  A producer wave that requests lines TWO
// blocks ahead needs the opposite -- "wait until the lines of THIS block have landed, leave the ones requested since in flight" --
// and on gfx9 loads return in order, so that is `s_waitcnt vmcnt(N)` with N = the number of loads issued after the ones needed.
// The compiler does not see an inline-assembly load as pending, so it inserts no wait of its own for these registers:
//   asyncLoad16(r, p) / asyncLoad8(r, p)   request 16 / 8 bytes per lane into r (r must not be read before ...)
//   asyncWait<N>()                          ... at most N younger vector-memory operations of this wave are still outstanding, and
//   asyncArrived(r)                         has been called on it (an empty asm statement that "redefines" r AFTER the wait, so no
//                                           use of r can be scheduled above the wait)
// Rules for a loop that uses them: every global load in it goes through these helpers (a compiler-tracked load would bring the
// compiler's own, coarser waits back) and the wave issues no global stores (vmcnt counts them too).
// tests/emu/smst_async.h is the CPU stand-in's version (plain loads, no waits).
#pragma once
#include <hip/hip_runtime.h>

namespace smst {

typedef float async_f4 __attribute__((ext_vector_type(4)));
typedef float async_f2 __attribute__((ext_vector_type(2)));
struct Async16 { async_f4 v; };
struct Async8 { async_f2 v; };
struct Async4 { float v; };

__device__ __forceinline__ void asyncLoad16(Async16 &r, const void *p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r.v) : "v"(p) : "memory"); }
__device__ __forceinline__ void asyncLoad8(Async8 &r, const void *p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(r.v) : "v"(p) : "memory"); }
__device__ __forceinline__ void asyncLoad4(Async4 &r, const void *p) { asm volatile("global_load_dword %0, %1, off" : "=&v"(r.v) : "v"(p) : "memory"); }
__device__ __forceinline__ void asyncClear(Async8 &r) { r.v = 0.0f; } // (a defined value for lanes that never request)
template <int N> __device__ __forceinline__ void asyncWait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void asyncArrived(Async16 &r) { asm volatile("" : "+v"(r.v)); }
__device__ __forceinline__ void asyncArrived(Async8 &r) { asm volatile("" : "+v"(r.v)); }
__device__ __forceinline__ void asyncArrived(Async4 &r) { asm volatile("" : "+v"(r.v)); }
__device__ __forceinline__ float asyncValue(const Async4 &r) { return r.v; }
__device__ __forceinline__ float4 asyncValue(const Async16 &r) { return make_float4(r.v.x, r.v.y, r.v.z, r.v.w); }
__device__ __forceinline__ float2 asyncValue(const Async8 &r) { return make_float2(r.v.x, r.v.y); }
// An empty statement that "uses and redefines" x: what produced x can no longer be sunk into a later conditional (the compiler turns
// `c ? a + lds[i] : a` into a branch around the read and waits for every read on its own: 24 LDS round trips in sequence).
__device__ __forceinline__ void keepUnconditional(float &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void keepUnconditional(int &x) { asm volatile("" : "+v"(x)); }

} // namespace smst
