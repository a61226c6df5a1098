// TEST INFRASTRUCTURE ONLY: the CPU stand-in's version of signalsmith-stretch_amd/csrc/smst_async.h (plain loads; nothing is in flight).
#pragma once
#include <hip/hip_runtime.h>

namespace smst {

struct Async16 { float4 v; };
struct Async8 { float2 v; };
struct Async4 { float v; };
static inline void asyncLoad16(Async16 &r, const void *p) { r.v = *static_cast<const float4 *>(p); }
static inline void asyncLoad8(Async8 &r, const void *p) { r.v = *static_cast<const float2 *>(p); }
static inline void asyncLoad4(Async4 &r, const void *p) { r.v = *static_cast<const float *>(p); }
static inline void asyncClear(Async8 &r) { r.v = make_float2(0.0f, 0.0f); }
template <int N> static inline void asyncWait() {}
static inline void asyncArrived(Async16 &) {}
static inline void asyncArrived(Async8 &) {}
static inline void asyncArrived(Async4 &) {}
static inline float asyncValue(const Async4 &r) { return r.v; }
static inline float4 asyncValue(const Async16 &r) { return r.v; }
static inline float2 asyncValue(const Async8 &r) { return r.v; }
static inline void keepUnconditional(float &) {}
static inline void keepUnconditional(int &) {}

} // namespace smst
