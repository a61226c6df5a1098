// TEST INFRASTRUCTURE ONLY: the CPU stand-in's version of signalsmith-stretch_amd/csrc/smst_async.h (plain loads; nothing is in flight).
#pragma once
#include <hip/hip_runtime.h>

namespace smst {

struct Async16 { float4 v; };
struct Async8 { float2 v; };
static inline void asyncLoad16(Async16 &r, const void *p) { r.v = *static_cast<const float4 *>(p); }
static inline void asyncLoad8(Async8 &r, const void *p) { r.v = *static_cast<const float2 *>(p); }
template <int N> static inline void asyncWait() {}
static inline void asyncArrived(Async16 &) {}
static inline void asyncArrived(Async8 &) {}
static inline float4 asyncValue(const Async16 &r) { return r.v; }
static inline float2 asyncValue(const Async8 &r) { return r.v; }
static inline void keepUnconditional(float &) {}

} // namespace smst
