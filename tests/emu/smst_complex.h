// TEST INFRASTRUCTURE ONLY: the CPU stand-in's version of signalsmith-stretch_amd/csrc/smst_complex.h (the product header is
// gfx950 inline assembly).  Same formulas, same roundings, written with std::fma.
#pragma once
#include <cmath>
#include <hip/hip_runtime.h>

namespace smst {

static inline float2 cmul(float2 a, float2 b) { // a*b
	return make_float2(std::fma(a.y, -b.y, a.x*b.x), std::fma(a.y, b.x, a.x*b.y));
}
static inline float2 cmulc(float2 a, float2 b) { // a*conj(b)
	return make_float2(std::fma(a.y, b.y, a.x*b.x), std::fma(a.y, b.x, -(a.x*b.y)));
}
static inline float2 cfma(float2 a, float2 b, float2 c) { // a*b + c
	return make_float2(std::fma(a.y, -b.y, std::fma(a.x, b.x, c.x)), std::fma(a.y, b.x, std::fma(a.x, b.y, c.y)));
}
static inline float2 clerp(float2 lo, float2 hi, float fr) {
	return make_float2(std::fma(hi.x - lo.x, fr, lo.x), std::fma(hi.y - lo.y, fr, lo.y));
}

} // namespace smst
