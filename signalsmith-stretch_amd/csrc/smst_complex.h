// Complex arithmetic on gfx950 packed-f32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32, VOP3P).
//
// A complex product written in C++ compiles to 4-6 VALU instructions: the compiler forms one lane of the packed multiply with
// a broadcast, but builds the swapped / negated operand of the second one with v_xor + v_mov.  VOP3P encodes exactly that in
// the instruction: op_sel / op_sel_hi pick, per result lane, which half of each 64-bit source is read, neg_lo / neg_hi negate
// a source for one lane.  So a complex multiply is TWO instructions, a complex multiply-add is two as well.  The bin
// recurrence (signalsmith-stretch.h:744-800) is five complex multiplies per step on ONE wave per stream, and that wave is
// a serial chain of M + 5*63 steps per 64-hop tile (it WAS the critical path of the kernel when these helpers were written; since
// then the producers' window loads are -- DESIGN.md section 5): these helpers take ~20 of its ~50 instructions per step.
//
// Rounding (every kernel uses the same helpers, so staged / gathering / single-hop / un-fused paths stay bit-identical):
//   cmul(a, b)     re = fma(a.y, -b.y, rnd(a.x*b.x))        im = fma(a.y, b.x, rnd(a.x*b.y))
//   cmulc(a, b)    re = fma(a.y,  b.y, rnd(a.x*b.x))        im = fma(a.y, b.x, rnd(-(a.x*b.y)))          a * conj(b)
//   cfma(a, b, c)  re = fma(a.y, -b.y, fma(a.x, b.x, c.x))  im = fma(a.y, b.x, fma(a.x, b.y, c.y))       a * b + c
// (reference: _impl::mul<false/true>, signalsmith-stretch.h:17-26 -- four roundings there, three here.)
//
// tests/emu/smst_complex.h is the CPU stand-in's version of this header (same formulas with std::fma).
#pragma once
#include <hip/hip_runtime.h>

namespace smst {

typedef float pk2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { // a*b          (reference _impl::mul<false>, :17-26)
	const pk2f av = {a.x, a.y}, bv = {b.x, b.y};
	pk2f t, r;
	asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(av), "v"(bv));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));
	return make_float2(r.x, r.y);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { // a*conj(b)   (reference _impl::mul<true>)
	const pk2f av = {a.x, a.y}, bv = {b.x, b.y};
	pk2f t, r;
	asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(av), "v"(bv));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));
	return make_float2(r.x, r.y);
}
__device__ __forceinline__ float2 cfma(float2 a, float2 b, float2 c) { // a*b + c
	const pk2f av = {a.x, a.y}, bv = {b.x, b.y}, cv = {c.x, c.y};
	pk2f t, r;
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(av), "v"(bv), "v"(cv));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));
	return make_float2(r.x, r.y);
}
// lo + (hi - lo)*fr on both components (getFractional, signalsmith-stretch.h:553-557): two instructions
__device__ __forceinline__ float2 clerp(float2 lo, float2 hi, float fr) {
	const pk2f l = {lo.x, lo.y}, h = {hi.x, hi.y};
	const pk2f d = h - l;
	const pk2f f = {fr, fr};
	const pk2f r = __builtin_elementwise_fma(d, f, l);
	return make_float2(r.x, r.y);
}

} // namespace smst
