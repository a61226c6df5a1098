// gfx950 (CDNA4, wave64) kernels for the SignalsmithStretch<float>::process() spectral hot path.
//
// Written from scratch for MI355X: no hipify, no CUDA shims, no rocFFT.  The reference computes one
// stream, one hop, one bin at a time (signalsmith-stretch.h:280-416, :633-813); here the batch of streams
// AND the time axis are the parallel axes:
//   * everything that depends only on the input (analysis FFTs, energies, peaks, frequency map, formant
//     envelope, per-bin twist coefficients) is computed for all hops of a tile at once;
//   * the one true recurrence -- Band.output, serial in the bin index and carried from hop to hop
//     (signalsmith-stretch.h:727-801) -- runs as a skewed wavefront: lane k of a wave owns hop k of a
//     64-hop tile and trails lane k-1 by `lag` bins, so 64 hops advance per step instead of one;
//   * synthesis FFTs and the overlap-add (a gather over the covering frames, no atomics) are again parallel
//     over (stream, hop, channel).
//
// This header: what every kernel translation unit shares -- the small helpers, the accessors of the carried per-stream state, the
// XCD-aware block mapping, the reference's random engine.  The kernels live in smst_fft.hip (analysis, synthesis, overlap-add /
// emission), smst_feed.hip (the feed-forward passes over a hop's spectrum), smst_vocoder.hip (the fused mono / stereo recurrence),
// smst_vocoder_n.hip (3-8 channels, single-hop tiles, the un-fused fallback and its record kernels) and smst_state.hip (what
// outlives a tile or a call).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstring>
#include "smst_device.h"
#include <smst_complex.h> // angle brackets: tests/emu shadows this header for the CPU stand-in
#include <smst_async.h>   // likewise


// Timing experiments that make the output meaningless (SMST_DEBUG_MODE=1: the record producers skip their arithmetic, =2: the recurrence
// wave only acknowledges its blocks) exist ONLY in builds with -DSMST_EXPERIMENTS (tools/probes/build_variant.sh <name> -- -DSMST_EXPERIMENTS);
// the product library has no such switch.
#ifdef SMST_EXPERIMENTS
#define SMST_SKIP_PRODUCER_MATH(d) ((d).debugMode == 1)
#define SMST_CONSUMER_ONLY_ACKNOWLEDGES(d) ((d).debugMode == 2)
#else
#define SMST_SKIP_PRODUCER_MATH(d) false
#define SMST_CONSUMER_ONLY_ACKNOWLEDGES(d) false
#endif


namespace smst {

// ------------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------------
// cmul(a, b) = a*b, cmulc(a, b) = a*conj(b), cfma(a, b, c) = a*b + c, clerp(lo, hi, fr): smst_complex.h (two packed-f32 VALU
// instructions each; reference _impl::mul<false/true>, :17-26)
// The FFT kernels keep the plain C++ products: they are bound by their memory operations, not by VALU issue, and the opaque
// assembly statements cost them the compiler's load / compute interleaving (kSynthFast 2.93 -> 3.42 ms per step with the packed
// helpers, same box)
__device__ __forceinline__ float2 cmulPlain(float2 a, float2 b) { return make_float2(a.x*b.x - a.y*b.y, a.x*b.y + a.y*b.x); }
__device__ __forceinline__ float2 cmulcPlain(float2 a, float2 b) { return make_float2(b.x*a.x + b.y*a.y, b.x*a.y - b.y*a.x); }
// |a|^2 with three separate roundings, never a fused multiply-add: the compiler otherwise contracts this differently from
// one call site to the next (mul+fma here, packed mul + add there), and the SAME energy is computed at several sites
// (carried Prediction.energy in kCarryFeed vs the producers' on-the-fly value) that must agree bit for bit
__device__ __forceinline__ float cnorm(float2 a) { return __fadd_rn(__fmul_rn(a.x, a.x), __fmul_rn(a.y, a.y)); } // :27-31
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x*s, a.y*s); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 mulI(float2 a) { return make_float2(-a.y, a.x); }
__device__ __forceinline__ float2 mulNegI(float2 a) { return make_float2(a.y, -a.x); }

// ------------------------------------------------------------------------------------------------------
// Carried per-stream state in fp32 or -- opt-in per batch (BASELINE config 5 "fp16 internal") -- in fp16: Band.output as
// half2, Prediction.energy as the half of its SQUARE ROOT (energies reach 1e6 and would overflow fp16; amplitudes do
// not), overlap-add partial sums as half.  All arithmetic stays fp32: these accessors convert at the load / store.
// d.halfState is uniform over the launch, so the branches below never diverge.
// ------------------------------------------------------------------------------------------------------
struct CarriedOutput { // Band.output after the previous tile's last hop, rows [C][M] of one stream
	const void *base;
	bool half;
	__device__ __forceinline__ float2 operator[](size_t i) const {
		if (half) { const Half2 h = static_cast<const Half2 *>(base)[i]; return make_float2(float(h.x), float(h.y)); }
		return static_cast<const float2 *>(base)[i];
	}
};
__device__ __forceinline__ CarriedOutput carriedOutput(const DevBatch &d, int sGlobal) {
	const size_t row = ((size_t)sGlobal*d.C)*(size_t)d.M;
	CarriedOutput r;
	r.half = d.halfState != 0;
	r.base = r.half ? static_cast<const void *>(reinterpret_cast<const Half2 *>(d.stOut) + row) : static_cast<const void *>(d.stOut + row);
	return r;
}
__device__ __forceinline__ void storeCarriedOutput(const DevBatch &d, size_t i, float2 v) {
	if (d.halfState) { Half2 h; h.x = half_t(v.x); h.y = half_t(v.y); reinterpret_cast<Half2 *>(d.stOut)[i] = h; }
	else d.stOut[i] = v;
}
__device__ __forceinline__ float loadCarriedEnergy(const DevBatch &d, size_t i) { // Prediction.energy of the previous tile's last hop
	if (d.halfState) { const float a = float(reinterpret_cast<const half_t *>(d.stEnergy)[i]); return a*a; }
	return d.stEnergy[i];
}
__device__ __forceinline__ void storeCarriedEnergy(const DevBatch &d, size_t i, float e) {
	if (d.halfState) reinterpret_cast<half_t *>(d.stEnergy)[i] = half_t(__builtin_amdgcn_sqrtf(e));
	else d.stEnergy[i] = e;
}
// rows of the overlap-add carry (DevBatch::carrySum / carryWp): element index of a row's first entry
__device__ __forceinline__ size_t carrySumRow(const DevBatch &d, int sg, int c) { return ((size_t)sg*d.C + c)*(size_t)d.carryPitch; }
__device__ __forceinline__ size_t carryWpRow(const DevBatch &d, int sg) { return (size_t)sg*(size_t)d.carryPitch; }
__device__ __forceinline__ float loadCarrySum(const DevBatch &d, int buf, size_t i) {
	return d.halfState ? float(reinterpret_cast<const half_t *>(d.carrySum[buf])[i]) : d.carrySum[buf][i];
}
__device__ __forceinline__ void storeCarrySum(const DevBatch &d, int buf, size_t i, float v) {
	if (d.halfState) reinterpret_cast<half_t *>(d.carrySum[buf])[i] = half_t(v);
	else d.carrySum[buf][i] = v;
}

__device__ __forceinline__ size_t rowOf(const DevBatch &d, int s, int k, int c) {
	return ((size_t)((size_t)s*d.T + k)*d.C + c)*(size_t)d.Mp; // Mp: padded row pitch (keeps lane strides off powers of two)
}
__device__ __forceinline__ size_t stateRow(const DevBatch &d, int sGlobal, int c) {
	return ((size_t)sGlobal*d.C + c)*(size_t)d.M;
}

// XCD-aware block mapping for grids (x, y, streams) whose blocks of ONE stream share data (the analysis windows of
// consecutive hops overlap by most of their length): workgroup number n is observed to run on XCD n % 8 (placement is an
// optimisation only), so streams are dealt to XCDs in groups of 8 -- all blocks of a stream land on one XCD and its L2
// serves the overlap, instead of 8 L2s each fetching the same samples.
struct BlockCoord { int x, y, s; };
// lin: position in the order in which the hardware deals workgroups (or a persistent team its jobs) to the XCDs
__device__ __forceinline__ BlockCoord xcdAwareCoord(int lin, int gx, int gy, int S) {
	const int n = gx*gy;
	const int g = lin/(8*n), rest = lin - g*8*n;
	BlockCoord r;
	int idx;
	if (8*g + 8 <= S) { r.s = 8*g + (rest & 7); idx = rest >> 3; }
	else { r.s = 8*g + rest/n; idx = rest%n; } // last, partial group: plain order
	r.x = idx%gx;
	r.y = idx/gx;
	return r;
}
__device__ __forceinline__ BlockCoord xcdAwareBlock() {
	return xcdAwareCoord(blockIdx.x + gridDim.x*(blockIdx.y + gridDim.y*blockIdx.z), gridDim.x, gridDim.y, gridDim.z);
}

// The reference draws its random time factors (stretch beyond 2x: :639-640, :749, :769) from std::default_random_engine through
// std::uniform_real_distribution<float>: implementation-defined, but DEFINED for the build a Linux user of the reference has --
// libstdc++: minstd_rand0, x <- 16807 x mod (2^31 - 1), seeded with `seed mod (2^31 - 1)` (0 becomes 1); the distribution is
// (generate_canonical<float, 24>() * (b - a)) + a with generate_canonical = float(x - 1) / 2^31, clamped below 1
// (bits/random.h:1869, bits/random.tcc:3348).  A hop makes 2M - 2 draws in bin order -- bin b: one for the upward steps (b > 0), one
// for the downward steps (b < M - 1) -- so draw j of a hop is state * 16807^(j+1): the host keeps every stream's engine state and
// advances it by 2M - 2 per randomised hop (HopDesc.seed = the state before the hop), the device jumps ahead with a table of powers.
// Same seed, same draws as the reference compiled with g++ (tests: case_engine_draws, case_random_time_factor_parity).
constexpr unsigned kLcgModulus = 2147483647u, kLcgMultiplier = 16807u;
__device__ __forceinline__ unsigned lcgMulMod(unsigned a, unsigned b) { // a*b mod (2^31 - 1), a, b < 2^31 - 1
	const unsigned long long p = (unsigned long long)a*b;
	unsigned r = unsigned(p & kLcgModulus) + unsigned(p >> 31); // 2^31 = 1 (mod m)
	r = (r & kLcgModulus) + (r >> 31);
	return r >= kLcgModulus ? r - kLcgModulus : r;
}
// draw `index` (0-based) of the hop whose engine state was `state`: uniform_real_distribution<float>(lo, hi)
__device__ __forceinline__ float engineDraw(const DevBatch &d, unsigned state, int index, float lo, float hi) {
	const unsigned x = lcgMulMod(state, d.lcgPow[index]);
	float u = float(x - 1u)*4.656612873077392578125e-10f; // / 2^31, exact
	if (u >= 1.0f) u = 0.999999940395355224609375f;     // nextafter(1, 0)
	return __fadd_rn(__fmul_rn(u, hi - lo), lo);        // two roundings, as the x86 build of the reference has no fused multiply-add
}

// ------------------------------------------------------------------------------------------------------
// host side, shared by the launchers of every translation unit
// ------------------------------------------------------------------------------------------------------
static inline int divUp(int a, int b) { return (a + b - 1)/b; }
void countLaunch(LaunchKind k); // smst_state.hip
// 3-8 channels: smst_vocoder_n.hip (launchVocoder in smst_vocoder.hip dispatches on the channel count)
void launchVocoderMany(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st);

} // namespace smst
