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
// In-LDS Stockham FFT (decimation in frequency, natural order in and out), H = 2^k * {1,3,5}.
// Radix-4 passes (+ one radix-2) come first so the stride `s` stays a power of two; the single odd-radix
// pass comes last, where all its twiddles are 1.  src/dst ping-pong; returns the buffer holding the result.
// SIGN = -1: forward (e^{-i...}), +1: inverse (unnormalised).
// ------------------------------------------------------------------------------------------------------
template <int SIGN>
__device__ __forceinline__ float2 twiddle(const float2 *__restrict__ tw, int idx) {
	float2 w = tw[idx];
	if (SIGN > 0) w.y = -w.y;
	return w;
}

template <int SIGN>
__device__ float2 *fftLds(float2 *src, float2 *dst, const FftPlan &plan, const float2 *__restrict__ tw) {
	const int H = plan.H;
	int nCur = H;
	int shift = 0; // s = 1 << shift while radices are powers of two
	for (int pass = 0; pass < plan.npass; ++pass) {
		const int r = plan.radix[pass];
		const int m = nCur/r;
		const int nb = H/r;
		const int twScale = H/nCur;
		if (r == 4) {
			const int s = 1 << shift;
			for (int t = threadIdx.x; t < nb; t += blockDim.x) {
				const int p = t >> shift, q0 = t & (s - 1);
				const float2 *in = src + q0 + (p << shift);
				const int inStride = m << shift;
				float2 *out = dst + q0 + ((4*p) << shift);
				float2 a = in[0], b = in[inStride], c = in[2*inStride], e = in[3*inStride];
				float2 apc = cadd(a, c), amc = csub(a, c), bpe = cadd(b, e), bme = csub(b, e);
				float2 jb = (SIGN < 0) ? mulNegI(bme) : mulI(bme);
				const int ti = p*twScale;
				out[0] = cadd(apc, bpe);
				out[s] = cmulPlain(cadd(amc, jb), twiddle<SIGN>(tw, ti));
				out[2*s] = cmulPlain(csub(apc, bpe), twiddle<SIGN>(tw, 2*ti));
				out[3*s] = cmulPlain(csub(amc, jb), twiddle<SIGN>(tw, 3*ti));
			}
			shift += 2;
		} else if (r == 2) {
			const int s = 1 << shift;
			for (int t = threadIdx.x; t < nb; t += blockDim.x) {
				const int p = t >> shift, q0 = t & (s - 1);
				const float2 *in = src + q0 + (p << shift);
				const int inStride = m << shift;
				float2 *out = dst + q0 + ((2*p) << shift);
				float2 a = in[0], b = in[inStride];
				out[0] = cadd(a, b);
				out[s] = cmulPlain(csub(a, b), twiddle<SIGN>(tw, p*twScale));
			}
			shift += 1;
		} else if (r == 3) { // last pass: m == 1, p == 0, all twiddles are 1
			const int s = nb;
			const float s3 = 0.86602540378443864676f;
			for (int t = threadIdx.x; t < nb; t += blockDim.x) {
				float2 a = src[t], b = src[t + s], c = src[t + 2*s];
				float2 bpc = cadd(b, c), bmc = csub(b, c);
				float2 tt = make_float2(a.x - 0.5f*bpc.x, a.y - 0.5f*bpc.y);
				float2 u = cscale((SIGN < 0) ? mulNegI(bmc) : mulI(bmc), s3);
				dst[t] = cadd(a, bpc);
				dst[t + s] = cadd(tt, u);
				dst[t + 2*s] = csub(tt, u);
			}
		} else { // r == 5, last pass
			const int s = nb;
			const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
			const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
			for (int t = threadIdx.x; t < nb; t += blockDim.x) {
				float2 a = src[t], b = src[t + s], c = src[t + 2*s], e = src[t + 3*s], f = src[t + 4*s];
				float2 bpf = cadd(b, f), bmf = csub(b, f), cpe = cadd(c, e), cme = csub(c, e);
				float2 t1 = make_float2(a.x + c1*bpf.x + c2*cpe.x, a.y + c1*bpf.y + c2*cpe.y);
				float2 t2 = make_float2(a.x + c2*bpf.x + c1*cpe.x, a.y + c2*bpf.y + c1*cpe.y);
				float2 u1 = make_float2(s1*bmf.x + s2*cme.x, s1*bmf.y + s2*cme.y);
				float2 u2 = make_float2(s2*bmf.x - s1*cme.x, s2*bmf.y - s1*cme.y);
				float2 ju1 = (SIGN < 0) ? mulNegI(u1) : mulI(u1);
				float2 ju2 = (SIGN < 0) ? mulNegI(u2) : mulI(u2);
				dst[t] = cadd(a, cadd(bpf, cpe));
				dst[t + s] = cadd(t1, ju1);
				dst[t + 2*s] = cadd(t2, ju2);
				dst[t + 3*s] = csub(t2, ju2);
				dst[t + 4*s] = csub(t1, ju1);
			}
		}
		__syncthreads();
		float2 *tmp = src; src = dst; dst = tmp;
		nCur = m;
	}
	return src;
}

// ------------------------------------------------------------------------------------------------------
// Register-blocked FFT for H = 256*R3: three Stockham stages 16 x 16 x R3, each butterfly held in registers, so the data
// (R3 = 10: 2560 bins = presetCheaper at 44.1/48 kHz; 12: 3072 = presetDefault at 44.1/48 kHz; 20: 5120 = presetCheaper at 88.2/96 kHz;
// 24: 6144 = presetDefault at 88.2/96 kHz -- every size signalsmith-stretch.h:63-68 produces up to 96 kHz; R3 = {2,4,8} x {3,5})
// crosses LDS only twice (vs. six times in the generic radix-4 ladder) and the stage-A output is padded by one
// element per 16 so that neither the 128-byte-strided writes nor the stage-B reads conflict on LDS banks.
// Twiddles come from per-stage tables laid out [n][p] (coalesced across the threads of a stage).
// ------------------------------------------------------------------------------------------------------
template <int SIGN>
__device__ __forceinline__ void dft4(float2 &a, float2 &b, float2 &c, float2 &d) {
	float2 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
	float2 jb = (SIGN < 0) ? mulNegI(bmd) : mulI(bmd);
	a = cadd(apc, bpd);
	b = cadd(amc, jb);
	c = csub(apc, bpd);
	d = csub(amc, jb);
}
template <int SIGN>
__device__ __forceinline__ float2 mulConst(float2 v, float re, float im) { // v * (re, SIGN<0 ? -im : +im)
	const float s = (SIGN < 0) ? -im : im;
	return make_float2(v.x*re - v.y*s, v.x*s + v.y*re);
}
// 16-point DFT in place; on return X[e + 4c] sits at v[c + 4e]
template <int SIGN>
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
#pragma unroll
	for (int i = 0; i < 4; ++i) dft4<SIGN>(v[i], v[i + 4], v[i + 8], v[i + 12]);
	const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
	// t_i[e] (at v[i + 4e]) *= w16^(i e)
	v[1 + 4] = mulConst<SIGN>(v[1 + 4], c1, s1);   // w^1
	v[1 + 8] = mulConst<SIGN>(v[1 + 8], h, h);     // w^2
	v[1 + 12] = mulConst<SIGN>(v[1 + 12], s1, c1); // w^3
	v[2 + 4] = mulConst<SIGN>(v[2 + 4], h, h);     // w^2
	v[2 + 8] = (SIGN < 0) ? mulNegI(v[2 + 8]) : mulI(v[2 + 8]); // w^4
	v[2 + 12] = mulConst<SIGN>(v[2 + 12], -h, h);  // w^6
	v[3 + 4] = mulConst<SIGN>(v[3 + 4], s1, c1);   // w^3
	v[3 + 8] = mulConst<SIGN>(v[3 + 8], -h, h);    // w^6
	v[3 + 12] = mulConst<SIGN>(v[3 + 12], -c1, -s1); // w^9
#pragma unroll
	for (int e = 0; e < 4; ++e) dft4<SIGN>(v[4*e], v[4*e + 1], v[4*e + 2], v[4*e + 3]);
}
template <int SIGN>
__device__ __forceinline__ void dft3(float2 &a, float2 &b, float2 &c) {
	const float s3 = 0.86602540378443864676f;
	float2 bpc = cadd(b, c), bmc = csub(b, c);
	float2 t = make_float2(a.x - 0.5f*bpc.x, a.y - 0.5f*bpc.y);
	float2 u = cscale((SIGN < 0) ? mulNegI(bmc) : mulI(bmc), s3);
	a = cadd(a, bpc);
	b = cadd(t, u);
	c = csub(t, u);
}
template <int SIGN>
__device__ __forceinline__ void dft5(float2 &a, float2 &b, float2 &c, float2 &d, float2 &e) {
	const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
	const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
	float2 bpe = cadd(b, e), bme = csub(b, e), cpd = cadd(c, d), cmd = csub(c, d);
	float2 t1 = make_float2(a.x + c1*bpe.x + c2*cpd.x, a.y + c1*bpe.y + c2*cpd.y);
	float2 t2 = make_float2(a.x + c2*bpe.x + c1*cpd.x, a.y + c2*bpe.y + c1*cpd.y);
	float2 u1 = make_float2(s1*bme.x + s2*cmd.x, s1*bme.y + s2*cmd.y);
	float2 u2 = make_float2(s2*bme.x - s1*cmd.x, s2*bme.y - s1*cmd.y);
	float2 ju1 = (SIGN < 0) ? mulNegI(u1) : mulI(u1);
	float2 ju2 = (SIGN < 0) ? mulNegI(u2) : mulI(u2);
	a = cadd(a, cadd(bpe, cpd));
	b = cadd(t1, ju1);
	c = cadd(t2, ju2);
	d = csub(t2, ju2);
	e = csub(t1, ju1);
}
// RA-point DFT over v[base + stride*j], j < RA (RA = 2, 4 or 8), in place, outputs in natural order
template <int SIGN, int RA, int N>
__device__ __forceinline__ void dftPow2(float2 (&v)[N], int base, int stride) {
	if constexpr (RA == 2) {
		const float2 a = v[base], b = v[base + stride];
		v[base] = cadd(a, b);
		v[base + stride] = csub(a, b);
	} else if constexpr (RA == 4) {
		dft4<SIGN>(v[base], v[base + stride], v[base + 2*stride], v[base + 3*stride]);
	} else {
		static_assert(RA == 8, "radix 2, 4 or 8");
		// even / odd halves (each a natural-order 4-point DFT), then X[k] = E[k] + w8^k O[k], X[k+4] = E[k] - w8^k O[k]
		float2 e0 = v[base], e1 = v[base + 2*stride], e2 = v[base + 4*stride], e3 = v[base + 6*stride];
		float2 o0 = v[base + stride], o1 = v[base + 3*stride], o2 = v[base + 5*stride], o3 = v[base + 7*stride];
		dft4<SIGN>(e0, e1, e2, e3);
		dft4<SIGN>(o0, o1, o2, o3);
		const float h = 0.70710678118654752440f;
		o1 = mulConst<SIGN>(o1, h, h);
		o2 = (SIGN < 0) ? mulNegI(o2) : mulI(o2);
		o3 = mulConst<SIGN>(o3, -h, h);
		v[base] = cadd(e0, o0); v[base + 4*stride] = csub(e0, o0);
		v[base + stride] = cadd(e1, o1); v[base + 5*stride] = csub(e1, o1);
		v[base + 2*stride] = cadd(e2, o2); v[base + 6*stride] = csub(e2, o2);
		v[base + 3*stride] = cadd(e3, o3); v[base + 7*stride] = csub(e3, o3);
	}
}
// R3-point DFT (R3 = RA*G: RA = 2, 4 or 8 and G = 3 or 5) in place; on return X[e + RA*c] sits at v[c + G*e]
template <int R3> struct LastStage {
	static constexpr int G = (R3%3 == 0) ? 3 : 5;
	static constexpr int RA = R3/G;
	static_assert(R3%G == 0 && (RA == 2 || RA == 4 || RA == 8), "last stage: {2,4,8} x {3,5} points");
};
template <int SIGN, int R3>
__device__ __forceinline__ void dftLast(float2 (&v)[R3]) {
	constexpr int G = LastStage<R3>::G, RA = LastStage<R3>::RA;
#pragma unroll
	for (int i = 0; i < G; ++i) dftPow2<SIGN, RA>(v, i, G);
	// t_i[e] (at v[i + G e]) *= w_R3^(i e)
#pragma unroll
	for (int i = 1; i < G; ++i) {
#pragma unroll
		for (int e = 1; e < RA; ++e) {
			// compile-time constant after unrolling
			const float ang = 6.28318530717958647692f*float(i*e)/float(R3);
			v[i + G*e] = mulConst<SIGN>(v[i + G*e], __builtin_cosf(ang), __builtin_sinf(ang));
		}
	}
#pragma unroll
	for (int e = 0; e < RA; ++e) {
		if constexpr (G == 3) dft3<SIGN>(v[G*e], v[G*e + 1], v[G*e + 2]);
		else dft5<SIGN>(v[G*e], v[G*e + 1], v[G*e + 2], v[G*e + 3 < R3 ? G*e + 3 : 0], v[G*e + 4 < R3 ? G*e + 4 : 0]);
	}
}

// load(idx) -> float2 supplies the natural-order input, store(idx, value) receives the natural-order output.
// lds: H + H/16 float2.  All threads of the block must call this (it synchronises).
// prep(idx) -> any value: called for ALL outputs of a thread before the first store(idx, value, prepared), so whatever it
// loads is in flight together (a load placed inside `store` is not moved above the preceding stores by the compiler).
// LEAN: fewer table bytes per frame (the tables are L2-resident, but 73 KB of them per frame crossed the CU's 64-B/clk vector
// memory path beside 48 KB of data -- ablation in EXPERIMENTS.md: tables held constant took 0.9 ms per step off the analysis and 0.4 off
// the synthesis).  The 15 stage-A twiddles w^n come from six loaded ones (w^1..w^4, w^8, w^12: three 16-byte loads instead of
// eight) and nine products of two of them: one extra rounding each.
struct BlockSync { __device__ __forceinline__ void operator()() const { __syncthreads(); } };
struct NoHook { __device__ __forceinline__ void operator()() const {} };
struct NoArrival { __device__ __forceinline__ float2 operator()(float2 v, int) const { return v; } };
// t / sync: a workgroup may hold several TEAMS of 256 threads, each transforming its own frame in its own `lds` at its own pace;
// t is then the index within the team and sync() the team's barrier.  Hooks of kSynthEmitTeams: lastRead() is called once the last
// stage's operands have left `lds` (a barrier there lets `store` write into the same buffer); requested() once the frame's loads have
// been issued and before anything is written to `lds` (the previous frame's overlap-add runs under the loads' latency), firstWritten()
// after the first stage's writes
template <int SIGN, int R3, bool LEAN, int ROUNDS = 0, typename Load, typename Prep, typename Store, typename Sync = BlockSync, typename LastRead = NoHook,
          typename Requested = NoHook, typename FirstWritten = NoHook, typename Arrived = NoArrival>
__device__ __forceinline__ void fftFast(float2 *lds, const float4 *__restrict__ twA, const float4 *__restrict__ twB, Load load, Prep prep, Store store,
                                        const int t = threadIdx.x, Sync sync = Sync(), LastRead lastRead = LastRead(), Requested requested = Requested(),
                                        FirstWritten firstWritten = FirstWritten(), Arrived arrived = Arrived()) {
	constexpr int MA = 16*R3;
	float2 v[16];
	// stage A: radix 16, stride 1
	if (t < MA) {
#pragma unroll
		for (int k = 0; k < 16; ++k) v[k] = load(t + MA*k, k);
	}
	requested();
	if (t < MA) {
#pragma unroll
		for (int k = 0; k < 16; ++k) v[k] = arrived(v[k], t + MA*k); // (anything `load` did to its value would be waited for in front of requested())
		float2 w[16]; // w[n] = w_H^(n t) (conjugated for the inverse transform)
		if constexpr (LEAN) {
			const float4 a = twA[t], b = twA[MA + t], c = twA[2*MA + t]; // (w1, w2) (w3, w4) (w8, w12)
			const float sg = (SIGN > 0) ? -1.0f : 1.0f;
			w[1] = make_float2(a.x, sg*a.y); w[2] = make_float2(a.z, sg*a.w); w[3] = make_float2(b.x, sg*b.y); w[4] = make_float2(b.z, sg*b.w);
			w[8] = make_float2(c.x, sg*c.y); w[12] = make_float2(c.z, sg*c.w);
#pragma unroll
			for (int hi = 4; hi <= 12; hi += 4) {
#pragma unroll
				for (int lo = 1; lo < 4; ++lo) w[hi + lo] = cmulPlain(w[hi], w[lo]);
			}
		} else {
			float4 wA[8]; // the 15 stage twiddles, two per 16-byte load, in flight during the butterflies
#pragma unroll
			for (int i = 0; i < 8; ++i) wA[i] = twA[i*MA + t];
#pragma unroll
			for (int n = 1; n < 16; ++n) {
				const float4 pr = wA[(n - 1) >> 1];
				w[n] = ((n - 1) & 1) ? make_float2(pr.z, pr.w) : make_float2(pr.x, pr.y);
				if (SIGN > 0) w[n].y = -w[n].y;
			}
		}
		dft16<SIGN>(v);
#pragma unroll
		for (int pos = 0; pos < 16; ++pos) {
			const int n = (pos >> 2) + 4*(pos & 3);
			float2 val = v[pos];
			if (n > 0) val = cmulPlain(val, w[n]);
			lds[17*t + n] = val; // padded index of 16 t + n
		}
	}
	firstWritten();
	sync();
	// stage B: radix 16, stride 16
	const int p = t >> 4, q0 = t & 15;
	if (t < MA) {
#pragma unroll
		for (int k = 0; k < 16; ++k) v[k] = lds[q0 + 17*(p + R3*k)];
	}
	sync();
	if (t < MA) {
		float4 wB[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) wB[i] = twB[i*R3 + p];
		dft16<SIGN>(v);
#pragma unroll
		for (int pos = 0; pos < 16; ++pos) {
			const int n = (pos >> 2) + 4*(pos & 3);
			float2 val = v[pos];
			if (n > 0) {
				const float4 pr = wB[(n - 1) >> 1];
				float2 w = ((n - 1) & 1) ? make_float2(pr.z, pr.w) : make_float2(pr.x, pr.y);
				if (SIGN > 0) w.y = -w.y;
				val = cmulPlain(val, w);
			}
			lds[q0 + 256*p + 16*n] = val;
		}
	}
	sync();
	// stage C: radix R3, stride 256, no twiddles
	if (t < 256) {
		constexpr int G = LastStage<R3>::G, RA = LastStage<R3>::RA;
		float2 u[R3];
#pragma unroll
		for (int k = 0; k < R3; ++k) u[k] = lds[t + 256*k];
		lastRead();
		constexpr int CHUNK = ROUNDS ? R3/ROUNDS : (R3 <= 12 ? R3 : (R3 == 24 ? 6 : 5)); // outputs prepared ahead of their stores (R3 = 20 / 24: four rounds, or the registers cost a wave of occupancy; ROUNDS: the caller's choice)
		decltype(prep(0, 0)) ready[CHUNK];
#pragma unroll
		for (int i = 0; i < CHUNK; ++i) { // the first round's loads fly during the butterflies
			const int e = i/G, c = i - G*e;
			ready[i] = prep(t + 256*(e + RA*c), e + RA*c);
		}
		dftLast<SIGN, R3>(u);
#pragma unroll
		for (int p0 = 0; p0 < R3; p0 += CHUNK) {
			if (p0 > 0) {
#pragma unroll
				for (int i = 0; i < CHUNK; ++i) {
					const int pos = p0 + i, e = pos/G, c = pos - G*e;
					if (pos < R3) ready[i] = prep(t + 256*(e + RA*c), e + RA*c);
				}
			}
#pragma unroll
			for (int i = 0; i < CHUNK; ++i) {
				const int pos = p0 + i, e = pos/G, c = pos - G*e;
				if (pos < R3) store(t + 256*(e + RA*c), u[pos], ready[i], e + RA*c);
			}
		}
	}
}

// (windowPad / analysisWindowInCall: smst_device.h -- the host scheduler evaluates the same condition)
template <int R3, bool LEAN>
__global__ __launch_bounds__(16*R3 > 256 ? 16*R3 : 256) __attribute__((amdgpu_waves_per_eu(4, 4))) void kAnalyseFast(DevBatch d, IoArgs io, int sBase, int hopBase, int lateOnly) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float2 *lds = reinterpret_cast<float2 *>(smemRaw);
	const BlockCoord bc = xcdAwareBlock();
	const int k = bc.x;
	const int c = bc.y >> 1;
	const int which = bc.y & 1;
	const int s = bc.s;
	const HopDesc hd = d.hops[(size_t)(sBase + s)*d.hopStride + hopBase + k];
	if (!(hd.flags & HOP_ACTIVE) || !(hd.flags & HOP_NEW_SPECTRUM)) return;
	if (which == 1 && !(hd.flags & HOP_REANALYSE_PREV)) return;
	if (lateOnly && analysisWindowInCall(d.B, d.M, d.I, hd.inputOffset, which, io.inSamples[sBase + s])) return; // kAnalyseTeams has taken this frame
	const int B = d.B, H = d.M, halfB = B/2, N = d.N;
	const int base = hd.inputOffset - (which ? d.I : 0) - B;
	const float *x = io.in + (size_t)(sBase + s)*io.inStreamStride + (size_t)c*io.inChannelStride;
	const float *hist = d.hist[d.histCur] + ((size_t)(sBase + s)*d.C + c)*(size_t)d.histLen + d.histLen;
	const float2 *__restrict__ winA = d.winA, *__restrict__ winB = d.winB;
	float2 *dst = (which ? d.Xprev : d.Xcur) + rowOf(d, s, k, c);
	auto prep = [](int, int) { return 0; };
	auto store = [&](int j, float2 u, int, int) {
		const int kk = 2*j;
		if (kk < H) dst[kk] = u;
		else dst[N - 1 - kk] = cconj(u);
	};
	constexpr int MA = 16*R3;
	if (base >= 0 && H - halfB == MA && B - halfB == 15*MA) {
		// the usual case (both presets): the whole window lies in this call's input, and the two halves of the packed
		// input change validity exactly at element-slot boundaries: slot 0 has no imaginary part, slot 15 no real part
		const float *x0 = x + base + halfB, *x1 = x + base - H + halfB;
		if constexpr (LEAN) {
			// the folded window (w_re, w_im)(m) * e^{-i pi m/N} as TWO floats per element and the modulation generated: element
			// m = t + MA*slot, and e^{-i pi MA/N} = e^{-i pi/32} whatever the size, so the modulation is halfTw[t] times one of
			// sixteen constants (8 bytes per element instead of 16; six more multiply-adds per element, one more rounding)
			const float2 *__restrict__ win2 = d.win2;
			const float2 hb = d.halfTw[min((int)threadIdx.x, MA - 1)];
			fftFast<-1, R3, true>(lds, d.twA6, d.twB4,
				[&](int m, int slot) {
					const float2 w = win2[m];
					float2 z = make_float2(0.f, 0.f);
					if (slot < 15) z.x = x0[m]*w.x;
					if (slot > 0) z.y = x1[m]*w.y;
					const float ang = 3.14159265358979323846f*float(slot)/32.0f; // compile-time constant after unrolling
					const float2 h = cmulPlain(hb, make_float2(__builtin_cosf(ang), -__builtin_sinf(ang)));
					return cmulPlain(z, h);
				}, prep, store);
			return;
		}
		const float4 *__restrict__ win4 = d.win4;
		fftFast<-1, R3, false>(lds, d.twA4, d.twB4,
			[&](int m, int slot) {
				// same roundings as the general path below: round(xi*b + round(xr*a)), with the absent half an exact zero
				const float4 w = win4[m]; // (winA, winB) in one 16-byte load
				float2 r = make_float2(0.f, 0.f);
				if (slot < 15) { const float xr = x0[m]; r = make_float2(xr*w.x, xr*w.y); }
				if (slot > 0) { const float xi = x1[m]; r = make_float2(fmaf(xi, w.z, r.x), fmaf(xi, w.w, r.y)); }
				return r;
			}, prep, store);
		return;
	}
	// the general case (a window that reaches into the carried history, or a block size whose halves do not fall on slot boundaries):
	// bounds-checked sample fetches, THE SAME ARITHMETIC as the fast case above, element for element -- a hop analysed in a short
	// call (history) and the same hop inside one long call must give the same bits (chunking invariance)
	const float2 hbG = d.halfTw[min((int)threadIdx.x, MA - 1)];
	fftFast<-1, R3, LEAN>(lds, LEAN ? d.twA6 : d.twA4, d.twB4,
		[&](int m, int slot) {
			float xr = 0, xi = 0;
			if (m < B - halfB) { int src = base + m + halfB; xr = (src >= 0) ? x[src] : hist[src]; }
			if (m >= H - halfB) { int src = base + m - H + halfB; xi = (src >= 0) ? x[src] : hist[src]; }
			if constexpr (LEAN) {
				const float2 w = d.win2[m];
				const float2 z = make_float2(xr*w.x, xi*w.y);
				const float ang = 3.14159265358979323846f*float(slot)/32.0f; // compile-time constant after unrolling
				const float2 h = cmulPlain(hbG, make_float2(__builtin_cosf(ang), -__builtin_sinf(ang)));
				return cmulPlain(z, h);
			} else {
				const float2 a = winA[m], b = winB[m];
				return make_float2(fmaf(xi, b.x, xr*a.x), fmaf(xi, b.y, xr*a.y));
			}
		}, prep, store);
}

// ------------------------------------------------------------------------------------------------------
// K1 by persistent TEAMS (the default for the 2560- and 3072-bin geometries; SMST_FFT_TEAMS=0: one frame per workgroup): one workgroup per CU, three teams of 256 threads, each team transforming
// frame after frame AT ITS OWN PACE -- the barriers of a transform are the team's (an LDS counter its four waves spin on), not the
// workgroup's.  The folded window and the stage twiddles (76 KB) are copied to LDS once per workgroup and shared by the teams, so
// per frame only the samples come in and the spectrum goes out (kAnalyseFast pulls 74 KB of tables through L1 per frame).
// Same table values, same operations in the same order: bit-identical to kAnalyseFast.
// ------------------------------------------------------------------------------------------------------
struct TeamSync { // LDS operations of a wave complete in order: a wave's counter increment is seen after its data writes / reads
	volatile int *word;
	int *generation;
	__device__ __forceinline__ void operator()() const {
		asm volatile("" ::: "memory");
		__builtin_amdgcn_wave_barrier(); // (the CPU stand-in runs a wave lane by lane: every lane reaches the barrier before lane 0 signals)
		if ((threadIdx.x & 63) == 0) (void)__hip_atomic_fetch_add(const_cast<int *>(word), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		*generation += 4;
		while (__hip_atomic_load(const_cast<int *>(word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < *generation) __builtin_amdgcn_s_sleep(1);
		__builtin_amdgcn_wave_barrier();
		asm volatile("" ::: "memory");
	}
};

// EXACT: block = 15/16 of the FFT size (windowPad = 0: the geometry is then a compile-time constant); otherwise see windowPad.
// Either way the arithmetic is kAnalyseFast's, element for element (an absent half contributes a zero: sample x zero weight).
template <int R3, int TEAMS, bool EXACT>
__global__ __launch_bounds__(256*TEAMS) void kAnalyseTeams(DevBatch d, IoArgs io, const HopDesc *__restrict__ hopTable, int sBase, int hopBase, int tileHops, int nStreams) {
	static_assert(16*R3 <= 256, "a team is 256 threads");
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	constexpr int MA = 16*R3, H = 256*R3, N = 2*H;
	const int B = EXACT ? 480*R3 : d.B, halfB = B/2; // EXACT: the geometry this kernel is launched for
	const int team = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), t = threadIdx.x & 255, tA = t < MA ? t : 0;
	const int total = tileHops*2*d.C*nStreams;
	float4 *winLds = reinterpret_cast<float4 *>(smemRaw); // (winA, winB) of all elements
	float4 *twALds = winLds + H;                           // first-stage twiddles, [8][MA]
	float4 *twBLds = twALds + 8*MA;                        // second-stage twiddles, [8][R3]
	float2 *lds = reinterpret_cast<float2 *>(twBLds + 8*R3) + (size_t)team*(H + H/16);
	volatile int *words = reinterpret_cast<volatile int *>(reinterpret_cast<float2 *>(twBLds + 8*R3) + (size_t)TEAMS*(H + H/16));
	for (int i = threadIdx.x; i < H; i += blockDim.x) winLds[i] = d.win4[i];
	for (int i = threadIdx.x; i < 8*MA; i += blockDim.x) twALds[i] = d.twA4[i];
	for (int i = threadIdx.x; i < 8*R3; i += blockDim.x) twBLds[i] = d.twB4[i];
	if (threadIdx.x < 16) words[threadIdx.x] = 0;
	__syncthreads();
	int generation = 0;
	const TeamSync sync{words + team, &generation};
	const int stride = gridDim.x*TEAMS;
	for (int lin = blockIdx.x + gridDim.x*team; lin < total; lin += stride) { // the same residue mod 8 for all of a workgroup's teams
		const BlockCoord bc = xcdAwareCoord(lin, tileHops, 2*d.C, nStreams);
		const HopDesc hd = hopTable[(size_t)(sBase + bc.s)*d.hopStride + hopBase + bc.x];
		const int c = bc.y >> 1, which = bc.y & 1;
		if (!(hd.flags & HOP_ACTIVE) || !(hd.flags & HOP_NEW_SPECTRUM) || (which && !(hd.flags & HOP_REANALYSE_PREV)) || !analysisWindowInCall(B, H, d.I, hd.inputOffset, which, io.inSamples[sBase + bc.s])) continue;
		const int base = hd.inputOffset - (which ? d.I : 0) - B;
		const float *x = io.in + (size_t)(sBase + bc.s)*io.inStreamStride + (size_t)c*io.inChannelStride;
		const float *x0 = x + base + halfB, *x1 = x + base - H + halfB;
		float2 *dst = (which ? d.Xprev : d.Xcur) + rowOf(d, bc.s, bc.x, c);
		fftFast<-1, R3, false>(lds, twALds, twBLds,
			[&](int m, int slot) { // kAnalyseFast's roundings: round(xi*b + round(xr*a)), the absent half an exact zero
				const float4 w = winLds[tA + MA*slot];
				float2 r = make_float2(0.f, 0.f);
				if (slot < 15) { const float xr = x0[m]; r = make_float2(xr*w.x, xr*w.y); }
				if (slot > 0) { const float xi = x1[m]; r = make_float2(fmaf(xi, w.z, r.x), fmaf(xi, w.w, r.y)); }
				return r;
			},
			[](int, int) { return 0; },
			[&](int j, float2 u, int, int) {
				const int kk = 2*j;
				if (kk < H) dst[kk] = u;
				else dst[N - 1 - kk] = cconj(u);
			}, t, sync);
		sync(); // the last stage's LDS reads, before the next frame's first-stage writes
	}
}

template <int R3, bool LEAN>
__global__ __launch_bounds__(16*R3 > 256 ? 16*R3 : 256) void kSynthFast(DevBatch d, int sBase, int hopBase) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float2 *lds = reinterpret_cast<float2 *>(smemRaw);
	const int k = blockIdx.x, c = blockIdx.y, s = blockIdx.z;
	const HopDesc hd = d.hops[(size_t)(sBase + s)*d.hopStride + hopBase + k];
	if (!(hd.flags & HOP_ACTIVE)) return;
	const int B = d.B, H = d.M, N = d.N, halfB = B/2;
	const float2 *X = d.OUT + rowOf(d, s, k, c);
	float *__restrict__ frame = d.frames + ((size_t)((size_t)s*d.T + k)*d.C + c)*(size_t)B;
	auto loadBin = [&](int j, int) {
		// one load at a selected address, conjugated afterwards: with a load in each arm of the conditional the
		// compiler emitted 16 loads each followed by s_waitcnt vmcnt(0) -- sixteen memory round trips per FFT
		const int kk = 2*j;
		const bool upper = kk >= H;
		float2 v = X[upper ? N - 1 - kk : kk];
		if (upper) v.y = -v.y;
		return v;
	};
	if constexpr (LEAN) {
		// an output needs e^{+i pi m/N} and its two window samples: the windows as one 8-byte load, the modulation generated --
		// m = t + 256 j, so it is halfTw[t] times one of R3 constants e^{-i pi j/(2 R3)} (conjugated in the product below)
		const float2 *__restrict__ syn2 = d.syn2;
		const float2 hb = d.halfTw[min((int)threadIdx.x, 255)];
		fftFast<+1, R3, true>(lds, d.twA6, d.twB4, loadBin,
			[&](int m, int) { return syn2[m]; },
			[&](int m, float2 u, float2 w, int j) {
				const float ang = 3.14159265358979323846f*float(j)/float(2*R3); // compile-time constant after unrolling
				const float2 h = cmulPlain(hb, make_float2(__builtin_cosf(ang), -__builtin_sinf(ang)));
				const float2 v = cmulcPlain(u, h); // * e^{+i pi m / N}
				if (m < B - halfB) frame[m + halfB] = (2*v.x)*w.x;
				if (m >= H - halfB) frame[m - H + halfB] = (2*v.y)*w.y;
			});
		return;
	}
	const float4 *__restrict__ synTab = d.synTab;
	fftFast<+1, R3, false>(lds, d.twA4, d.twB4,
		[&](int j, int) {
			// one load at a selected address, conjugated afterwards: with a load in each arm of the conditional the
			// compiler emitted 16 loads each followed by s_waitcnt vmcnt(0) -- sixteen memory round trips per FFT
			const int kk = 2*j;
			const bool upper = kk >= H;
			float2 v = X[upper ? N - 1 - kk : kk];
			if (upper) v.y = -v.y;
			return v;
		},
		[&](int m, int) { // everything an output needs from memory in ONE 16-byte load (twiddle + its two window samples),
			// requested for all of a thread's outputs before the first store
			return synTab[m];
		},
		[&](int m, float2 u, float4 r, int) {
			const float2 v = cmulcPlain(u, make_float2(r.x, r.y)); // * e^{+i pi m / N}
			if (m < B - halfB) frame[m + halfB] = (2*v.x)*r.z;
			if (m >= H - halfB) frame[m - H + halfB] = (2*v.y)*r.w;
		});
}

// K4a by persistent teams (see kAnalyseTeams): kSynthFast's transform with kAnalyseTeams' organisation -- the (twiddle, window) table of
// the outputs and the stage twiddles sit in LDS, a team synthesises frame after frame at its own pace.  Bit-identical to kSynthFast.
template <int R3, int TEAMS>
__global__ __launch_bounds__(256*TEAMS) void kSynthTeams(DevBatch d, const HopDesc *__restrict__ hopTable, int sBase, int hopBase, int tileHops, int nStreams) {
	static_assert(16*R3 <= 256, "a team is 256 threads");
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	constexpr int MA = 16*R3, H = 256*R3, N = 2*H;
	const int team = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), t = threadIdx.x & 255;
	const int B = d.B, halfB = B/2;
	const int total = tileHops*d.C*nStreams;
	float4 *synLds = reinterpret_cast<float4 *>(smemRaw); // (e^{+i pi m/N}, the two window samples of output m)
	float4 *twALds = synLds + H;
	float4 *twBLds = twALds + 8*MA;
	float2 *lds = reinterpret_cast<float2 *>(twBLds + 8*R3) + (size_t)team*(H + H/16);
	volatile int *words = reinterpret_cast<volatile int *>(reinterpret_cast<float2 *>(twBLds + 8*R3) + (size_t)TEAMS*(H + H/16));
	for (int i = threadIdx.x; i < H; i += blockDim.x) synLds[i] = d.synTab[i];
	for (int i = threadIdx.x; i < 8*MA; i += blockDim.x) twALds[i] = d.twA4[i];
	for (int i = threadIdx.x; i < 8*R3; i += blockDim.x) twBLds[i] = d.twB4[i];
	if (threadIdx.x < 16) words[threadIdx.x] = 0;
	__syncthreads();
	int generation = 0;
	const TeamSync sync{words + team, &generation};
	const int stride = gridDim.x*TEAMS;
	for (int lin = blockIdx.x + gridDim.x*team; lin < total; lin += stride) {
		const BlockCoord bc = xcdAwareCoord(lin, tileHops, d.C, nStreams); // x: hop, y: channel
		const HopDesc hd = hopTable[(size_t)(sBase + bc.s)*d.hopStride + hopBase + bc.x];
		if (!(hd.flags & HOP_ACTIVE)) continue;
		const float2 *X = d.OUT + rowOf(d, bc.s, bc.x, bc.y);
		float *__restrict__ frame = d.frames + ((size_t)((size_t)bc.s*d.T + bc.x)*d.C + bc.y)*(size_t)B;
		fftFast<+1, R3, false>(lds, twALds, twBLds,
			[&](int j, int) { // one load at a selected address, conjugated afterwards (see kSynthFast)
				const int kk = 2*j;
				const bool upper = kk >= H;
				float2 v = X[upper ? N - 1 - kk : kk];
				if (upper) v.y = -v.y;
				return v;
			},
			[&](int m, int) { return synLds[m]; },
			[&](int m, float2 u, float4 r, int) {
				const float2 v = cmulcPlain(u, make_float2(r.x, r.y)); // * e^{+i pi m / N}
				if (m < B - halfB) frame[m + halfB] = (2*v.x)*r.z;
				if (m >= H - halfB) frame[m - H + halfB] = (2*v.y)*r.w;
			}, t, sync);
		sync(); // the last stage's LDS reads, before the next frame's first-stage writes
	}
}

// The window-product sum under output sample i of a tile (i counted from the tile's first sample): what the carry holds there, then
// the covering frames' products oldest first -- kEmit's sum, term for term.
__device__ __forceinline__ float windowProductAt(const DevBatch &d, const EmitDesc &ed, const float *__restrict__ carryWpOld, int i) {
	float wp = i < d.carryLen ? carryWpOld[i] : 1e-30f;
	if (ed.hopCount > 0) {
		const int rel = ed.nLo + i - ed.firstHopPos - d.delta, B = d.B, I = d.I;
		int qHi = rel >= 0 ? rel/I : -1;
		const int qLo = (rel - B + 1 > 0) ? (rel - B + 1 + I - 1)/I : 0;
		if (qHi > ed.hopCount - 1) qHi = ed.hopCount - 1;
		for (int q = qLo; q <= qHi; ++q) wp += d.wprod[rel - q*I];
	}
	return wp;
}
// kSynthEmitTeams' companion: the window products do not depend on the channel or on the data, and from the sample on that the carried
// sums no longer reach they repeat with the interval -- the teams keep that steady pattern in registers.  What is left is per stream: the
// products under the tile's first wpHeadLen samples (-> wpHead) and under what the next tile starts from (-> the new carry).
__global__ __launch_bounds__(256) void kEmitProducts(DevBatch d, int sBase, int tileIndex) {
	const int sg = sBase + blockIdx.y;
	const EmitDesc ed = d.emit[(size_t)sg*d.emitStride + tileIndex];
	const int span = ed.nHi - ed.nLo, CL = d.carryLen;
	const float *carryWpOld = d.carryWp[d.carryCur] + (size_t)sg*CL;
	const int i = blockIdx.x*blockDim.x + threadIdx.x;
	if (i < d.wpHeadLen) d.wpHead[(size_t)sg*d.wpHeadLen + i] = windowProductAt(d, ed, carryWpOld, i);
	else if (i - d.wpHeadLen < CL) d.carryWp[d.carryCur ^ 1][(size_t)sg*CL + (i - d.wpHeadLen)] = windowProductAt(d, ed, carryWpOld, span + (i - d.wpHeadLen));
}

// K4a + K4b in one kernel: synthesis, overlap-add, window-product normalisation and emission (signalsmith-stretch.h:397-415) without
// the frames ever reaching HBM.  A team of 256 threads takes ONE (stream, channel) of the tile and synthesises its hops in order;
// the overlap-add ring (the carried partial sums in front, NI = QN + 1 intervals) lives in the team's registers -- thread t owns
// the positions r = t + 256*slot of every interval.  Per hop: kSynthTeams' transform; its windowed outputs go to the team's own
// transform buffer (free once the last stage has read it) instead of HBM; each thread adds its positions of the QN intervals the frame
// covers, oldest frame first as kEmit (and the reference's ring) sums them; the interval that no later frame reaches is divided by
// its window products (kEmitProducts) and stored; the ring moves on by one interval.  The previous frame's additions run while the
// next frame's spectrum is on its way in.  Two teams per workgroup (the ring takes NI*SLOTS registers on top of the transform's), one
// workgroup per CU.  Same operations in the same order on the same values as kSynthTeams + kEmit: bit-identical output and carry
// (test_synth_emit_equals_two_kernels).
template <int R3, int QN, int SLOTS, bool SPLIT>
__global__ __launch_bounds__(512) void kSynthEmitTeams(DevBatch d, IoArgs io, int sBase, int tileIndex, int nStreams) {
	constexpr int TEAMS = 2, MA = 16*R3, H = 256*R3, N = 2*H, NI = QN + 1, DQ = SPLIT ? 1 : 0;
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	const int team = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), t = threadIdx.x & 255;
	const int B = d.B, halfB = B/2, I = d.I, CL = d.carryLen;
	float4 *synLds = reinterpret_cast<float4 *>(smemRaw); // (e^{+i pi m/N}, the two window samples of output m)
	float4 *twALds = synLds + H;
	float4 *twBLds = twALds + 8*MA;
	float2 *lds = reinterpret_cast<float2 *>(twBLds + 8*R3) + (size_t)team*(H + H/16);
	volatile int *words = reinterpret_cast<volatile int *>(reinterpret_cast<float2 *>(twBLds + 8*R3) + (size_t)TEAMS*(H + H/16));
	for (int i = threadIdx.x; i < H; i += blockDim.x) synLds[i] = d.synTab[i];
	for (int i = threadIdx.x; i < 8*MA; i += blockDim.x) twALds[i] = d.twA4[i];
	for (int i = threadIdx.x; i < 8*R3; i += blockDim.x) twBLds[i] = d.twB4[i];
	if (threadIdx.x < 16) words[threadIdx.x] = 0;
	__syncthreads();
	int generation = 0;
	const TeamSync sync{words + team, &generation};
	float *ex = reinterpret_cast<float *>(lds); // the frame, B floats, in the transform buffer
	float steady[SLOTS]; // the window products under a sample that only this tile's frames cover: oldest frame first, as the ring sums them
#pragma unroll
	for (int slot = 0; slot < SLOTS; ++slot) {
		const int r = 256*slot + t;
		steady[slot] = 1e-30f;
#pragma unroll
		for (int a = QN - 1; a >= 0; --a) {
			if (r < I && a*I + r < B) steady[slot] += d.wprod[a*I + r];
		}
		keepUnconditional(steady[slot]);
	}
	const int items = nStreams*d.C;
	for (int item = blockIdx.x*TEAMS + team; item < items; item += gridDim.x*TEAMS) {
		const int s = item/d.C, c = item - s*d.C, sg = sBase + s;
		const EmitDesc ed = d.emit[(size_t)sg*d.emitStride + tileIndex];
		const int cnt = ed.hopCount, span = ed.nHi - ed.nLo;
		const size_t carryRow = ((size_t)sg*d.C + c)*(size_t)CL;
		const float *wpOld = d.carryWp[d.carryCur] + (size_t)sg*CL;
		const float *wpHead = d.wpHead + (size_t)sg*d.wpHeadLen; // (kEmitProducts; it also writes the new carry's products)
		float *out = io.out + (size_t)sg*io.outStreamStride + (size_t)c*io.outChannelStride;
		auto carryAt = [&](int i) { return i < CL ? loadCarrySum(d, d.carryCur, carryRow + i) : 0.0f; };
		auto wpAt = [&](int i) { return i < CL ? wpOld[i] : 1e-30f; };
		auto place = [&](int n, float sum, float wp) { // output sample n of the call: final, or part of what the next tile starts from
			if (n < ed.nHi) out[n] = sum/wp;
			else if (n - ed.nHi < CL) storeCarrySum(d, d.carryCur ^ 1, carryRow + (n - ed.nHi), sum);
		};
		if (cnt == 0) { // nothing synthesised for this stream in this tile: the carried sums are emitted / move up
			for (int i = t; i < span + CL; i += 256) place(ed.nLo + i, carryAt(i), wpAt(i));
			continue;
		}
		const int off = ed.firstHopPos - ed.nLo; // samples in front of the tile's first hop (the first tile of a call only)
		for (int i = t; i < off; i += 256) place(ed.nLo + i, carryAt(i), wpAt(i));
		float acc[NI][SLOTS];
#pragma unroll
		for (int u = 0; u < NI; ++u) {
#pragma unroll
			for (int slot = 0; slot < SLOTS; ++slot) {
				const int r = 256*slot + t, i = off + u*I + r;
				acc[u][slot] = r < I ? carryAt(i) : 0.0f;
			}
		}
		// (waited for HERE: a register that is still "being loaded" when the hop loop is entered makes the compiler wait for ALL loads at
		// its first use inside the loop -- i.e. for the spectrum that was requested just before)
#pragma unroll
		for (int u = 0; u < NI; ++u) {
#pragma unroll
			for (int slot = 0; slot < SLOTS; ++slot) keepUnconditional(acc[u][slot]);
		}
		// frame q - 1 is added to the ring while frame q's spectrum is on its way in, and its finished interval leaves after frame
		// q's first-stage writes (no store between a load and its use: vmcnt counts both)
		auto overlapAdd = [&]() {
			float e[QN][SLOTS]; // all reads in flight together, no branch around any of them
			int tt = t;
			keepUnconditional(tt); // (the 24 addresses are formed here, per hop: held across the loop they cost the registers the next spectrum needs)
#pragma unroll
			for (int a = 0; a < QN; ++a) {
#pragma unroll
				for (int slot = 0; slot < SLOTS; ++slot) {
					const int r = 256*slot + tt, i = a*I + r;
					e[a][slot] = ex[r < I && i < B ? i : 0];
				}
			}
#pragma unroll
			for (int a = 0; a < QN; ++a) {
#pragma unroll
				for (int slot = 0; slot < SLOTS; ++slot) {
					const int r = 256*slot + t, i = a*I + r;
					keepUnconditional(e[a][slot]);
					acc[a + DQ][slot] = (r < I && i < B) ? acc[a + DQ][slot] + e[a][slot] : acc[a + DQ][slot];
				}
			}
		};
		auto emitInterval = [&](int q) {
			const int n0 = ed.firstHopPos + q*I;
			float wp[SLOTS];
#pragma unroll
			for (int slot = 0; slot < SLOTS; ++slot) wp[slot] = steady[slot];
			if (q < NI) { // the carried sums may reach into this interval (loads of the kernel's own here too: see `next`)
				Async4 head[SLOTS];
#pragma unroll
				for (int slot = 0; slot < SLOTS; ++slot) {
					const int r = 256*slot + t;
					asyncLoad4(head[slot], wpHead + (r < I ? off + q*I + r : 0));
				}
				asyncWait<0>();
#pragma unroll
				for (int slot = 0; slot < SLOTS; ++slot) {
					asyncArrived(head[slot]);
					if (256*slot + t < I) wp[slot] = asyncValue(head[slot]);
				}
			}
			if (n0 + I <= ed.nHi) { // (all but a call's last hop)
#pragma unroll
				for (int slot = 0; slot < SLOTS; ++slot) {
					const int r = 256*slot + t;
					if (r < I) out[n0 + r] = acc[0][slot]/wp[slot];
				}
			} else {
#pragma unroll
				for (int slot = 0; slot < SLOTS; ++slot) {
					const int r = 256*slot + t;
					if (r < I) place(n0 + r, acc[0][slot], wp[slot]);
				}
			}
#pragma unroll
			for (int u = 0; u + 1 < NI; ++u) {
#pragma unroll
				for (int slot = 0; slot < SLOTS; ++slot) acc[u][slot] = acc[u + 1][slot];
			}
#pragma unroll
			for (int slot = 0; slot < SLOTS; ++slot) acc[NI - 1][slot] = 0.0f;
		};
		// frame q + 1's spectrum is requested as soon as frame q's first stage has taken its own out of the registers
		// (loads the kernel waits for itself, smst_async.h: a compiler-tracked load whose value crosses the loop's back-edge makes the
		// compiler wait for EVERYTHING at the next barrier's counter update, i.e. right behind the request)
		Async8 next[16];
#pragma unroll
		for (int k = 0; k < 16; ++k) asyncClear(next[k]);
		auto request = [&](int q) {
			const float2 *X = d.OUT + rowOf(d, s, q, c);
			if (t < MA) {
#pragma unroll
				for (int k = 0; k < 16; ++k) { const int kk = 2*(t + MA*k); asyncLoad8(next[k], X + (kk >= H ? N - 1 - kk : kk)); } // one load at a selected address (see kSynthFast) ...
			}
		};
		auto landed = [&]() { // at the END of a hop, in front of the back-edge: nothing of the compiler's may touch a register in flight
			asyncWait<0>();
#pragma unroll
			for (int k = 0; k < 16; ++k) asyncArrived(next[k]);
		};
		request(0);
		landed();
		for (int q = 0; q < cnt; ++q) {
			fftFast<+1, R3, false, 2>(lds, twALds, twBLds, // (two rounds of prepared outputs: the ring and the next spectrum need the registers)
				[&](int, int k) { return asyncValue(next[k]); },
				[&](int m, int) { return synLds[m]; },
				[&](int m, float2 u, float4 r, int) {
					const float2 v = cmulcPlain(u, make_float2(r.x, r.y)); // * e^{+i pi m / N}
					if (m < B - halfB) ex[m + halfB] = (2*v.x)*r.z;
					if (m >= H - halfB) ex[m - H + halfB] = (2*v.y)*r.w;
				}, t, sync, sync,
				[&]() {
					if (q > 0) overlapAdd();
					sync(); // the previous frame has been read (or the previous item's last one), before this transform's first-stage writes
				},
				[&]() {
					if (q > 0) emitInterval(q - 1);
					request(q + 1 < cnt ? q + 1 : q);
				},
				[&](float2 v, int j) { if (2*j >= H) v.y = -v.y; return v; }); // ... conjugated once it is there
			sync(); // the frame is complete
			landed();
		}
		overlapAdd();
		sync();
		emitInterval(cnt - 1);
		const int n0 = ed.firstHopPos + cnt*I; // what the ring still holds: the start of the next tile's sums
#pragma unroll
		for (int u = 0; u < NI; ++u) {
#pragma unroll
			for (int slot = 0; slot < SLOTS; ++slot) {
				const int r = 256*slot + t;
				if (r < I) place(n0 + u*I + r, acc[u][slot], 1.0f); // (all of it behind the call's last final sample)
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------------
// K5: per-stream input energy (silence gate, signalsmith-stretch.h:231-238)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kEnergy(DevBatch d, IoArgs io, int sBase, float *__restrict__ energyOut) {
	// grid (stream, part): partial sums, the host adds the kEnergyParts partials of a stream
	const int s = blockIdx.x, part = blockIdx.y, parts = gridDim.y;
	const int n = io.inSamples[sBase + s];
	const int lo = (int)((long long)n*part/parts), hi = (int)((long long)n*(part + 1)/parts);
	float acc = 0;
	for (int c = 0; c < d.C; ++c) {
		const float *x = io.in + (size_t)(sBase + s)*io.inStreamStride + (size_t)c*io.inChannelStride;
		for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
			float v = x[i];
			acc += v*v;
		}
	}
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float *red = reinterpret_cast<float *>(smemRaw);
	red[threadIdx.x] = acc;
	__syncthreads();
	for (int w = 128; w > 0; w >>= 1) {
		if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
		__syncthreads();
	}
	if (threadIdx.x == 0) energyOut[(size_t)(sBase + s)*parts + part] = red[0];
}

// ------------------------------------------------------------------------------------------------------
// K1: analysis.  One workgroup per (hop, channel x {current, previous}, stream):
// gather B samples ending at the hop's input offset from the contiguous per-channel sample block (or the
// carried history for negative indices), multiply by the analysis window, fold into the N/2-point complex
// sequence of the half-bin-shifted real FFT, Stockham FFT in LDS, write the M = N/2 bins.
// Replaces stft.analyseStep at signalsmith-stretch.h:337,:359 (+ the copies :344-350,:366-372).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kAnalyse(DevBatch d, IoArgs io, int sBase, int hopBase) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float2 *bufA = reinterpret_cast<float2 *>(smemRaw);
	float2 *bufB = bufA + d.M;

	const BlockCoord bc = xcdAwareBlock();
	const int k = bc.x;
	const int c = bc.y >> 1;
	const int which = bc.y & 1; // 0: current window, 1: window one interval earlier
	const int s = bc.s;
	const HopDesc hd = d.hops[(size_t)(sBase + s)*d.hopStride + hopBase + k];
	if (!(hd.flags & HOP_ACTIVE) || !(hd.flags & HOP_NEW_SPECTRUM)) return;
	if (which == 1 && !(hd.flags & HOP_REANALYSE_PREV)) return;

	const int B = d.B, H = d.M, halfB = B/2;
	const int base = hd.inputOffset - (which ? d.I : 0) - B; // index of block element 0 in the call's input
	const float *x = io.in + (size_t)(sBase + s)*io.inStreamStride + (size_t)c*io.inChannelStride;
	const float *hist = d.hist[d.histCur] + ((size_t)(sBase + s)*d.C + c)*(size_t)d.histLen;
	const float *__restrict__ win = d.window;

	for (int m = threadIdx.x; m < H; m += blockDim.x) {
		float re = 0, im = 0;
		if (m < B - halfB) {
			int i = m + halfB;
			int src = base + i;
			float v = (src >= 0) ? x[src] : hist[d.histLen + src];
			re = v*win[i];
		}
		if (m >= H - halfB) {
			int i = m - H + halfB;
			int src = base + i;
			float v = (src >= 0) ? x[src] : hist[d.histLen + src];
			im = v*win[i];
		}
		bufA[m] = cmulPlain(make_float2(re, im), d.halfTw[m]);
	}
	__syncthreads();
	float2 *res = fftLds<-1>(bufA, bufB, d.plan, d.twH);

	float2 *dst = (which ? d.Xprev : d.Xcur) + rowOf(d, s, k, c);
	const int N = d.N;
	for (int j = threadIdx.x; j < H; j += blockDim.x) {
		float2 u = res[j];
		int kk = 2*j;
		if (kk < H) dst[kk] = u;
		else dst[N - 1 - kk] = cconj(u);
	}
}

// ------------------------------------------------------------------------------------------------------
// Row lookup shared by the feed-forward kernels
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ const float2 *inputRow(const DevBatch &d, const HopDesc &hd, int s, int sGlobal, int c) {
	return (hd.inSrc >= 0) ? d.Xcur + rowOf(d, s, hd.inSrc, c) : d.stInput + stateRow(d, sGlobal, c);
}
__device__ __forceinline__ const float2 *prevRow(const DevBatch &d, const HopDesc &hd, int s, int k, int sGlobal, int c) {
	const float2 *fromTile = ((hd.prevSrc >= 0) ? d.Xcur : d.Xprev) + rowOf(d, s, (hd.prevSrc >= 0) ? hd.prevSrc : k, c);
	return (hd.prevSrc >= 0 || hd.prevSrc == SRC_REANALYSED) ? fromTile : d.stPrev + stateRow(d, sGlobal, c);
}

__device__ __forceinline__ float mapFreqDev(const DevBatch &d, const StreamParams &p, int sGlobal, float freq) { // :850-856
	if (p.hasCustomMap) {
		const float *t = d.mapTable + (size_t)sGlobal*d.mapTableLen; // row pitch: the longest table of the batch
		const int n = p.mapLen;                                      // this stream's own knots
		float pos = freq*2*float(n) - 0.5f;
		if (pos <= 0) return t[0] + (t[1] - t[0])*pos;
		if (pos >= n - 1) return t[n - 1] + (t[n - 1] - t[n - 2])*(pos - (n - 1));
		int lo = (int)floorf(pos);
		float fr = pos - lo;
		return t[lo] + (t[lo + 1] - t[lo])*fr;
	}
	if (freq > p.freqTonalityLimit) return freq + (p.freqMultiplier - 1)*p.freqTonalityLimit;
	return freq*p.freqMultiplier;
}

// ------------------------------------------------------------------------------------------------------
// K2b-e: channel-summed energy, 4-pass one-pole smoothing, peak centroids, output map, formant envelope and ratio
// (signalsmith-stretch.h:818-848, :859-880, :882-917, :929-966, :972-1036).  All of these are recurrences over the
// bin index; the reference's rounding is kept by evaluating them serially IN THE REFERENCE'S ORDER -- the
// parallel axis is (stream, hop): one wave per stream, lane k = hop k of the tile, scratch arrays laid out
// [bin][64 hops] so that every serial step of the wave is one coalesced 256-byte access.
//   kFeedEnergy : energyT[s][b][k] = sum_c |input_c[b]|^2           (parallel, LDS-transposed)
//   kFeedSerial : everything else, lanes = hops
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kFeedEnergy(DevBatch d, int sBase, int hopBase) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float *tile = reinterpret_cast<float *>(smemRaw); // [64 hops][65]
	const int s = blockIdx.y, sg = sBase + s;
	const int b0 = blockIdx.x*64;
	const int nh = d.nHops[s];
	if (nh == 0) return;
	const int M = d.M, C = d.C;
	const int lane = threadIdx.x & 63, rq = threadIdx.x >> 6;
	for (int k = rq; k < 64; k += 4) {
		float e = 0;
		const int b = b0 + lane;
		if (k < nh && b < M) {
			const HopDesc hd = d.hops[(size_t)sg*d.hopStride + hopBase + k];
			if (hd.flags & (HOP_MAPPED | HOP_FORMANTS)) {
				for (int c = 0; c < C; ++c) e += cnorm(inputRow(d, hd, s, sg, c)[b]);
			}
		}
		tile[k*65 + lane] = e;
	}
	__syncthreads();
	float *eT = d.energyT + (size_t)s*M*64;
	for (int bq = rq; bq < 64; bq += 4) {
		const int b = b0 + bq;
		if (b < M) eT[(size_t)b*64 + lane] = tile[lane*65 + bq];
	}
}

// One serial pass over the M bins of a [bin][64]-strided column, software-pipelined: the 16 loads of a chunk are
// independent of the recurrence and are issued together, the recurrence then runs on registers.  step(e, x) -> e.
template <bool DOWN, typename F>
__device__ __forceinline__ float serialPass(const float *src, float *dst, int M, float e, F step) {
	// double-buffered: chunk c+1 is fetched before chunk c is reduced, so no memory round trip sits between chunks
	// (src may equal dst: the in-place passes only ever overwrite elements that were already fetched)
	constexpr int U = 16;
	float cur[U], nxt[U];
	auto fetch = [&](float (&x)[U], int c0) {
#pragma unroll
		for (int i = 0; i < U; ++i) {
			const int b = DOWN ? (M - 1 - c0 - i) : (c0 + i);
			const int bc = min(max(b, 0), M - 1);
			x[i] = src[(size_t)bc*64];
		}
	};
	fetch(cur, 0);
	for (int c0 = 0; c0 < M; c0 += U) {
		fetch(nxt, c0 + U); // clamped: past the end it re-reads the last element, which is never used
#pragma unroll
		for (int i = 0; i < U; ++i) {
			const int b = DOWN ? (M - 1 - c0 - i) : (c0 + i);
			if (b >= 0 && b < M) {
				e = step(e, cur[i]);
				dst[(size_t)b*64] = e;
			}
		}
#pragma unroll
		for (int i = 0; i < U; ++i) cur[i] = nxt[i];
	}
	return e;
}

__global__ __launch_bounds__(64) void kFeedSerial(DevBatch d, int sBase, int hopBase) {
	const int s = blockIdx.x, sg = sBase + s, k = threadIdx.x;
	const int nh = d.nHops[s];
	if (nh == 0) return;
	const int M = d.M;
	const float Nf = float(d.N);
	const HopDesc hd = d.hops[(size_t)sg*d.hopStride + hopBase + (k < nh ? k : 0)];
	const bool active = k < nh;
	const bool mapped = active && (hd.flags & HOP_MAPPED), formants = active && (hd.flags & HOP_FORMANTS);
	const StreamParams prm = d.params[sg];
	const float *eT = d.energyT + (size_t)s*M*64 + k;
	float *sT = d.smoothT + (size_t)s*M*64 + k;
	float2 *pk = d.peaksT + (size_t)s*(M/2 + 2)*64 + k;

	if (__any(mapped)) {
		// smoothEnergy: (down, up) x 2 with the state carried through, :837-847 (first pass reads the energy)
		const float smoothingBins = Nf/float(d.I);
		const float slew = 1/(1 + smoothingBins*0.5f);
		float e = 0;
		auto pole = [slew](float acc, float x) { return acc + (x - acc)*slew; };
		e = serialPass<true>(eT, sT, M, e, pole);
		e = serialPass<false>(sT, sT, M, e, pole);
		e = serialPass<true>(sT, sT, M, e, pole);
		e = serialPass<false>(sT, sT, M, e, pole);
		// findPeaks, :859-880: maximal runs with energy > smoothed, centroid, mapped centre
		int nPeaks = 0;
		bool inRun = false;
		float bandSum = 0, energySum = 0;
		for (int c0 = 0; c0 <= M; c0 += 16) {
			float en16[16], sm16[16];
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int b = c0 + i;
				en16[i] = (b < M) ? eT[(size_t)b*64] : 0.0f;
				sm16[i] = (b < M) ? sT[(size_t)b*64] : 0.0f;
			}
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int b = c0 + i;
				if (b > M) break;
				const float en = en16[i];
				const bool above = (b < M) && en > sm16[i];
				if (above) {
					if (!inRun) { bandSum = 0; energySum = 0; inRun = true; }
					bandSum += b*en;
					energySum += en;
				} else if (inRun) {
					inRun = false;
					const float avgBand = bandSum/energySum;
					const float avgFreq = (avgBand + 0.5f)/Nf;
					if (mapped) pk[(size_t)nPeaks*64] = make_float2(avgBand, mapFreqDev(d, prm, sg, avgFreq)*Nf - 0.5f);
					++nPeaks;
				}
			}
		}
		// updateOutputMap, :882-917: segment rules reproduce the reference's write order (top segment written last)
		if (mapped) {
			float2 *mapRow = d.map + ((size_t)s*d.T + k)*M;
			const float2 first = nPeaks > 0 ? pk[0] : make_float2(0.f, 0.f);
			const float2 lastP = nPeaks > 0 ? pk[(size_t)(nPeaks - 1)*64] : make_float2(0.f, 0.f);
			const int topStart = max(0, (int)lastP.y), bottomEnd = min(M, (int)ceilf(first.y));
			int p = 1;
			float2 prev = first, next = nPeaks > 1 ? pk[64] : first;
			for (int b = 0; b < M; ++b) {
				float2 mp = make_float2(float(b), 1.0f);
				if (nPeaks > 0) {
					if (b >= topStart) {
						mp = make_float2(b + (lastP.x - lastP.y), 1.0f);
					} else if (b < bottomEnd) {
						mp = make_float2(b + (first.x - first.y), 1.0f);
					} else if (nPeaks >= 2) {
						// largest p in [1, nPeaks) with ceil(peaks[p-1].out) <= b
						while (p + 1 < nPeaks && max(0, (int)ceilf(next.y)) <= b) {
							++p;
							prev = next;
							next = pk[(size_t)p*64];
						}
						if (b < min(M, (int)ceilf(next.y))) {
							float rangeScale = 1/(next.y - prev.y);
							float outOffset = prev.x - prev.y;
							float outScale = next.x - next.y - prev.x + prev.y;
							float gradScale = outScale*rangeScale;
							float r = (b - prev.y)*rangeScale;
							float h = r*r*(3 - 2*r);
							float outB = b + outOffset + h*outScale;
							float gradH = 6*r*(1 - r);
							mp = make_float2(outB, 1 + gradH*gradScale);
						} // else: not covered by any segment (non-monotonic map only): identity, see DESIGN.md
					}
				}
				mapRow[b] = mp;
			}
		}
	}

	if (__any(formants)) {
		// updateFormants, :972-1036.  The metric is the channel-summed energy (:974-980).
		const bool autoBase = formants && prm.formantBaseFreq <= 0;
		float pw = 0, ww = 0;
		if (__any(autoBase)) { // estimateFrequency() raw part, :929-960
			int p0 = 0, p1 = 0, p2 = 0;
			float e0 = eT[0], e1 = eT[0], e2 = eT[0]; // metric at p0, p1, p2
			float em = eT[0], ec = eT[64];             // metric at b-1, b
			for (int b = 1; b < M - 1; ++b) {
				const float en = eT[(size_t)(b + 1)*64];
				const float e = ec;
				if (!(e < em || e <= en)) {
					if (e > e0) {
						if (e > e1) {
							if (e > e2) { p0 = p1; e0 = e1; p1 = p2; e1 = e2; p2 = b; e2 = e; }
							else { p0 = p1; e0 = e1; p1 = b; e1 = e; }
						} else {
							p0 = b; e0 = e;
						}
					}
				}
				em = ec;
				ec = en;
			}
			int peakEstimate = p2;
			if (e1 > e2*0.1f) {
				int diff = abs(peakEstimate - p1);
				if (diff > peakEstimate/8 && diff < peakEstimate*7/8) peakEstimate = peakEstimate%diff;
				if (e0 > e2*0.01f) {
					int diff2 = abs(peakEstimate - p0);
					if (diff2 > peakEstimate/8 && diff2 < peakEstimate*7/8) peakEstimate = peakEstimate%diff2;
				}
			}
			pw = peakEstimate*e2;
			ww = e2;
			if (autoBase) {
				d.est[((size_t)s*d.T + k)*2] = pw;
				d.est[((size_t)s*d.T + k)*2 + 1] = ww;
			}
		}
		float freqEstimate = prm.formantBaseFreq*Nf - 0.5f; // freqToBand, :982
		if (__any(autoBase)) { // :962-965 -- the estimate is smoothed from hop to hop: replay the hops of the tile in order
			float w = d.stFreq[2*sg], wt = d.stFreq[2*sg + 1];
			float mine = 0;
			for (int j = 0; j < 64; ++j) {
				const float pwj = __shfl(pw, j), wwj = __shfl(ww, j);
				const int on = __shfl((int)autoBase, j);
				if (on) {
					w += (pwj - w)*0.25f;
					wt += (wwj - wt)*0.25f;
				}
				if (j == k) mine = w/(wt + 1e-30f);
			}
			if (autoBase) freqEstimate = mine;
		}
		float decay = 1 - 1/(freqEstimate*0.5f + 1);
		float e = 0;
		// max-decay passes (first one reads the metric), then min-grow passes, state carried throughout
		{
			const float dk = decay;
			auto maxDecay = [dk](float acc, float x) { return fmaxf(x, acc*dk); };
			e = serialPass<true>(eT, sT, M, e, maxDecay);
			e = serialPass<false>(sT, sT, M, e, maxDecay);
			e = serialPass<true>(sT, sT, M, e, maxDecay);
			e = serialPass<false>(sT, sT, M, e, maxDecay);
		}
		decay = 1/decay;
		{
			const float dk = decay;
			auto minGrow = [dk](float acc, float x) { return fminf(x, acc*dk); };
			for (int rep = 0; rep < 2; ++rep) {
				e = serialPass<true>(sT, sT, M, e, minGrow);
				e = serialPass<false>(sT, sT, M, e, minGrow);
			}
		}
		if (formants) {
			float *ratio = d.ratio + ((size_t)s*d.T + k)*M;
			for (int b = 0; b < M; ++b) {
				float inputF = (b + 0.5f)/Nf;
				float outputF = prm.formantCompensation ? mapFreqDev(d, prm, sg, inputF) : inputF;
				// invMapFormant, :920-925
				if (outputF*prm.invFormantMultiplier > prm.freqTonalityLimit) outputF = outputF + (1 - prm.formantMultiplier)*prm.freqTonalityLimit;
				else outputF = outputF*prm.invFormantMultiplier;
				const float inputE = sT[(size_t)b*64];
				float band = outputF*Nf - 0.5f;
				float targetE = 0;
				if (!(band < 0)) { // getFormant, :1009-1016 (entries M and M+1 of the metric are zero)
					band = fminf(band, float(M));
					const int fl = (int)floorf(band);
					const float fr = band - fl;
					const float low = (fl < M) ? sT[(size_t)fl*64] : 0.0f, high = (fl + 1 < M) ? sT[(size_t)(fl + 1)*64] : 0.0f;
					targetE = low + (high - low)*fr;
				}
				ratio[b] = targetE/(inputE + 1e-30f);
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------------
// K2b-e, scan form.  The same recurrences, one workgroup per (stream, hop) with the hop's arrays in LDS: every pass over
// the bins is a composition of per-bin maps  y -> m*y + c  (one-pole smoothing),  y -> max(c, m*y)  (formant decay) or
// y -> min(c, m*y)  (formant growth), which are closed under composition.  Each thread composes the maps of its own
// chunk of bins, the chunks' maps are scanned across the workgroup, and each thread then RE-RUNS THE REFERENCE'S SERIAL
// FORMULA over its chunk from the carry it received -- so only the carry entering a chunk is rounded differently from a
// bin-by-bin evaluation (and its influence decays geometrically inside the chunk).  Peak runs are summed by the thread
// that owns the run's first bin, in the reference's order.  The serial form above (kFeedEnergy + kFeedSerial) streams
// five [bin][64] arrays per tile through HBM (10 GB per 64-hop tile of 1024 streams, 4 ms); this form reads the input
// spectra once.
// ------------------------------------------------------------------------------------------------------
struct ScanMap { float m, c; };
template <int OP> __device__ __forceinline__ float scanApply(ScanMap f, float y) { // OP 0: m*y + c, 1: max(c, m*y), 2: min(c, m*y)
	const float v = f.m*y;
	return OP == 0 ? v + f.c : (OP == 1 ? fmaxf(f.c, v) : fminf(f.c, v));
}
template <int OP> __device__ __forceinline__ ScanMap scanCompose(ScanMap second, ScanMap first) { // second after first
	ScanMap r;
	r.m = second.m*first.m;
	r.c = scanApply<OP>(second, first.c);
	return r;
}
// One pass over src[0..M) in the given direction, result to dst (may alias src); `carry` enters the first bin of the
// pass and the value after the last bin is returned.  mStep / cOf(x): the per-bin map; step(e, x): the serial formula.
// maps: 4 entries of LDS scratch (the wave totals).  All 256 threads must call this.
template <int OP, bool DOWN, typename COf, typename Step>
__device__ __forceinline__ float scanPass(const float *src, float *dst, int M, float carry, float mStep, COf cOf, Step step, ScanMap *maps) {
	const int t = threadIdx.x;
	const int n = (M + 255)/256;              // bins per thread, in pass order
	const int p0 = t*n, p1 = min(M, p0 + n);  // pass positions [p0, p1); position p is bin DOWN ? M-1-p : p
	ScanMap f; f.m = 1.0f; f.c = OP == 0 ? 0.0f : (OP == 1 ? -INFINITY : INFINITY); // identity
	for (int p = p0; p < p1; ++p) {
		ScanMap g; g.m = mStep; g.c = cOf(src[DOWN ? M - 1 - p : p]);
		f = scanCompose<OP>(g, f);
	}
	// inclusive scan of the chunk maps inside each wave (Hillis-Steele over lane shuffles), the wave totals through LDS
	const int lane = t & 63;
	ScanMap inc = f;
#pragma unroll
	for (int dlt = 1; dlt < 64; dlt <<= 1) {
		ScanMap prev;
		prev.m = __shfl(inc.m, max(lane - dlt, 0));
		prev.c = __shfl(inc.c, max(lane - dlt, 0));
		if (lane >= dlt) inc = scanCompose<OP>(inc, prev);
	}
	ScanMap ex; // exclusive: the composition of the chunks before this one inside the wave
	ex.m = __shfl(inc.m, max(lane - 1, 0));
	ex.c = __shfl(inc.c, max(lane - 1, 0));
	if (lane == 0) { ex.m = 1.0f; ex.c = OP == 0 ? 0.0f : (OP == 1 ? -INFINITY : INFINITY); }
	if (lane == 63) maps[t >> 6] = inc;
	__syncthreads();
	float waveCarry = carry; // value entering this thread's wave
	for (int w = 0; w < (t >> 6); ++w) waveCarry = scanApply<OP>(maps[w], waveCarry);
	float e = scanApply<OP>(ex, waveCarry);
	float total = carry;
	for (int w = 0; w < 4; ++w) total = scanApply<OP>(maps[w], total);
	// the chunk again, with the reference's own formula, from the carry (src may alias dst: every thread reads and then
	// writes only its own chunk, and nobody reads another chunk after the barriers above)
	for (int p = p0; p < p1; ++p) {
		const int b = DOWN ? M - 1 - p : p;
		e = step(e, src[b]);
		dst[b] = e;
	}
	__syncthreads();
	return total;
}

// The same pass with every thread's chunk in REGISTERS.  Thread t owns bins [t*n, t*n + cnt) in both directions (the LDS form
// above partitions by pass position, so an up pass and a down pass give a thread different bins and every pass goes through
// LDS with two barriers and two dependent-latency walks); only the chunk maps cross lanes (shuffles, towards higher lanes for
// an up pass, towards lower lanes for a down pass) and the four wave totals cross waves (LDS, one barrier: the totals of
// consecutive passes use alternate halves of `maps`).  v[i], i < cnt: in = the pass's source, out = its result.
template <int OP, bool DOWN, int NMAX, typename COf, typename Step>
__device__ __forceinline__ float scanPassReg(float (&v)[NMAX], int cnt, float carry, float mStep, COf cOf, Step step, ScanMap *maps) {
	const int t = threadIdx.x, lane = t & 63, w = t >> 6;
	ScanMap f; f.m = 1.0f; f.c = OP == 0 ? 0.0f : (OP == 1 ? -INFINITY : INFINITY); // identity
#pragma unroll
	for (int j = 0; j < NMAX; ++j) {
		const int i = DOWN ? NMAX - 1 - j : j;
		if (i < cnt) { ScanMap g; g.m = mStep; g.c = cOf(v[i]); f = scanCompose<OP>(g, f); }
	}
	ScanMap inc = f;
#pragma unroll
	for (int dlt = 1; dlt < 64; dlt <<= 1) {
		const int from = DOWN ? min(lane + dlt, 63) : max(lane - dlt, 0);
		ScanMap prev;
		prev.m = __shfl(inc.m, from);
		prev.c = __shfl(inc.c, from);
		if (DOWN ? (lane + dlt <= 63) : (lane >= dlt)) inc = scanCompose<OP>(inc, prev);
	}
	ScanMap ex; // the chunks before this one in pass order, inside the wave
	ex.m = __shfl(inc.m, DOWN ? min(lane + 1, 63) : max(lane - 1, 0));
	ex.c = __shfl(inc.c, DOWN ? min(lane + 1, 63) : max(lane - 1, 0));
	if (lane == (DOWN ? 63 : 0)) { ex.m = 1.0f; ex.c = OP == 0 ? 0.0f : (OP == 1 ? -INFINITY : INFINITY); }
	if (lane == (DOWN ? 0 : 63)) maps[w] = inc;
	__syncthreads();
	float e = carry, total = carry;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int ww = DOWN ? 3 - j : j; // waves in pass order
		if (DOWN ? (ww > w) : (ww < w)) e = scanApply<OP>(maps[ww], e);
		total = scanApply<OP>(maps[ww], total);
	}
	e = scanApply<OP>(ex, e);
#pragma unroll
	for (int j = 0; j < NMAX; ++j) {
		const int i = DOWN ? NMAX - 1 - j : j;
		if (i < cnt) { e = step(e, v[i]); v[i] = e; }
	}
	return total;
}

// channel-summed energy of one hop into LDS, en[b] = sum_c |input_c[b]|^2 in channel order.  Eight independent loads per
// channel are in flight at a time: the plain loop (one bin per iteration, trip count unknown to the compiler) paid a full
// memory round trip per iteration -- 13 of them per workgroup, 27 % of the kernel (ablation on the GPU: 16.1 -> 11.7 ms
// per step of config 3 with the loads removed).
__device__ __forceinline__ void feedEnergyToLds(const DevBatch &d, const HopDesc &hd, int s, int sg, float *en) {
	const int M = d.M, C = d.C, t = threadIdx.x;
	for (int b0 = t; b0 < M; b0 += 8*256) {
		float acc[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
		for (int c = 0; c < C; ++c) {
			const float2 *row = inputRow(d, hd, s, sg, c);
			float2 v[8];
#pragma unroll
			for (int i = 0; i < 8; ++i) v[i] = row[min(b0 + 256*i, M - 1)];
#pragma unroll
			for (int i = 0; i < 8; ++i) acc[i] += cnorm(v[i]);
		}
#pragma unroll
		for (int i = 0; i < 8; ++i) if (b0 + 256*i < M) en[b0 + 256*i] = acc[i];
	}
}

// Pass A (Prediction.input / .energy rows, kPredictA below) for one hop, folded into the feed kernel when no hop of the tile has
// formant processing: the thread that has just computed the map entry of a bin forms the bin's (P, E) at once -- the map row is
// not read back (8 B per bin) and the input rows, which this workgroup read a moment ago for the energies, come out of L2
// instead of HBM.  Same arithmetic as kPredictA: bit-identical entries.  Eight bins per thread in flight.
struct LerpIndex;
__device__ __forceinline__ LerpIndex lerpIndex(float x);
__device__ __forceinline__ float2 bandAt(const float2 *row, int idx, int M);
template <typename MapAt>
__device__ __forceinline__ void feedPredictionRows(const DevBatch &d, const HopDesc &hd, int s, int sg, int k, bool mapped, MapAt mapAt, bool storeMap, const float *ratioLds);

// energy, smoothing, peaks, output map, raw pitch estimate: one workgroup per (hop, stream)
template <int NMAX, bool FUSE_PE = false> // NMAX: bins per thread held in registers during the smoothing passes; 0: through LDS (any M)
__global__ __launch_bounds__(256) void kFeedScanA(DevBatch d, int sBase, int hopBase) {
	static_assert(!FUSE_PE || NMAX > 0, "pass A is folded into the register form only");
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	const int k = blockIdx.x, s = blockIdx.y, sg = sBase + s;
	if (k >= d.nHops[s]) return;
	const HopDesc hd = d.hops[(size_t)sg*d.hopStride + hopBase + k];
	const bool mapped = hd.flags & HOP_MAPPED, formants = hd.flags & HOP_FORMANTS;
	if (!mapped && !formants) {
		if constexpr (FUSE_PE) feedPredictionRows(d, hd, s, sg, k, false, [](int bb) { return make_float2(float(bb), 1.0f); }, false, nullptr);
		return;
	}
	const int M = d.M, C = d.C, t = threadIdx.x;
	const float Nf = float(d.N);
	float *en = reinterpret_cast<float *>(smemRaw);         // [M] channel-summed energy
	float *sm = en + M;                                       // [M] smoothed
	float2 *pk = reinterpret_cast<float2 *>(sm + M);          // [M/2 + 2] peaks
	ScanMap *maps = reinterpret_cast<ScanMap *>(pk + M/2 + 2); // [264]
	int *counts = reinterpret_cast<int *>(maps + 264);         // [264]
	const StreamParams prm = d.params[sg];
	feedEnergyToLds(d, hd, s, sg, en);
	__syncthreads();
	if (mapped) {
		const float smoothingBins = Nf/float(d.I);
		const float slew = 1/(1 + smoothingBins*0.5f);
		auto pole = [slew](float acc, float x) { return acc + (x - acc)*slew; };
		auto cOf = [slew](float x) { return slew*x; };
		float e = 0;
		if constexpr (NMAX > 0) {
			const int n = (M + 255)/256, cnt = min(max(M - t*n, 0), n);
			float v[NMAX];
#pragma unroll
			for (int i = 0; i < NMAX; ++i) v[i] = (i < cnt) ? en[t*n + i] : 0.0f;
			e = scanPassReg<0, true>(v, cnt, e, 1 - slew, cOf, pole, maps);
			e = scanPassReg<0, false>(v, cnt, e, 1 - slew, cOf, pole, maps + 4);
			e = scanPassReg<0, true>(v, cnt, e, 1 - slew, cOf, pole, maps);
			e = scanPassReg<0, false>(v, cnt, e, 1 - slew, cOf, pole, maps + 4);
#pragma unroll
			for (int i = 0; i < NMAX; ++i) if (i < cnt) sm[t*n + i] = v[i];
			__syncthreads();
		} else {
			e = scanPass<0, true>(en, sm, M, e, 1 - slew, cOf, pole, maps);
			e = scanPass<0, false>(sm, sm, M, e, 1 - slew, cOf, pole, maps);
			e = scanPass<0, true>(sm, sm, M, e, 1 - slew, cOf, pole, maps);
			e = scanPass<0, false>(sm, sm, M, e, 1 - slew, cOf, pole, maps);
		}
		// findPeaks: every thread counts the runs that START in its chunk, an exclusive scan numbers them, and the owner
		// of a run's first bin sums the run in the reference's order (:866-873)
		const int n = (M + 255)/256, b0 = t*n, b1 = min(M, b0 + n);
		int starts = 0;
		unsigned startMask = 0; // bit i: a run starts at bin b0 + i
		if constexpr (NMAX > 0) { // the chunk's energies and smoothed energies side by side in registers: 2 n independent LDS reads
			bool prevAbove = b0 > 0 && b0 <= M && en[b0 - 1] > sm[b0 - 1];
#pragma unroll
			for (int i = 0; i < NMAX; ++i) {
				const bool above = (b0 + i < b1) && en[min(b0 + i, M - 1)] > sm[min(b0 + i, M - 1)];
				if (above && !prevAbove) { startMask |= 1u << i; ++starts; }
				prevAbove = above;
			}
		} else {
			for (int b = b0; b < b1; ++b) starts += (en[b] > sm[b]) && !(b > 0 && en[b - 1] > sm[b - 1]);
		}
		{ // exclusive prefix sum of the run starts: lane shuffles inside the wave (a serial 64-entry loop by one lane per wave
			// cost 6 us per workgroup), wave totals through LDS
			const int lane = t & 63;
			int inc = starts;
#pragma unroll
			for (int dlt = 1; dlt < 64; dlt <<= 1) {
				const int prev = __shfl(inc, max(lane - dlt, 0));
				if (lane >= dlt) inc += prev;
			}
			counts[t] = inc - starts;
			if (lane == 63) counts[256 + (t >> 6)] = inc;
		}
		__syncthreads();
		int idx = counts[t];
		for (int w = 0; w < (t >> 6); ++w) idx += counts[256 + w];
		const int nPeaks = counts[256] + counts[257] + counts[258] + counts[259];
		for (int b = b0; b < b1; ++b) {
			if (NMAX > 0 ? ((startMask >> (b - b0)) & 1u) != 0 : ((en[b] > sm[b]) && !(b > 0 && en[b - 1] > sm[b - 1]))) {
				float bandSum = 0, energySum = 0;
				for (int q = b; q < M; q += 4) { // four bins per LDS round trip; the additions stay in the reference's order
					float e4[4], s4[4];
#pragma unroll
					for (int i = 0; i < 4; ++i) { const int qi = min(q + i, M - 1); e4[i] = en[qi]; s4[i] = sm[qi]; }
					bool open = true;
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						open = open && (q + i < M) && e4[i] > s4[i];
						if (open) { bandSum += (q + i)*e4[i]; energySum += e4[i]; }
					}
					if (!open) break;
				}
				const float avgBand = bandSum/energySum;
				const float avgFreq = (avgBand + 0.5f)/Nf;
				pk[idx++] = make_float2(avgBand, mapFreqDev(d, prm, sg, avgFreq)*Nf - 0.5f);
			}
		}
		__syncthreads();
		// updateOutputMap, :882-917 (same segment rules as the serial form; the covering pair by bisection)
		float2 *mapRow = d.map + ((size_t)s*d.T + k)*M;
		const float2 first = nPeaks > 0 ? pk[0] : make_float2(0.f, 0.f);
		const float2 lastP = nPeaks > 0 ? pk[nPeaks - 1] : make_float2(0.f, 0.f);
		const int topStart = max(0, (int)lastP.y), bottomEnd = min(M, (int)ceilf(first.y));
		auto mapOf = [&](int b, int lo) { // lo = largest q in [0, nPeaks-2] with max(0, ceil(peaks[q].out)) <= b (only used between bottomEnd and topStart)
			float2 mp = make_float2(float(b), 1.0f);
			if (nPeaks > 0) {
				if (b >= topStart) {
					mp = make_float2(b + (lastP.x - lastP.y), 1.0f);
				} else if (b < bottomEnd) {
					mp = make_float2(b + (first.x - first.y), 1.0f);
				} else if (nPeaks >= 2) {
					const float2 prev = pk[lo], next = pk[lo + 1];
					if (b < min(M, (int)ceilf(next.y))) {
						float rangeScale = 1/(next.y - prev.y);
						float outOffset = prev.x - prev.y;
						float outScale = next.x - next.y - prev.x + prev.y;
						float gradScale = outScale*rangeScale;
						float r = (b - prev.y)*rangeScale;
						float h = r*r*(3 - 2*r);
						float outB = b + outOffset + h*outScale;
						float gradH = 6*r*(1 - r);
						mp = make_float2(outB, 1 + gradH*gradScale);
					}
				}
			}
			return mp;
		};
		if constexpr (NMAX > 0) {
			// The covering pair of every bin without a search: every peak marks the bin its segment starts at (LDS atomic max:
			// several peaks may start at one bin, the last one counts), a prefix maximum over the bins spreads the marks.  The
			// bisection it replaces cost ten rounds of nine instructions per bin for a noise spectrum (700 peaks): 5.2 of the
			// kernel's 13 ms per step of config 3.  Same result for ascending peak positions (every map the tonality-limit rule
			// or an ascending table produces); for a descending custom map both are arbitrary.
			int *cover = reinterpret_cast<int *>(sm);          // the smoothed energies are dead after the run sums
			int *waveMax = reinterpret_cast<int *>(maps + 8);
			const int cnt = max(b1 - b0, 0), lane = t & 63, w = t >> 6;
#pragma unroll
			for (int i = 0; i < NMAX; ++i) if (i < cnt) cover[b0 + i] = -1;
			__syncthreads();
			for (int q = t; q <= nPeaks - 2; q += 256) {
				const int start = max(0, (int)ceilf(pk[q].y));
				if (start < M) atomicMax(&cover[start], q);
			}
			__syncthreads();
			int c[NMAX], run = -1;
#pragma unroll
			for (int i = 0; i < NMAX; ++i) { if (i < cnt) run = max(run, cover[b0 + i]); c[i] = run; }
			int inc = run;
#pragma unroll
			for (int dlt = 1; dlt < 64; dlt <<= 1) {
				const int prev = __shfl(inc, max(lane - dlt, 0));
				if (lane >= dlt) inc = max(inc, prev);
			}
			int base = __shfl(inc, max(lane - 1, 0));
			if (lane == 0) base = -1;
			if (lane == 63) waveMax[w] = inc;
			__syncthreads();
			for (int ww = 0; ww < w; ++ww) base = max(base, waveMax[ww]);
#pragma unroll
			for (int i = 0; i < NMAX; ++i) if (i < cnt) cover[b0 + i] = max(c[i], base);
			__syncthreads();
			if constexpr (FUSE_PE) feedPredictionRows(d, hd, s, sg, k, true, [&](int bb) { return mapOf(bb, max(cover[bb], 0)); }, true, nullptr);
			else for (int b = t; b < M; b += 256) mapRow[b] = mapOf(b, max(cover[b], 0)); // coalesced stores
		} else {
			// by bisection; a thread's bins are 256 apart, so eight independent bisections run in lock step
			for (int bb = t; bb < M; bb += 8*256) {
				int lo[8], hi[8];
#pragma unroll
				for (int i = 0; i < 8; ++i) { lo[i] = 0; hi[i] = nPeaks - 2; }
				if (nPeaks >= 2) {
					for (int span = nPeaks - 2; span > 0; span >>= 1) { // ceil(log2(nPeaks - 1)) rounds settle every bisection
						float y[8];
#pragma unroll
						for (int i = 0; i < 8; ++i) y[i] = pk[(lo[i] + hi[i] + 1) >> 1].y;
#pragma unroll
						for (int i = 0; i < 8; ++i) {
							const int mid = (lo[i] + hi[i] + 1) >> 1;
							if (lo[i] < hi[i]) { if (max(0, (int)ceilf(y[i])) <= bb + 256*i) lo[i] = mid; else hi[i] = mid - 1; }
						}
					}
				}
#pragma unroll
				for (int i = 0; i < 8; ++i) if (bb + 256*i < M) mapRow[bb + 256*i] = mapOf(bb + 256*i, max(lo[i], 0));
			}
		}
	}
	if (formants && prm.formantBaseFreq <= 0) {
		// estimateFrequency() raw part, :929-960: the three highest local maxima of the metric (= the channel-summed
		// energy), ties to the earlier bin, three copies of bin 0 as the initial entries -- a serial walk by one thread
		// (compares only, no arithmetic: 3 k steps)
		__syncthreads();
		if (t == 0) {
			int p0 = 0, p1 = 0, p2 = 0;
			float e0 = en[0], e1 = en[0], e2 = en[0];
			for (int b = 1; b < M - 1; ++b) {
				const float e = en[b];
				if (!(e < en[b - 1] || e <= en[b + 1])) {
					if (e > e0) {
						if (e > e1) {
							if (e > e2) { p0 = p1; e0 = e1; p1 = p2; e1 = e2; p2 = b; e2 = e; }
							else { p0 = p1; e0 = e1; p1 = b; e1 = e; }
						} else {
							p0 = b; e0 = e;
						}
					}
				}
			}
			int peakEstimate = p2;
			if (e1 > e2*0.1f) {
				int diff = abs(peakEstimate - p1);
				if (diff > peakEstimate/8 && diff < peakEstimate*7/8) peakEstimate = peakEstimate%diff;
				if (e0 > e2*0.01f) {
					int diff2 = abs(peakEstimate - p0);
					if (diff2 > peakEstimate/8 && diff2 < peakEstimate*7/8) peakEstimate = peakEstimate%diff2;
				}
			}
			d.est[((size_t)s*d.T + k)*2] = peakEstimate*e2;
			d.est[((size_t)s*d.T + k)*2 + 1] = e2;
		}
	}
}

// The pitch estimate is smoothed from hop to hop (:962-965): one thread per stream replays the tile's hops in order
__global__ __launch_bounds__(64) void kFeedFreq(DevBatch d, int sBase, int nStreams, int hopBase) {
	const int s = blockIdx.x*blockDim.x + threadIdx.x;
	if (s >= nStreams) return;
	const int sg = sBase + s, nh = d.nHops[s];
	const StreamParams prm = d.params[sg];
	const float Nf = float(d.N);
	float w = d.stFreq[2*sg], wt = d.stFreq[2*sg + 1];
	for (int j = 0; j < nh; ++j) {
		const HopDesc hj = d.hops[(size_t)sg*d.hopStride + hopBase + j];
		float fe = prm.formantBaseFreq*Nf - 0.5f; // freqToBand, :982
		if ((hj.flags & HOP_FORMANTS) && prm.formantBaseFreq <= 0) {
			w += (d.est[((size_t)s*d.T + j)*2] - w)*0.25f;
			wt += (d.est[((size_t)s*d.T + j)*2 + 1] - wt)*0.25f;
			fe = w/(wt + 1e-30f);
		}
		d.freqEst[(size_t)s*d.T + j] = fe;
	}
}

// formant envelope (2 x (down, up) max-decay, 2 x (down, up) min-grow, :987-1006) and the per-bin energy ratio (:1018-1033)
template <int NMAX, bool FUSE_PE = false> // FUSE_PE: pass A of the tile's hops here (tiles WITH formant processing; the ratios stay in LDS)
__global__ __launch_bounds__(256) void kFeedScanC(DevBatch d, int sBase, int hopBase) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	const int k = blockIdx.x, s = blockIdx.y, sg = sBase + s;
	if (k >= d.nHops[s]) return;
	const HopDesc hd = d.hops[(size_t)sg*d.hopStride + hopBase + k];
	if (!(hd.flags & HOP_FORMANTS)) {
		if constexpr (FUSE_PE) {
			const float2 *mapRowIn = d.map + ((size_t)s*d.T + k)*d.M;
			feedPredictionRows(d, hd, s, sg, k, (hd.flags & HOP_MAPPED) != 0, [&](int bb) { return mapRowIn[bb]; }, false, nullptr);
		}
		return;
	}
	const int M = d.M, C = d.C, t = threadIdx.x;
	const float Nf = float(d.N);
	float *en = reinterpret_cast<float *>(smemRaw);
	float *sm = en + M;
	ScanMap *maps = reinterpret_cast<ScanMap *>(sm + M);
	const StreamParams prm = d.params[sg];
	feedEnergyToLds(d, hd, s, sg, en);
	__syncthreads();
	const float freqEstimate = d.freqEst[(size_t)s*d.T + k];
	float decay = 1 - 1/(freqEstimate*0.5f + 1);
	float e = 0;
	auto ident = [](float x) { return x; };
	if constexpr (NMAX > 0) {
		const int n = (M + 255)/256, cnt = min(max(M - t*n, 0), n);
		float v[NMAX];
#pragma unroll
		for (int i = 0; i < NMAX; ++i) v[i] = (i < cnt) ? en[t*n + i] : 0.0f;
		{
			const float dk = decay;
			auto maxDecay = [dk](float acc, float x) { return fmaxf(x, acc*dk); };
			e = scanPassReg<1, true>(v, cnt, e, dk, ident, maxDecay, maps);
			e = scanPassReg<1, false>(v, cnt, e, dk, ident, maxDecay, maps + 4);
			e = scanPassReg<1, true>(v, cnt, e, dk, ident, maxDecay, maps);
			e = scanPassReg<1, false>(v, cnt, e, dk, ident, maxDecay, maps + 4);
		}
		decay = 1/decay;
		{
			const float dk = decay;
			auto minGrow = [dk](float acc, float x) { return fminf(x, acc*dk); };
			for (int rep = 0; rep < 2; ++rep) {
				e = scanPassReg<2, true>(v, cnt, e, dk, ident, minGrow, maps);
				e = scanPassReg<2, false>(v, cnt, e, dk, ident, minGrow, maps + 4);
			}
		}
#pragma unroll
		for (int i = 0; i < NMAX; ++i) if (i < cnt) sm[t*n + i] = v[i];
		__syncthreads();
	} else {
		{
			const float dk = decay;
			auto maxDecay = [dk](float acc, float x) { return fmaxf(x, acc*dk); };
			e = scanPass<1, true>(en, sm, M, e, dk, ident, maxDecay, maps);
			e = scanPass<1, false>(sm, sm, M, e, dk, ident, maxDecay, maps);
			e = scanPass<1, true>(sm, sm, M, e, dk, ident, maxDecay, maps);
			e = scanPass<1, false>(sm, sm, M, e, dk, ident, maxDecay, maps);
		}
		decay = 1/decay;
		{
			const float dk = decay;
			auto minGrow = [dk](float acc, float x) { return fminf(x, acc*dk); };
			for (int rep = 0; rep < 2; ++rep) {
				e = scanPass<2, true>(sm, sm, M, e, dk, ident, minGrow, maps);
				e = scanPass<2, false>(sm, sm, M, e, dk, ident, minGrow, maps);
			}
		}
	}
	float *ratio = d.ratio + ((size_t)s*d.T + k)*M;
	for (int b = t; b < M; b += 256) {
		float inputF = (b + 0.5f)/Nf;
		float outputF = prm.formantCompensation ? mapFreqDev(d, prm, sg, inputF) : inputF;
		if (outputF*prm.invFormantMultiplier > prm.freqTonalityLimit) outputF = outputF + (1 - prm.formantMultiplier)*prm.freqTonalityLimit; // invMapFormant, :920-925
		else outputF = outputF*prm.invFormantMultiplier;
		const float inputE = sm[b];
		float band = outputF*Nf - 0.5f;
		float targetE = 0;
		if (!(band < 0)) { // getFormant, :1009-1016 (entries M and M+1 of the metric are zero)
			band = fminf(band, float(M));
			const int fl = (int)floorf(band);
			const float fr = band - fl;
			const float low = (fl < M) ? sm[fl] : 0.0f, high = (fl + 1 < M) ? sm[fl + 1] : 0.0f;
			targetE = low + (high - low)*fr;
		}
		if constexpr (FUSE_PE) en[b] = targetE/(inputE + 1e-30f); // the energies are dead: the ratios take their place in LDS
		else ratio[b] = targetE/(inputE + 1e-30f);
	}
	if constexpr (FUSE_PE) {
		__syncthreads();
		const float2 *mapRowIn = d.map + ((size_t)s*d.T + k)*M;
		feedPredictionRows(d, hd, s, sg, k, (hd.flags & HOP_MAPPED) != 0, [&](int bb) { return mapRowIn[bb]; }, false, en);
	}
}

// ------------------------------------------------------------------------------------------------------
// K2a+K2f: per-(hop, channel, bin) prediction coefficients -- everything the bin recurrence needs that does
// not depend on previous outputs (signalsmith-stretch.h:642-660 rotation, :697-719 preliminary prediction,
// :748-785 vertical twists):
//   P  = lerp(input, map.inputBin)                      (Prediction.input)
//   E  = lerp(inputEnergy, map.inputBin)*max(0, grad)   (Prediction.energy)
//   TW = rot[b] * P * conj(lerp(rot*prevInput, map.inputBin))      (output twist of the preliminary prediction)
//   S  = P * conj(lerp(input, map.inputBin - tf)),  T = P * conj(lerp(input, map.inputBin - L*tf))
// ------------------------------------------------------------------------------------------------------
struct LerpIndex {
	int lo;
	float fr;
};
__device__ __forceinline__ LerpIndex lerpIndex(float x) { // :559-562
	LerpIndex r;
	float fl = floorf(x);
	r.lo = (int)fl;
	r.fr = x - fl;
	return r;
}
__device__ __forceinline__ float2 bandAt(const float2 *row, int idx, int M) { // getBand, :548-551
	// branch-free: always load (clamped index), then zero outside [0, M) -- keeps every load of a record in one
	// basic block so the compiler can issue them back to back (memory-level parallelism of the record producers)
	const int ci = min(max(idx, 0), M - 1);
	const float2 v = row[ci];
	return (ci == idx) ? v : make_float2(0.f, 0.f);
}
struct BandPair { float2 lo, hi; };
// bins idx and idx+1 of a row with ONE 16-byte load (half the memory instructions of two 8-byte loads): the pair is read at
// a clamped position and the taps outside [0, M) are zeroed afterwards with selects (a branch here would end the basic
// block and the loads of the next tap would wait behind it)
__device__ __forceinline__ BandPair pairAt(const float2 *row, int idx, int M) {
	const int ci = min(max(idx, 0), M - 2);
	const float4 v = *reinterpret_cast<const float4 *>(row + ci); // 8-byte aligned; gfx9 global loads need dword alignment only
	const int delta = idx - ci; // 0 in range; -1: low tap is bin -1; +1: low tap is bin M-1; otherwise both taps are outside
	const bool d0 = delta == 0, dp = delta == 1, dm = delta == -1;
	BandPair r;
	r.lo = make_float2(d0 ? v.x : (dp ? v.z : 0.f), d0 ? v.y : (dp ? v.w : 0.f));
	r.hi = make_float2(d0 ? v.z : (dm ? v.x : 0.f), d0 ? v.w : (dm ? v.y : 0.f));
	return r;
}
__device__ __forceinline__ float2 lerpBand(const float2 *row, LerpIndex li, int M) { // getFractional, :553-557
	const BandPair p = pairAt(row, li.lo, M);
	return clerp(p.lo, p.hi, li.fr);
}
__device__ __forceinline__ float2 rotAt(const float2 *rot, int M, int idx, bool rotate) { // hop rotation of bin idx, 1 outside / when off
	const int ci = min(max(idx, 0), M - 1);
	const float2 v = rot[ci];
	return (rotate && ci == idx) ? v : make_float2(1.f, 0.f);
}

// pass A: P and E in row layout [s][k][c][M]
__global__ __launch_bounds__(256) void kPredictA(DevBatch d, int sBase, int hopBase) {
	const int b = blockIdx.x*blockDim.x + threadIdx.x;
	const int k = blockIdx.y, s = blockIdx.z, sg = sBase + s;
	const HopDesc hd = d.hops[(size_t)sg*d.hopStride + hopBase + k];
	if (!(hd.flags & HOP_ACTIVE) || b >= d.M) return;
	const int M = d.M;
	const bool mapped = hd.flags & HOP_MAPPED, formants = hd.flags & HOP_FORMANTS;
	float2 mp = mapped ? d.map[((size_t)s*d.T + k)*M + b] : make_float2(float(b), 1.0f);
	const LerpIndex li = lerpIndex(mp.x);
	const float gradScale = fmaxf(0.0f, mp.y);
	const float *ratio = formants ? d.ratio + ((size_t)s*d.T + k)*M : nullptr;
	const bool loIn = li.lo >= 0 && li.lo < M, hiIn = li.lo + 1 >= 0 && li.lo + 1 < M;
	for (int c = 0; c < d.C; ++c) {
		const float2 *in = inputRow(d, hd, s, sg, c);
		const size_t o = rowOf(d, s, k, c) + b;
		float2 inLo = bandAt(in, li.lo, M), inHi = bandAt(in, li.lo + 1, M);
		float eLo = cnorm(inLo), eHi = cnorm(inHi);
		if (formants) {
			if (loIn) eLo *= ratio[li.lo];
			if (hiIn) eHi *= ratio[li.lo + 1];
		}
		// Prediction.input and Prediction.energy of a bin side by side: their readers fetch both with one 12-byte load
		PredEntry pe;
		pe.x = inLo.x + (inHi.x - inLo.x)*li.fr;
		pe.y = inLo.y + (inHi.y - inLo.y)*li.fr;
		pe.e = (eLo + (eHi - eLo)*li.fr)*gradScale;
		d.PE[o] = pe;
	}
}

// Per-channel fields of a record (floats 9..).  Any channel count but two: {P_c, sqrt(E_c)} per channel, the recurrence forms
// the lock  makeOutput(out_m * P_c conj(P_m), P_c, sqrt(E_c))  itself (:791-800).  STEREO: the one locked channel's
// makeOutput is folded into the record -- |out_m|^2 = E_m by construction (:602), so the norm of the lock's phase is
// E_m |P_o conj(P_m)|^2 and the producer can scale the twist itself:
//   9..11  Fb = P_m * sqrt(E_m) / sqrt(|P_m|^2 + 1e-15), sqrt(E_m)   (the maximum channel's own makeOutput: Fb is what it returns when
//          the prediction is below the noise floor, :598-601 -- formed here, off the serial path, from the same operations)
//   12,13  T' = P_o conj(P_m) * sqrt(E_o) / sqrt(E_m |P_o conj(P_m)|^2)        (0 if that norm is below the noise floor)
//   14,15  F  = P_o * sqrt(E_o) / sqrt(|P_o|^2 + 1e-15)  in that weak case, else 0   (the fallback to the input, :598-601)
// and the recurrence wave computes  out_o = out_m T' + F : one complex multiply-add instead of two multiplies, a norm, a
// compare, a reciprocal square root and four selects ON THE SERIAL PATH -- 77 of the 563 clock cycles a step took (cycle
// trace, tools/probes/voc_trace.py).  The norm is E_m |T|^2 instead of |out_m T|^2: equal up to rounding (1e-7 relative).
template <int CH, int NFLOATS>
__device__ __forceinline__ void recordChannelFields(float (&f)[NFLOATS], const float2 (&p)[CH], const float (&e)[CH], int mc) {
	if constexpr (CH == 2) {
		const float2 Pm = mc ? p[1] : p[0], Po = mc ? p[0] : p[1];
		const float eM = mc ? e[1] : e[0], eO = mc ? e[0] : e[1];
		const float2 T = cmulc(Po, Pm);
		const float nT = eM*cnorm(T);
		const bool weak = nT <= 1e-15f;
		const float g = __builtin_amdgcn_sqrtf(eO)*__builtin_amdgcn_rsqf(weak ? cnorm(Po) + 1e-15f : nT);
		const float sM = __builtin_amdgcn_sqrtf(eM);
		const float2 Fb = cscale(Pm, sM*__builtin_amdgcn_rsqf(cnorm(Pm) + 1e-15f));
		f[9] = Fb.x; f[10] = Fb.y; f[11] = sM;
		f[12] = weak ? 0.0f : T.x*g; f[13] = weak ? 0.0f : T.y*g;
		f[14] = weak ? Po.x*g : 0.0f; f[15] = weak ? Po.y*g : 0.0f;
	} else {
#pragma unroll
		for (int c = 0; c < CH; ++c) { f[9 + 3*c] = p[c].x; f[10 + 3*c] = p[c].y; f[11 + 3*c] = __builtin_amdgcn_sqrtf(e[c]); } // 1-ulp hardware square root
	}
}
// stereo: the locked channel's output from the maximum channel's (see recordChannelFields)
template <int NFLOATS>
__device__ __forceinline__ float2 lockedOutput(float2 om, const float (&f)[NFLOATS]) {
	return cfma(om, make_float2(f[12], f[13]), make_float2(f[14], f[15]));
}

// storeMap: mapAt has just computed the entry (kFeedScanA) and it goes to the map row; otherwise mapAt reads that row.
// ratioLds: the hop's formant energy ratios (kFeedScanC keeps them in LDS), applied to the two energy taps as kPredictA does.
template <typename MapAt>
__device__ __forceinline__ void feedPredictionRows(const DevBatch &d, const HopDesc &hd, int s, int sg, int k, bool mapped, MapAt mapAt, bool storeMap, const float *ratioLds) {
	const int M = d.M, C = d.C, t = threadIdx.x;
	float2 *mapRow = d.map + ((size_t)s*d.T + k)*M;
	for (int b0 = t; b0 < M; b0 += 8*256) {
		LerpIndex li[8];
		float grad[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const int b = b0 + 256*i;
			float2 mp = make_float2(float(b), 1.0f);
			if (b < M && mapped) { mp = mapAt(b); if (storeMap) mapRow[b] = mp; }
			li[i] = lerpIndex(mp.x);
			grad[i] = fmaxf(0.0f, mp.y);
		}
		for (int c = 0; c < C; ++c) {
			const float2 *in = inputRow(d, hd, s, sg, c);
			float2 lo[8], hi[8];
#pragma unroll
			for (int i = 0; i < 8; ++i) { lo[i] = bandAt(in, li[i].lo, M); hi[i] = bandAt(in, li[i].lo + 1, M); }
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const int b = b0 + 256*i;
				if (b >= M) continue;
				float eLo = cnorm(lo[i]), eHi = cnorm(hi[i]);
				if (ratioLds) {
					if (li[i].lo >= 0 && li[i].lo < M) eLo *= ratioLds[li[i].lo];
					if (li[i].lo + 1 >= 0 && li[i].lo + 1 < M) eHi *= ratioLds[li[i].lo + 1];
				}
				PredEntry pe;
				pe.x = lo[i].x + (hi[i].x - lo[i].x)*li[i].fr;
				pe.y = lo[i].y + (hi[i].y - lo[i].y)*li[i].fr;
				pe.e = (eLo + (eHi - eLo)*li[i].fr)*grad[i];
				d.PE[rowOf(d, s, k, c) + b] = pe;
			}
		}
	}
}

// previous-hop part of the prediction,  prevOut[b+1]*Cc + prevOut[b+L]*Dc : THE definition every recurrence kernel uses (first
// operation of its accumulation), and what FOLD0 records carry in Cc's place
__device__ __forceinline__ float2 prevHopTerms(float2 p1, float2 Cc, float2 pL, float2 Dc) { return cfma(pL, Dc, cmul(p1, Cc)); }
__device__ __forceinline__ void foldCarriedTaps(const CarriedOutput &prev, int mc, int b, int M, int L, float2 &Cc, float2 &Dc) {
	// taps beyond the last bin multiply coefficients that are already zero (:765,:776): any finite value does
	const float2 p1 = prev[(size_t)mc*M + min(b + 1, M - 1)], pL = prev[(size_t)mc*M + min(b + L, M - 1)];
	Cc = prevHopTerms(p1, Cc, pL, Dc);
	Dc = make_float2(0.f, 0.f);
}

// One record of the bin recurrence = everything hop k needs at bin b, with the maximum-energy channel m(b)
// already selected (signalsmith-stretch.h:729-737):
//   phi = out_m[b-1]*A + out_m[b-L]*B + prevHopOut_m[b+1]*Cc + prevHopOut_m[b+L]*Dc         (:744-786)
//   A  = P_m[b] conj(lerp(in_m, map[b]-tf)),  B = P_m[b] conj(lerp(in_m, map[b]-L tf))      (up-steps, :748-762)
//   Cc = TW_m[b+1]/(max(Eprev_m[b+1],E_m[b+1])+eps) * conj(P_m[b+1] conj(lerp(in_m, map[b+1]-tf)))   (:765-774 with
//        the preliminary prediction :714-716 folded in; TW = rot[b+1] P_m[b+1] conj(lerp(rot*prev_m, map[b+1])))
//   Dc = the same at b+L with L tf                                                          (:776-785)
// followed, per channel c, by {P_c[b], sqrt(E_c[b])} for makeOutput (:596-603) and the channel lock
// (:791-800, lock twist P_c conj(P_m) formed in the recurrence kernel).  Floats: 0-7 A,B,Cc,Dc; 8 m; 9+3c.. per channel.  Records live in a SKEWED layout
// REC[s][t][chunk][lane k] (float4 chunks, t = b + lag*k) so that step t of the wavefront is one contiguous block.

// PLAIN = no hop of the tile has a pitch map or formant processing: then Prediction.input is the input spectrum
// itself and Prediction.energy its squared magnitude (signalsmith-stretch.h:676-685,:708-710), so pass A is skipped.
template <int CH, bool PLAIN>
struct RecordSource {
	const DevBatch &d;
	const HopDesc &hd;
	int s, k, sg, M;
	const float2 *in0; // channel 0 input row; channel rows are `pitch` apart (tile buffer: Mp, carried state: M)
	int pitch;
	__device__ RecordSource(const DevBatch &d_, const HopDesc &hd_, int s_, int k_, int sg_) : d(d_), hd(hd_), s(s_), k(k_), sg(sg_), M(d_.M) {
		in0 = inputRow(d, hd, s, sg, 0);
		pitch = (hd.inSrc >= 0) ? d.Mp : d.M;
	}
	__device__ __forceinline__ const float2 *inRow(int c) const { return in0 + (size_t)c*pitch; }
	// (Prediction.input.x, .y, Prediction.energy, -) of channel c at bin b
	__device__ __forceinline__ float4 PE(int c, int b) const {
		if (PLAIN) { const float2 p = in0[(size_t)c*pitch + b]; return make_float4(p.x, p.y, cnorm(p), 0.0f); }
		const PredEntry pe = d.PE[rowOf(d, s, k, c) + b];
		return make_float4(pe.x, pe.y, pe.e, 0.0f);
	}
	__device__ __forceinline__ float2 mapAt(int b) const {
		if (PLAIN) return make_float2(float(b), 1.0f);
		const float2 m = d.map[((size_t)s*d.T + k)*M + b]; // always loaded (the row exists, mapped or not), selected afterwards
		return (hd.flags & HOP_MAPPED) ? m : make_float2(float(b), 1.0f);
	}
};

// Prediction.energy of the previous hop at a bin: the carried state (fp32 or fp16, by ELEMENT INDEX through the accessor -- the
// typed pointer into d.stEnergy addresses a half-sized allocation in fp16 mode), hop k-1's (P, E) entries, or -- plain tiles --
// the squared magnitude of hop k-1's input spectrum
struct PrevEnergy {
	bool carried;        // the tile's first hop: the carried state, element carriedBase + bin
	size_t carriedBase;
	const float *row;    // inside a mapped tile: third float of hop k-1's 12-byte entries (stride 3)
	const float2 *input; // inside a plain tile: hop k-1's input spectrum
	__device__ __forceinline__ float at(const DevBatch &d, int bc) const {
		if (carried) return loadCarriedEnergy(d, carriedBase + bc);
		if (input) return cnorm(input[bc]);
		return row[(size_t)bc*3];
	}
};
// coefficient multiplying the previous hop's final output at bin bx (bx = b+1 or b+L), see the record description
template <int CH, bool PLAIN>
__device__ __forceinline__ float2 twistAt(const RecordSource<CH, PLAIN> &src, int mc, int bx, float2 mp, bool rotate, const float2 *in,
                                          const float2 *pv, const PrevEnergy &prevE, float tfDown, float stepMul,
                                          const float2 *rot) {
	// bx may be one past the last bin for the callers' masked-out cases: every access below clamps; mp = mapAt(min(bx, M-1))
	const DevBatch &d = src.d;
	const int M = src.M;
	const int bc = min(bx, M - 1);
	const float2 rotB = rotAt(rot, M, bc, rotate);
	float2 Q;
	if (PLAIN) { // identity map: the previous-input tap sits exactly on bin bc (fraction 0), and shares its rotation
		Q = cmul(pv[bc], rotB);
	} else {
		const LerpIndex li = lerpIndex(mp.x);
		const BandPair pvp = pairAt(pv, li.lo, M), rp = pairAt(rot, li.lo, M); // two 16-byte loads instead of four 8-byte ones
		const bool loIn = li.lo >= 0 && li.lo < M, hiIn = li.lo + 1 >= 0 && li.lo + 1 < M;
		const float2 one = make_float2(1.f, 0.f);
		const float2 qLo = cmul(pvp.lo, (rotate && loIn) ? rp.lo : one);
		const float2 qHi = cmul(pvp.hi, (rotate && hiIn) ? rp.hi : one);
		Q = clerp(qLo, qHi, li.fr);
	}
	const float4 pe = src.PE(mc, bc);
	const float2 Px = make_float2(pe.x, pe.y);
	const float2 TW = cmul(rotB, cmulc(Px, Q));
	const float eNow = pe.z;
	const float ePrev = prevE.at(d, bc);
	const float den = fmaxf(ePrev, eNow) + 1e-15f; // :716
	const float2 down = cmulc(Px, lerpBand(in, lerpIndex(mp.x - stepMul*tfDown), M));
	const float2 r = cmulc(TW, down);
	const float inv = __builtin_amdgcn_rcpf(den); // 1-ulp hardware reciprocal (an IEEE division costs ten instructions per record)
	return make_float2(r.x*inv, r.y*inv);
}

// Fills one record.  Per-channel fields: see recordChannelFields.  (LOCK: kept in the signature for the call sites, always false.)
// SPEC (mono/stereo): the four twists are evaluated for EVERY channel and the maximum-energy channel's set is
// selected afterwards, so no load address depends on loaded data (one memory round trip per record instead of two).
// ROT_LDS: the hop-rotation table is read from `rotLds` (a copy in LDS) instead of d.rot -- three of a mapped record's 18
// gathers per twist pair go to the table, and the texture-address unit is what bounds the gathering producers.
// FOLD0 (kVocoder): hop 0 of a tile takes its previous-hop taps from the CARRIED Band.output, which is known before the kernel
// starts -- so its record carries  Cc := prevOut[b+1]*Cc + prevOut[b+L]*Dc  (formed with the very helper calls, in the very
// order, the recurrence uses: bit-identical) and Dc := 0, and the recurrence wave's lane 0 holds the constant taps (1, 0) and
// (0, 0).  Nothing on the serial path reads the carried state any more (it was two LDS reads a step plus a staging window).
template <int CH, bool PLAIN, bool LOCK, bool SPEC, int NFLOATS, bool ROT_LDS = false, bool FOLD0 = false>
__device__ __forceinline__ void computeRecord(const DevBatch &d, const HopDesc &hd, const HopDesc &hp, int s, int sg, int k, int b, float (&f)[NFLOATS],
                                              const float2 *rotLds = nullptr) {
	const float2 *rot;
	if constexpr (ROT_LDS) rot = rotLds; else rot = d.rot;
	const int M = d.M, L = d.L;
	const bool rotate = hd.flags & HOP_NEW_SPECTRUM, randomTf = hd.flags & HOP_RANDOM_TF;
	const RecordSource<CH, PLAIN> src(d, hd, s, k, sg);
	float2 p[CH];
	float e[CH];
#pragma unroll
	for (int c = 0; c < CH; ++c) {
		const float4 pe = src.PE(c, b);
		p[c] = make_float2(pe.x, pe.y);
		e[c] = pe.z;
	}
	int mc = 0; // maximum-energy channel, first maximum wins (:729-737)
	float eMax = e[0];
#pragma unroll
	for (int c = 1; c < CH; ++c) {
		if (e[c] > eMax) { mc = c; eMax = e[c]; }
	}
	float2 Pm = p[0];
#pragma unroll
	for (int c = 1; c < CH; ++c) if (c == mc) Pm = p[c];
	// the map entries of bins b and b+1 are adjacent: one 16-byte load; b+L separately
	float2 mp, mp1;
	if (PLAIN) {
		mp = make_float2(float(b), 1.0f);
		mp1 = make_float2(float(min(b + 1, M - 1)), 1.0f);
	} else {
		const int ci = min(b, M - 2);
		const float4 pr = *reinterpret_cast<const float4 *>(d.map + ((size_t)s*d.T + k)*M + ci);
		const bool mapped = hd.flags & HOP_MAPPED;
		mp = mapped ? ((b == ci) ? make_float2(pr.x, pr.y) : make_float2(pr.z, pr.w)) : make_float2(float(b), 1.0f);
		mp1 = mapped ? make_float2(pr.z, pr.w) : make_float2(float(ci + 1), 1.0f);
	}
	const float2 mpL = src.mapAt(min(b + L, M - 1));
	float tfUp = hd.timeFactor, tfDn = hd.timeFactor;
	if (randomTf) { // uniform(4 - tf, tf): the upward steps of bin b take draw 2b - 1 of the hop, the downward steps draw 2b (:640,:749,:769)
		const float lo = 4.0f - hd.timeFactor;
		if (b > 0) tfUp = engineDraw(d, hd.seed, 2*b - 1, lo, hd.timeFactor);
		if (b < M - 1) tfDn = engineDraw(d, hd.seed, 2*b, lo, hd.timeFactor);
	}
	auto twists = [&](int cm, float2 Pcm, float2 &A, float2 &B, float2 &Cc, float2 &Dc) {
		const float2 *in = src.inRow(cm);
		const float2 *pv = prevRow(d, hd, s, k, sg, cm);
		// Prediction.energy of the previous hop: the carried state for the tile's first hop, else hop k-1's
		PrevEnergy prevE;
		prevE.carried = k == 0;
		prevE.carriedBase = stateRow(d, sg, cm);
		prevE.input = (PLAIN && k > 0) ? inputRow(d, hp, s, sg, cm) : nullptr;
		prevE.row = (!PLAIN && k > 0) ? reinterpret_cast<const float *>(d.PE + rowOf(d, s, k - 1, cm)) + 2 : nullptr;
		const float2 zero = make_float2(0.f, 0.f);
		A = cmulc(Pcm, lerpBand(in, lerpIndex(mp.x - tfUp), M));
		B = cmulc(Pcm, lerpBand(in, lerpIndex(mp.x - L*tfUp), M));
		Cc = twistAt<CH, PLAIN>(src, cm, b + 1, mp1, rotate, in, pv, prevE, tfDn, 1.0f, rot);
		Dc = twistAt<CH, PLAIN>(src, cm, b + L, mpL, rotate, in, pv, prevE, tfDn, float(L), rot);
		if (!(b > 0)) A = zero;      // :748
		if (!(b >= L)) B = zero;     // :756
		if (!(b < M - 1)) Cc = zero; // :765
		if (!(b < M - L)) Dc = zero; // :776
	};
	float2 A, B, Cc, Dc;
	if (SPEC) {
		twists(0, p[0], A, B, Cc, Dc);
#pragma unroll
		for (int c = 1; c < CH; ++c) {
			float2 a2, b2, c2, d2;
			twists(c, p[c], a2, b2, c2, d2);
			if (c == mc) { A = a2; B = b2; Cc = c2; Dc = d2; }
		}
	} else {
		twists(mc, Pm, A, B, Cc, Dc);
	}
	if constexpr (FOLD0) {
		if (k == 0) foldCarriedTaps(carriedOutput(d, sg), mc, b, M, L, Cc, Dc);
	}
	f[0] = A.x; f[1] = A.y; f[2] = B.x; f[3] = B.y; f[4] = Cc.x; f[5] = Cc.y; f[6] = Dc.x; f[7] = Dc.y;
	f[8] = __int_as_float(mc);
	static_assert(!LOCK, "the separate lock-twist fields are gone: stereo records carry the scaled twist (recordChannelFields)");
	recordChannelFields<CH>(f, p, e, mc);
}

// One workgroup = 8 wavefront steps x all 64 hops of a stream.  Reads are coalesced along the bin index (8 lanes
// per row); the 512 records are transposed through LDS so that the stores to the skewed array are contiguous 1-KiB
// rows (the scattered 16-byte stores of the first version ran at 1.3 TB/s and dominated the whole pipeline).
// (Used for more than 2 channels; mono/stereo use the fused kVocoder below, which never writes records to HBM.)
template <int CH, bool PLAIN>
__global__ __launch_bounds__(256) void kPredictB(DevBatch d, int sBase, int hopBase) {
	constexpr int NF = 9 + 3*CH, NCH = (NF + 3)/4;
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float4 *tile = reinterpret_cast<float4 *>(smemRaw); // [(st*NCH + j)*65 + k]
	const int s = blockIdx.y, sg = sBase + s;
	const int T0 = blockIdx.x*8;
	const int M = d.M;
	const int nh = d.nHops[s];
	if (nh == 0 || T0 >= M + d.lag*(nh - 1)) return; // nothing of this stream's wavefront in these steps
	const int st = threadIdx.x & 7, r = threadIdx.x >> 3;
	const int t = T0 + st;
#pragma unroll
	for (int half = 0; half < 2; ++half) {
		const int k = r + 32*half;
		const int b = t - d.lag*k;
		float f[NCH*4];
#pragma unroll
		for (int j = 0; j < NCH*4; ++j) f[j] = 0.0f;
		if (k < nh && b >= 0 && b < M) {
			const HopDesc hd = d.hops[(size_t)sg*d.hopStride + hopBase + k];
			const HopDesc hp = d.hops[(size_t)sg*d.hopStride + hopBase + (k > 0 ? k - 1 : 0)];
			computeRecord<CH, PLAIN, false, false>(d, hd, hp, s, sg, k, b, f);
		}
#pragma unroll
		for (int j = 0; j < NCH; ++j) tile[(st*NCH + j)*65 + k] = make_float4(f[4*j], f[4*j + 1], f[4*j + 2], f[4*j + 3]);
	}
	__syncthreads();
	float4 *rec = d.REC + ((size_t)s*d.recSteps + T0)*(size_t)d.recPitch;
#pragma unroll
	for (int n = 0; n < 2*NCH; ++n) { // 8*NCH*64 float4 per tile / 256 threads
		const int idx = threadIdx.x + 256*n;
		const int row = idx >> 6, k = idx & 63; // row = st*NCH + j
		const int stw = row/NCH, j = row - stw*NCH;
		rec[(size_t)stw*d.recPitch + j*64 + k] = tile[row*65 + k];
	}
}

// ------------------------------------------------------------------------------------------------------
// K3: the bin recurrence (main prediction + channel locking, signalsmith-stretch.h:722-803) as a skewed
// wavefront.  One wave per stream; lane k = hop k of the tile; at step t lane k finalises bin t - lag*k.
// Hop k at bin b needs hop k's own outputs at b-1 and b-L, and hop k-1's FINAL outputs at b+1 and b+L
// (they enter through the preliminary prediction of hop k); lag >= L+1 guarantees they exist.  Outputs of the
// last `ringSlots` bins of every lane live in an LDS ring that the next lane reads; lane 0 reads the carried
// Band.output state, staged through LDS 64 bins at a time with one coalesced load per channel.  The per-step
// records are prefetched PD steps ahead with fully coalesced 1-KiB wave loads, so no global-memory latency sits
// on the serial path.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 makeOutput(float2 phase, float2 input, float sqrtEnergy) { // :596-603
	// branch-free: the fallback (prediction too weak -> use the input's phase) is a select, not a divergent branch
	const float n = cnorm(phase);
	const bool weak = n <= 1e-15f;
	const float nIn = cnorm(input) + 1e-15f;
	const float2 ph = weak ? input : phase;
	const float g = sqrtEnergy*__builtin_amdgcn_rsqf(weak ? nIn : n);
	return cscale(ph, g);
}

// stereo: the same with the fallback value precomputed by the record producer (recordChannelFields): two selects instead of
// a norm, an add, three selects on the serial path
__device__ __forceinline__ float2 makeOutputFb(float2 phase, float2 fallback, float sqrtEnergy) {
	const float n = cnorm(phase);
	const float2 o = cscale(phase, sqrtEnergy*__builtin_amdgcn_rsqf(n)); // n == 0: inf / nan, discarded by the select
	return (n <= 1e-15f) ? fallback : o;
}

template <int CH>
__global__ __launch_bounds__(64) void kChain(DevBatch d, int sBase, int hopBase) {
	constexpr int NF = 9 + 3*CH, NCH = (NF + 3)/4;
	constexpr int PD = 4; // prefetch depth (one wave per SIMD slot: the register file is not the limit)
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	const int R = d.ringSlots, Rm = R - 1;
	float2 *lds = reinterpret_cast<float2 *>(smemRaw); // ring [CH][R][64], then stage [CH][128]
	const int stageBase = CH*R*64;                      // carried Band.output of the previous tile, 128-bin window

	const int s = blockIdx.x, sg = sBase + s, k = threadIdx.x;
	const int nh = d.nHops[s];
	if (nh == 0) return;
	__builtin_amdgcn_s_setprio(3); // serial path of the whole pipeline: win issue arbitration against co-resident bulk waves
	const int M = d.M, L = d.L, lag = d.lag;
	const bool active = k < nh;
	const float4 *rec = d.REC + (size_t)s*d.recSteps*(size_t)d.recPitch + k;
	const size_t recPitch = d.recPitch;
	float2 *OUT = d.OUT + rowOf(d, s, active ? k : 0, 0);
	float2 *dump = d.dump + (size_t)s*CH*64 + k; // where lanes outside their bin range park their stores
	const CarriedOutput stOut = carriedOutput(d, sg);

	for (int i = k; i < CH*R*64; i += 64) lds[i] = make_float2(0.f, 0.f);
	for (int c = 0; c < CH; ++c) { // prologue: stage bins [0,128) of the carried output
		lds[stageBase + c*128 + k] = (k < M) ? stOut[(size_t)c*M + k] : make_float2(0.f, 0.f);
		lds[stageBase + c*128 + 64 + k] = (64 + k < M) ? stOut[(size_t)c*M + 64 + k] : make_float2(0.f, 0.f);
	}
	float2 pf[CH];
	float2 own1[CH]; // this lane's outputs at bin b-1
#pragma unroll
	for (int c = 0; c < CH; ++c) { pf[c] = make_float2(0.f, 0.f); own1[c] = make_float2(0.f, 0.f); }

	const int steps = M + lag*(nh - 1);
	float4 q[PD][NCH];
#pragma unroll
	for (int u = 0; u < PD; ++u) {
#pragma unroll
		for (int j = 0; j < NCH; ++j) q[u][j] = rec[(size_t)u*recPitch + j*64];
	}
	__syncthreads();

	const int chunks = (steps + 63) >> 6; // REC is padded, so running to the end of the last 64-step chunk is safe
	for (int ch = 0; ch < chunks; ++ch) {
		const int tb = ch << 6;
		// bins [tb+64, tb+128) were fetched one chunk ago: publish them, then fetch [tb+128, tb+192)
		if (ch > 0) {
#pragma unroll
			for (int c = 0; c < CH; ++c) lds[stageBase + c*128 + ((tb + 64 + k) & 127)] = pf[c];
		}
		{
			const int bb = tb + 128 + k;
			const int bc = (bb < M) ? bb : M - 1;
#pragma unroll
			for (int c = 0; c < CH; ++c) {
				float2 v = stOut[(size_t)c*M + bc];
				pf[c] = (bb < M) ? v : make_float2(0.f, 0.f);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		for (int i = 0; i < 64/PD; ++i) {
#pragma unroll
			for (int u = 0; u < PD; ++u) {
				const int t = tb + i*PD + u;
				float f[NCH*4];
#pragma unroll
				for (int j = 0; j < NCH; ++j) { f[4*j] = q[u][j].x; f[4*j + 1] = q[u][j].y; f[4*j + 2] = q[u][j].z; f[4*j + 3] = q[u][j].w; }
				const int b = t - lag*k;
				const bool valid = active && b >= 0 && b < M;
				int mc = __float_as_int(f[8]);
				mc = (mc < 0) ? 0 : ((mc > CH - 1) ? CH - 1 : mc); // records of out-of-range steps are not initialised
				float2 o1 = own1[0], pm = make_float2(f[9], f[10]);
				float sm = f[11];
#pragma unroll
				for (int c = 1; c < CH; ++c) {
					if (c == mc) {
						o1 = own1[c];
						if (CH != 2) { pm = make_float2(f[9 + 3*c], f[10 + 3*c]); sm = f[11 + 3*c]; } // stereo records lead with the maximum channel
					}
				}
				const int ringRow = mc*R;
				const float2 oL = lds[(ringRow + ((b - L) & Rm))*64 + k];
				const int a1 = (k == 0) ? stageBase + mc*128 + ((b + 1) & 127) : (ringRow + ((b + 1) & Rm))*64 + k - 1;
				const int aL = (k == 0) ? stageBase + mc*128 + ((b + L) & 127) : (ringRow + ((b + L) & Rm))*64 + k - 1;
				const float2 p1 = lds[a1];
				const float2 pL = lds[aL];
				float2 phi = prevHopTerms(p1, make_float2(f[4], f[5]), pL, make_float2(f[6], f[7])); // previous hop's part first (what FOLD0 records pre-compute)
				phi = cfma(oL, make_float2(f[2], f[3]), phi);
				phi = cfma(o1, make_float2(f[0], f[1]), phi); // the newest operand last: two dependent instructions behind it
				const float2 om = (CH == 2) ? makeOutputFb(phi, pm, sm) : makeOutput(phi, pm, sm); // :788 (stereo records carry the fallback output in pm's place)
				const float2 olock = (CH == 2) ? lockedOutput(om, f) : om; // stereo: see recordChannelFields
#pragma unroll
				for (int c = 0; c < CH; ++c) {
					float2 oc;
					if constexpr (CH == 2) {
						oc = olock;
					} else {
						const float2 pc = make_float2(f[9 + 3*c], f[10 + 3*c]);
						oc = makeOutput(cmul(om, cmulc(pc, pm)), pc, f[11 + 3*c]); // channel lock, :791-800
					}
					if (c == mc) oc = om;
					if (!valid) oc = make_float2(0.f, 0.f);
					own1[c] = oc;
					lds[(c*R + (b & Rm))*64 + k] = oc;
					float2 *dst = valid ? OUT + ((size_t)c*d.Mp + b) : dump + c*64;
					*dst = oc;
				}
				// refill this slot for step t + PD only now: the old contents are dead, so the new load can reuse
				// the same registers and nothing has to be copied (or waited for) at the loop back-edge
#pragma unroll
				for (int j = 0; j < NCH; ++j) q[u][j] = rec[(size_t)(t + PD)*recPitch + j*64];
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------------
// K3 fused (mono / stereo): the recurrence and its coefficients in ONE kernel, so the records never touch HBM.
// One workgroup of 16 waves per stream: wave 0 is the CONSUMER (the skewed wavefront, one lane per hop), wave 4 the
// WRITER (results -> HBM), and 8 (staged) or 14 (gathering) of the others are PRODUCERS that compute the records (same
// arithmetic as kPredictB, 8 rows x 8 steps per wave-pass) into an LDS ring of 2 or 3 blocks x 8 steps; unused waves
// retire at once.  Hand-off is by LDS counters (units produced per slot, blocks consumed, result blocks ready /
// written); LDS operations of a wave execute in order, so a counter update issued after the data writes is seen after them.
//
// The consumer keeps the serial path in registers: each lane holds its last 8 outputs per channel (h[t & 7]), so
//   own taps      out[b-1], out[b-L]                    = h[(i+7)&7], h[(i+8-L)&7]
//   previous hop  out_{k-1}[b+1], out_{k-1}[b+L]        = the SAME two registers of lane k-1 (it runs lag = L+1 bins
//                                                         ahead), fetched with one DPP wave_shr:1 each
// and only lane 0 (whose "previous hop" is the carried Band.output) takes them from the staged LDS copy, read one step
// ahead and passed as the DPP's `old` operand.  No LDS round trip and no memory load sits on the recurrence.
// ------------------------------------------------------------------------------------------------------
// Hand-off words in LDS: relaxed workgroup-scope atomics.  (A `volatile` access makes the backend drain EVERY
// outstanding memory operation -- s_waitcnt vmcnt(0) -- around it, which serialised the producers' prefetch loads
// behind each poll.)  Ordering against the data they guard comes from the in-order LDS pipe plus compiler barriers.
__device__ __forceinline__ int ldsPeek(volatile int *p) { return __hip_atomic_load(const_cast<int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ldsPost(volatile int *p, int v) { __hip_atomic_store(const_cast<int *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ldsCount(volatile int *p) { (void)__hip_atomic_fetch_add(const_cast<int *>(p), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Staged producers (PLAIN tiles without random time factors, L <= 5).  Measured on the first version of this kernel
// (profiles/r1_pmc_vocoder_ta.json): the texture-address unit was busy 76% of the kernel -- every record issued 14
// narrow gathers (8 rows x 64 B each).  Here one producer wave owns 8 fixed rows; per 8-step block it fetches the
// rows' windows once with 16-byte loads (IN: bins b0-2L..b0+7+L of every channel, PV and ROT: b0+1..b0+7+L, plus the
// row above its first row for Prediction.energy of the previous hop), parks them in a private LDS buffer, and computes
// its 64 records from LDS.  The loads of block n+1 are in flight while block n is computed.
template <int CH, int L>
struct StageGeom {
	static constexpr int PIN = (8 + 3*L + 1)/2;  // 16-byte pieces (2 bins) of one IN window
	static constexpr int PPV = (7 + L + 1)/2;    // pieces of one PV / ROT window
	static constexpr int PV_OFF = CH*2*PIN, ROT_OFF = PV_OFF + CH*2*PPV, ROWUSED = ROT_OFF + 2*PPV; // float2 units
	// row pitch = 8 (mod 16) float2: the 16 lanes an LDS cycle serves are two rows x eight consecutive bins, and with this
	// pitch the two rows fall into the two halves of the 32 banks (76 float2 put rows r and r+4 on the same banks: every
	// read of the record computation two-way conflicted; SQ_LDS_BANK_CONFLICT 143 M cycles per launch)
	static constexpr int ROWLEN = ((ROWUSED + 7)/16)*16 + 8;
	static constexpr int ROW_PIECES = CH*PIN + CH*PPV + PPV;
	static constexpr int X_FIRST = L/2, X_PIECES = 3 + L - L/2 + 1; // extra row (hop above): window indices [L, 6+2L]
	static constexpr int TOTAL = 8*ROW_PIECES + CH*X_PIECES;
	static constexpr int LOADS = (TOTAL + 63)/64;
	static constexpr int ROWS = 9; // local rows -1..7
};

// two adjacent entries of the carried Prediction.energy, by element index (fp32: one 8-byte load)
__device__ __forceinline__ float2 loadEnergyPair(const DevBatch &d, size_t e) {
	if (d.halfState) return make_float2(loadCarriedEnergy(d, e), loadCarriedEnergy(d, e + 1));
	return *reinterpret_cast<const float2 *>(d.stEnergy + e);
}

template <int CH, int L, int NB, int NP>
__device__ __forceinline__ void vocoderProduceStaged(const DevBatch &d, int s, int sg, int nh, int pIndex, int k, int totalBlocks,
                                                     float4 *recs, volatile int *sync, const HopDesc *hopsLds, float2 *sbuf, const CarriedOutput &stOut) {
	using G = StageGeom<CH, L>;
	constexpr int NF = 9 + 3*CH, NCH = (NF + 3)/4, BS = 8, lag = L + 1;
	static_assert(NP%8 == 0, "one producer wave per group of 8 rows");
	const int M = d.M;
	const int it = pIndex & 7;
	// ---- block-invariant description of this lane's pieces
	const float2 *psrc[G::LOADS];
	int pbin[G::LOADS], plds[G::LOADS];
	bool pok[G::LOADS], pen[G::LOADS]; // piece wanted / piece is carried Prediction.energy (row above the tile's first hop)
	const size_t carriedEnergy = stateRow(d, sg, 0); // element index of the stream's carried Prediction.energy, [C][M]
	size_t penergy = carriedEnergy;
#pragma unroll
	for (int i = 0; i < G::LOADS; ++i) {
		const int q = k + 64*i;
		int rl, j;
		if (q < 8*G::ROW_PIECES) { rl = q/G::ROW_PIECES; j = q%G::ROW_PIECES; }
		else { const int x = q - 8*G::ROW_PIECES; rl = -1; j = (x/G::X_PIECES)*G::PIN + G::X_FIRST + x%G::X_PIECES; }
		const int row = 8*it + rl;
		const bool energy = q < G::TOTAL && row == -1; // the hop above row 0 is the carried state: stage its energy as (E, 0)
		const bool ok = q < G::TOTAL && row >= -1 && row < nh;
		const HopDesc hd = hopsLds[row >= 0 && ok ? row : 0];
		const float2 *src;
		int rel, off;
		if (j < CH*G::PIN) { // IN
			const int c = j/G::PIN, pp = j%G::PIN;
			src = energy ? d.rot : inputRow(d, hd, s, sg, 0) + (size_t)c*((hd.inSrc >= 0) ? d.Mp : d.M); // energy: dummy address for the wide load
			if (energy) penergy = carriedEnergy + (size_t)c*M;
			rel = -2*L + 2*pp;
			off = c*2*G::PIN + 2*pp;
		} else if (j < CH*G::PIN + CH*G::PPV) { // PV
			const int jj = j - CH*G::PIN, c = jj/G::PPV, pp = jj%G::PPV;
			src = prevRow(d, hd, s, row, sg, c);
			rel = 1 + 2*pp;
			off = G::PV_OFF + c*2*G::PPV + 2*pp;
		} else { // ROT
			const int pp = j - CH*G::PIN - CH*G::PPV;
			src = d.rot;
			rel = 1 + 2*pp;
			off = G::ROT_OFF + 2*pp;
		}
		psrc[i] = ok ? src : d.rot;
		pbin[i] = rel - lag*row;
		plds[i] = (rl + 1)*G::ROWLEN + off;
		pok[i] = ok;
		pen[i] = energy;
	}
	static_assert(64*(G::LOADS - 1) <= 8*G::ROW_PIECES, "pieces of the row above sit in the last load slot");
	float4 v[G::LOADS];
	float2 ve = make_float2(0.f, 0.f);
	// A block whose windows all lie strictly inside [0, M-2] (nine blocks in ten) needs no clamping on the way in and no
	// edge selects on the way to LDS.  The bins a wave touches in block n span rows 8*it-1 .. 8*it+7 and window offsets
	// -2L .. 7+L, so the test is wave-uniform: a scalar branch, no vote.
	auto interior = [&](int n) {
		const int lo = BS*n - 2*L - lag*(8*it + 7), hi = BS*n + 7 + L + 1 - lag*(8*it - 1);
		return lo >= 0 && hi <= M - 2;
	};
	auto issue = [&](int n) {
		if (interior(n)) {
#pragma unroll
			for (int i = 0; i < G::LOADS; ++i) {
				// lanes without a piece (beyond TOTAL, rows beyond the tile's hops) load from the start of the rotation table: their
				// piece description may point a few bins past a row's end (found by the address sanitiser on the CPU stand-in)
				const int sb = pok[i] ? BS*n + pbin[i] : 0;
				v[i] = *reinterpret_cast<const float4 *>(psrc[i] + sb); // 8-byte aligned; dword alignment suffices on gfx9
				if (i == G::LOADS - 1 && pen[i]) ve = loadEnergyPair(d, penergy + sb);
			}
			return;
		}
#pragma unroll
		for (int i = 0; i < G::LOADS; ++i) {
			const int sb = BS*n + pbin[i];
			const int cb = min(max(sb, 0), M - 2);
			v[i] = *reinterpret_cast<const float4 *>(psrc[i] + cb); // 8-byte aligned; dword alignment suffices on gfx9
			// pieces of the carried energy (row above hop 0; only the last slot can hold them) are 2 floats: kept in their own
			// registers until park(), so that no select waits for the loads here
			if (i == G::LOADS - 1 && pen[i]) ve = loadEnergyPair(d, penergy + cb);
		}
	};
	auto park = [&](int n) {
		if (interior(n)) {
#pragma unroll
			for (int i = 0; i < G::LOADS; ++i) {
				const bool en = (i == G::LOADS - 1) && pen[i];
				if (pok[i]) *reinterpret_cast<float4 *>(sbuf + plds[i]) = en ? make_float4(ve.x, 0.f, ve.y, 0.f) : v[i];
			}
			return;
		}
#pragma unroll
		for (int i = 0; i < G::LOADS; ++i) {
			const int sb = BS*n + pbin[i];
			const int delta = sb - min(max(sb, 0), M - 2); // 0 in range; -1: first bin is -1; +1: first bin is M-1; else both outside
			// selects, not branches: lo = v.xy / v.zw / 0 for delta 0 / +1 / other; hi = v.zw / v.xy / 0 for delta 0 / -1 / other
			const bool d0 = delta == 0, dp = delta == 1, dm = delta == -1;
			const bool en = (i == G::LOADS - 1) && pen[i];
			const float wx = en ? ve.x : v[i].x, wy = en ? 0.f : v[i].y, wz = en ? ve.y : v[i].z, ww = en ? 0.f : v[i].w;
			float lx = dp ? wz : 0.f, ly = dp ? ww : 0.f, hx = dm ? wx : 0.f, hy = dm ? wy : 0.f;
			if (d0) { lx = wx; ly = wy; hx = wz; hy = ww; }
			if (pok[i]) *reinterpret_cast<float4 *>(sbuf + plds[i]) = make_float4(lx, ly, hx, hy);
		}
	};
	const int st = k & 7, r = k >> 3, row = 8*it + r;
	const HopDesc hd = hopsLds[row < nh ? row : 0];
	const bool rotate = hd.flags & HOP_NEW_SPECTRUM;
	const float tf = hd.timeFactor;
	const float2 *mine = sbuf + (r + 1)*G::ROWLEN, *above = sbuf + r*G::ROWLEN;
	constexpr int NPB = NP/8;
	// Hop 0's previous-hop taps are the carried Band.output (FOLD0, see computeRecord): the wave that owns row 0 fetches them with
	// its windows, one block ahead (lanes 0..7 = row 0, steps 0..7; the other lanes load in-range values they never use)
	float2 car1[CH], carL[CH], carNext1[CH], carNextL[CH];
#pragma unroll
	for (int c = 0; c < CH; ++c) car1[c] = carL[c] = carNext1[c] = carNextL[c] = make_float2(0.f, 0.f);
	auto issueCarried = [&](int nn) {
		const int b = BS*nn + st; // row 0: no skew
#pragma unroll
		for (int c = 0; c < CH; ++c) {
			carNext1[c] = stOut[(size_t)c*M + min(b + 1, M - 1)];
			carNextL[c] = stOut[(size_t)c*M + min(b + L, M - 1)];
		}
	};
	int n = pIndex >> 3;
	if (n < totalBlocks) { issue(n); if (it == 0) issueCarried(n); }
	for (; n < totalBlocks; n += NPB) {
		park(n);
		if (it == 0) {
#pragma unroll
			for (int c = 0; c < CH; ++c) { car1[c] = carNext1[c]; carL[c] = carNextL[c]; }
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (n + NPB < totalBlocks) { issue(n + NPB); if (it == 0) issueCarried(n + NPB); }
		const int slot = n%NB;
		// (waiting only before the store, as the gathering producers do, was slower here: 8.2 -> 8.5 ms per step -- records
		// computed early take issue slots from the recurrence wave exactly when it is not waiting for them)
		while (n - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2); // slot still being read
		asm volatile("" ::: "memory");
		const int b0 = BS*n - lag*row, b = b0 + st;
		float f[NCH*4];
#pragma unroll
		for (int j = 0; j < NCH*4; ++j) f[j] = 0.0f;
		if (row < nh && b >= 0 && b < M && !SMST_SKIP_PRODUCER_MATH(d)) {
			// same arithmetic as computeRecord<CH, true, false, false>, operands from the staged windows
			auto IN = [&](int c, int x) { return mine[c*2*G::PIN + (x - b0 + 2*L)]; };
			auto lerpIN = [&](int c, LerpIndex li) {
				const float2 low = IN(c, li.lo), high = IN(c, li.lo + 1);
				return clerp(low, high, li.fr);
			};
			float2 p[CH];
			float e[CH];
#pragma unroll
			for (int c = 0; c < CH; ++c) { p[c] = IN(c, b); e[c] = cnorm(p[c]); }
			int mc = 0;
			float eMax = e[0];
#pragma unroll
			for (int c = 1; c < CH; ++c) if (e[c] > eMax) { mc = c; eMax = e[c]; }
			float2 Pm = p[0];
#pragma unroll
			for (int c = 1; c < CH; ++c) if (c == mc) Pm = p[c];
			const float fb = float(b);
			float2 A = cmulc(Pm, lerpIN(mc, lerpIndex(fb - tf)));
			float2 B = cmulc(Pm, lerpIN(mc, lerpIndex(fb - L*tf)));
			auto twist = [&](int bx, float stepMul) {
				const int bc = min(bx, M - 1);
				const float2 rotB = rotate ? mine[G::ROT_OFF + (bx - b0 - 1)] : make_float2(1.f, 0.f);
				const float2 Q = cmul(mine[G::PV_OFF + mc*2*G::PPV + (bx - b0 - 1)], rotB);
				const float2 Px = IN(mc, bx);
				const float2 TW = cmul(rotB, cmulc(Px, Q));
				const float eNow = cnorm(Px);
				// Prediction.energy of the previous hop: hop row-1's input (its window starts lag bins later), or the carried state
				const float2 up = above[mc*2*G::PIN + (bx - b0 - lag + 2*L)];
				const float ePrev = (row > 0) ? cnorm(up) : up.x;
				const float den = fmaxf(ePrev, eNow) + 1e-15f;
				const float2 down = cmulc(Px, lerpIN(mc, lerpIndex(float(bc) - stepMul*tf)));
				const float2 rr = cmulc(TW, down);
				const float inv = __builtin_amdgcn_rcpf(den); // 1-ulp hardware reciprocal (an IEEE division costs ten instructions per record)
				return make_float2(rr.x*inv, rr.y*inv);
			};
			float2 Cc = twist(b + 1, 1.0f), Dc = twist(b + L, float(L));
			const float2 zero = make_float2(0.f, 0.f);
			if (!(b > 0)) A = zero;
			if (!(b >= L)) B = zero;
			if (!(b < M - 1)) Cc = zero;
			if (!(b < M - L)) Dc = zero;
			if (it == 0) { // FOLD0: row 0's record carries the previous-hop part ready-made (wave-uniform branch, lane select inside)
				float2 c1 = car1[0], cL = carL[0];
#pragma unroll
				for (int c = 1; c < CH; ++c) if (c == mc) { c1 = car1[c]; cL = carL[c]; }
				const float2 K = prevHopTerms(c1, Cc, cL, Dc);
				if (r == 0) { Cc = K; Dc = zero; }
			}
			f[0] = A.x; f[1] = A.y; f[2] = B.x; f[3] = B.y; f[4] = Cc.x; f[5] = Cc.y; f[6] = Dc.x; f[7] = Dc.y;
			f[8] = __int_as_float(mc);
			recordChannelFields<CH>(f, p, e, mc);
		}
#pragma unroll
		for (int j = 0; j < NCH; ++j) recs[((slot*BS + st)*NCH + j)*64 + ((row + st) & 63)] = make_float4(f[4*j], f[4*j + 1], f[4*j + 2], f[4*j + 3]);
		asm volatile("" ::: "memory");
		if (k == 0) ldsCount(&sync[slot]);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier(); // every lane has read its operands before the next block's windows are parked
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
}

// Line-aligned staged producers (PLAIN tiles without random time factors, L <= 4, M a multiple of 16), wavefront lag 8.
// What bounded the staged kernel above was the CU's L1-miss line rate: with lag = L+1 every row's
// windows sit at their own odd alignment, a 160-byte IN window touches 2-3 lines of which it needs 64 new bytes, and each
// line comes through L1 again in three or four consecutive blocks -- ~600 lines per 8-step block.  With a lag of EIGHT bins
// row r covers bins b0 = 8(n-r) .. b0+7 in block n: every row advances by exactly half a 128-byte line per block, in step.
// So each (row, array) keeps the two lines around the row's current bins in LDS, LINEARLY (32 bins: lines lo, lo+1), and a
// line is fetched from memory exactly ONCE, whole and aligned, in the block before its first use:
//   IN  needs bins b0-2L .. b0+7+L  (within [b0-8, b0+11]):  lines j-1, j at b0 = 16j; lines j, j+1 at b0 = 16j+8
//   PV  needs bins b0+1  .. b0+7+L                         :  line  j      at b0 = 16j; lines j, j+1 at b0 = 16j+8
// i.e. in the blocks with m = n - row odd (m = -1 brings line 0) a row moves its upper line to the lower half of the buffer
// and parks line (m+1)/2 of every array in the upper half -- every lane moves and parks its own 16-byte piece, so no lane
// reads what another one writes.  The bins of a block then sit at  base + (x - b0)  with base = 16 + st in a row's even blocks
// and 8 + st in its odd ones: ONE select per block, after which every operand of a record is an immediate offset from that
// base (a ring indexed by x & 31 cost three VALU instructions per LDS read, ~70 per record: the first form of this function
// was 50 % SLOWER than the staged producers above for all its saved loads -- the CU's VALU issue is the shared limit).
// A producer wave owns 8 rows; per block 4 of them take a new line of 2*CH arrays: 8*CH lines = CH 16-byte loads per lane,
// each instruction 8 whole lines (the lag-(L+1) form: 6 loads per lane and block, ~75 lines per wave).  The rotation factors
// of a lane's two previous-hop bins travel in registers (the table's active region is 4 KB and stays in L1); the row above a
// wave's first row (Prediction.energy of the previous hop, owned by the neighbouring wave) is staged as the 16 bins its block
// needs.  The 8-bin lag costs 63*3 more steps per tile (+5.6 %) and puts rows r and r+1 on complementary halves of the LDS
// banks with no row padding.  Same operands, same operations in the same order as vocoderProduceStaged / computeRecord:
// bit-identical records.
template <int CH, int L>
struct AlignGeom {
	static constexpr int RING = 32;                   // bins per (row, array): two lines
	static constexpr int ROWLEN = 2*CH*RING;          // float2 per row: CH input buffers, then CH previous-input buffers
	static constexpr int XLEN = CH*16;                // the row above the wave's first row: 16 bins per channel
	static constexpr int PER_PRODUCER = 8*ROWLEN + XLEN;
	static constexpr int LOADS = CH;                  // (4 rows x 2*CH arrays x 8 pieces) / 64 lanes
};

template <int CH, int L, int NB, bool FIRST>
__device__ __forceinline__ void vocoderProduceAligned(const DevBatch &d, int s, int sg, int nh, int it, int k, int totalBlocks,
                                                      float4 *recs, volatile int *sync, const HopDesc *hopsLds, float2 *sbuf, const CarriedOutput &stOut) {
	// FIRST: the wave that owns rows 0..7 (it == 0): the hop above its first row is the carried state, its row 0 folds the carried
	// Band.output into its records (FOLD0), and it has a parking-only block before the tile's first one
	using G = AlignGeom<CH, L>;
	constexpr int NF = 9 + 3*CH, NCH = (NF + 3)/4, BS = 8;
	static_assert(2*L <= 8 && 7 + L <= 11, "the windows must fit the two lines around the row's bins");
	const int M = d.M, lines = M >> 4;
	float2 *xbuf = sbuf + 8*G::ROWLEN;
	for (int i = k; i < G::PER_PRODUCER/2; i += 64) reinterpret_cast<float4 *>(sbuf)[i] = make_float4(0.f, 0.f, 0.f, 0.f); // bins below 0 read as zero
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	// ---- block-invariant description of this lane's line pieces: [parity of the block][load]
	const float2 *lsrc[2][G::LOADS];
	int llds[2][G::LOADS], lrow[2][G::LOADS];
#pragma unroll
	for (int par = 0; par < 2; ++par) {
#pragma unroll
		for (int i = 0; i < G::LOADS; ++i) {
			const int q = k + 64*i, li = q >> 3, piece = q & 7;
			const int rr = li/(2*CH), a = li%(2*CH);
			const int r = 2*rr + par, row = 8*it + r;
			const bool ok = row < nh;
			const HopDesc hd = hopsLds[ok ? row : 0];
			const float2 *src = (a < CH) ? inputRow(d, hd, s, sg, a) : prevRow(d, hd, s, row, sg, a - CH);
			lsrc[par][i] = (ok ? src : d.rot) + 2*piece;
			llds[par][i] = r*G::ROWLEN + a*G::RING + 2*piece; // this lane's piece of the LOWER line; the upper one is 16 bins on
			lrow[par][i] = ok ? r : (1 << 20); // a row beyond the tile's hops never reaches m >= -1
		}
	}
	// the row above this wave's first row: hop 8*it - 1 of the tile, or (FIRST) the carried Prediction.energy (fp32: the launcher keeps
	// batches with fp16 state on the staged producers above)
	const int xc = (k >> 3) < CH ? (k >> 3) : 0, xpiece = k & 7;
	const bool xlane = k < 8*CH;
	const float2 *xsrc = d.rot;
	if (!FIRST) xsrc = inputRow(d, hopsLds[8*it - 1], s, sg, xc);
	const float *xenergy = d.stEnergy + stateRow(d, sg, xc);
	const float2 *carried = static_cast<const float2 *>(stOut.base); // [CH][M]
	// Loads are requested through smst_async.h and waited for by COUNT (loads return in order): the lines of block n+2, the small
	// loads of block n+1 (row above, rotation factors, FIRST: carried taps) are requested during block n, in the order
	//   ... lines(n) | small(n) lines(n+1) | small(n+1) lines(n+2) ...
	// so at the top of block n everything but the LOADS youngest requests -- lines(n+1) -- has to have landed.  A line comes from HBM
	// and a block lasts ~2 us: with one block of lead the producers waited for their loads a third of every block (cycle trace,
	// tools/probes/voc_trace_patch_aligned.py), and left to the compiler the waits degenerate to vmcnt(0) behind the branches of
	// this loop.  Two register sets, one per block parity: the set parked in block n is free for block n+2's lines.
	Async16 vE[G::LOADS], vO[G::LOADS], xv;
	Async8 xe, rotNext1, rotNextL, carNext1[CH], carNextL[CH];
	const int st = k & 7, r = k >> 3, row = 8*it + r;
	float2 rot1 = make_float2(1.f, 0.f), rotL = rot1;
	// PAR = (n + 1) & 1: the rows with that parity take a new line in block n; m = n - row is odd
	auto lineOf = [&](int n, int par, int i) { const int m = n - 8*it - lrow[par][i]; return m >= -1 ? (m + 1) >> 1 : -1; };
	auto issueLines = [&](int n, int par, Async16 (&v)[G::LOADS]) {
#pragma unroll
		for (int i = 0; i < G::LOADS; ++i) {
			const int jc = min(max(lineOf(n, par, i), 0), lines - 1);
			asyncLoad16(v[i], lsrc[par][i] + 16*jc);
		}
	};
	auto issueSmall = [&](int n) { // 3 requests (FIRST: 3 + 2*CH), every address clamped into its row: n may run past the tile's last block
		const int x0 = BS*(n - 8*it) + 2*xpiece, xcl = min(max(x0, 0), M - 2);
		if (FIRST) asyncLoad8(xe, xenergy + xcl);
		else asyncLoad16(xv, xsrc + xcl);
		const int b = BS*(n - row) + st;
		asyncLoad8(rotNext1, d.rot + min(max(b + 1, 0), M - 1));
		asyncLoad8(rotNextL, d.rot + min(max(b + L, 0), M - 1));
		if (FIRST) { // hop 0's previous-hop taps are the carried Band.output (FOLD0): lanes 0..7 = row 0 (no skew), steps 0..7
			const int b0 = min(max(BS*n + st, 0), M - 1);
#pragma unroll
			for (int c = 0; c < CH; ++c) {
				asyncLoad8(carNext1[c], carried + (size_t)c*M + min(b0 + 1, M - 1));
				asyncLoad8(carNextL[c], carried + (size_t)c*M + min(b0 + L, M - 1));
			}
		}
	};
	float2 car1[CH], carL[CH];
#pragma unroll
	for (int c = 0; c < CH; ++c) car1[c] = carL[c] = make_float2(0.f, 0.f);
	// Everything block n needs has landed: called at the very END of block n-1 (and once in front of the loop), not at the top of
	// block n -- the compiler resolves loop-carried values with register copies at the top of the loop body, and a copy of a register
	// whose load is still in flight reads garbage (seen in the generated code of the first version; tools/check_async_isa.py scans the
	// ISA of every build for such reads)
	auto landed = [&](Async16 (&v)[G::LOADS]) {
		asyncWait<G::LOADS>();
#pragma unroll
		for (int i = 0; i < G::LOADS; ++i) asyncArrived(v[i]);
		if (FIRST) asyncArrived(xe); else asyncArrived(xv);
		asyncArrived(rotNext1);
		asyncArrived(rotNextL);
		if (FIRST) {
#pragma unroll
			for (int c = 0; c < CH; ++c) { asyncArrived(carNext1[c]); asyncArrived(carNextL[c]); }
		}
	};
	// (Reading the line that park() moves down a block EARLIER, so that its LDS round trip does not sit between the landed lines and
	// their parking, costs eight more live registers: the kernel is at its 128-register budget -- 48 bytes of scratch, recurrence
	// 6.75 -> 9.3 ms per step.  Measured, not kept.)
	auto park = [&](int n, int par, Async16 (&v)[G::LOADS]) {
		if (FIRST) {
#pragma unroll
			for (int c = 0; c < CH; ++c) { car1[c] = asyncValue(carNext1[c]); carL[c] = asyncValue(carNextL[c]); }
		}
#pragma unroll
		for (int i = 0; i < G::LOADS; ++i) {
			const int j = lineOf(n, par, i);
			if (j >= 0) { // upper line -> lower half, the new line -> upper half (this lane's piece of both)
				float4 *lower = reinterpret_cast<float4 *>(sbuf + llds[par][i]), *upper = reinterpret_cast<float4 *>(sbuf + llds[par][i] + 16);
				*lower = *upper;
				*upper = (j < lines) ? asyncValue(v[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
			}
		}
		const int x0 = BS*(n - 8*it) + 2*xpiece;
		if (xlane) {
			float4 piece;
			if (FIRST) { const float2 e = asyncValue(xe); piece = make_float4(e.x, 0.f, e.y, 0.f); }
			else piece = asyncValue(xv);
			*reinterpret_cast<float4 *>(xbuf + xc*16 + 2*xpiece) = (x0 >= 0 && x0 + 1 < M) ? piece : make_float4(0.f, 0.f, 0.f, 0.f);
		}
		rot1 = asyncValue(rotNext1);
		rotL = asyncValue(rotNextL);
	};
	const HopDesc hd = hopsLds[row < nh ? row : 0];
	const bool rotate = hd.flags & HOP_NEW_SPECTRUM;
	const float tf = hd.timeFactor;
	// The wave's first lines are due in block n0 = 8*it - 1 (m = -1 of its first row: line 0); for FIRST that is a block BEFORE the
	// tile's first one, which only parks.  Blocks before n0 (the wavefront has not reached this wave's rows): all-zero records.
	const int n0 = 8*it - 1;
	for (int skip = 0; skip < min(n0, totalBlocks); ++skip) {
		const int slot = skip%NB;
		while (skip - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2);
		asm volatile("" ::: "memory");
#pragma unroll
		for (int j = 0; j < NCH; ++j) recs[((slot*BS + st)*NCH + j)*64 + ((row + st) & 63)] = make_float4(0.f, 0.f, 0.f, 0.f);
		asm volatile("" ::: "memory");
		if (k == 0) ldsCount(&sync[slot]);
	}
	if (n0 >= totalBlocks) return; // (a tile so short that the wavefront never reaches this wave's rows)
	// n0 is odd (or -1): block n0 takes the register set of parity 0 ("E": blocks n with (n + 1) & 1 == 0), block n0 + 1 the other one
	issueLines(n0, 0, vE);
	issueSmall(n0);
	issueLines(n0 + 1, 1, vO);
	landed(vE);
	// Block n0 itself: every row of the wave is still in front of bin 0 (m = -1 for the first row) -- park, request, all-zero records
	// (FIRST: n0 = -1 lies before the tile, no records).  The blocks after it come in pairs, one per register set, with no
	// condition inside the loop: n0 + 1 and the number of blocks are both even.
	{
		park(n0, 0, vE);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		issueSmall(n0 + 1);
		issueLines(n0 + 2, 0, vE);
		if (!FIRST) {
			const int slot = n0%NB;
			while (n0 - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2);
			asm volatile("" ::: "memory");
#pragma unroll
			for (int j = 0; j < NCH; ++j) recs[((slot*BS + st)*NCH + j)*64 + ((row + st) & 63)] = make_float4(0.f, 0.f, 0.f, 0.f);
			asm volatile("" ::: "memory");
			if (k == 0) ldsCount(&sync[slot]);
		}
		landed(vO);
	}
	auto step = [&](int n, int par, Async16 (&v)[G::LOADS], Async16 (&vNextBlock)[G::LOADS]) {
		park(n, par, v);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		issueSmall(n + 1);
		issueLines(n + 2, par, v);
		const int slot = n%NB;
		while (n - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2); // slot still being read
		asm volatile("" ::: "memory");
		const int b0 = BS*(n - row), b = b0 + st;
		float f[NCH*4];
#pragma unroll
		for (int j = 0; j < NCH*4; ++j) f[j] = 0.0f;
		if (row < nh && b >= 0 && b < M) {
			// same arithmetic as computeRecord<CH, true, false, false>, operands from the line buffers.  This row's buffer holds the
			// lines (j-1, j) in its even blocks (b0 = 16j) and (j, j+1) in its odd ones, so bin b sits at 16 + st resp. 8 + st; the row
			// above runs 8 bins ahead (opposite parity): its buffer holds (j, j+1) either way, bin b at st resp. 8 + st.
			const bool odd = (n - row) & 1;
			const float2 *mine = sbuf + r*G::ROWLEN + (odd ? 8 : 16) + st;                               // bin b of channel 0's input
			const float2 *above = (r > 0) ? sbuf + (r - 1)*G::ROWLEN + (odd ? 8 : 0) + st : xbuf + st; // bin b of the hop above (r == 0: the staged 16 bins start at b0)
			const int abovePitch = (r > 0) ? G::RING : 16; // channel pitch of `above`
			auto IN = [&](int c, int off) { return mine[c*G::RING + off]; }; // bin b + off
			auto lerpIN = [&](int c, LerpIndex li) { // li.lo is an absolute bin
				const float2 low = mine[c*G::RING + (li.lo - b)], high = mine[c*G::RING + (li.lo - b) + 1];
				return clerp(low, high, li.fr);
			};
			float2 p[CH];
			float e[CH];
#pragma unroll
			for (int c = 0; c < CH; ++c) { p[c] = IN(c, 0); e[c] = cnorm(p[c]); }
			int mc = 0;
			float eMax = e[0];
#pragma unroll
			for (int c = 1; c < CH; ++c) if (e[c] > eMax) { mc = c; eMax = e[c]; }
			float2 Pm = p[0];
#pragma unroll
			for (int c = 1; c < CH; ++c) if (c == mc) Pm = p[c];
			const float fb = float(b);
			float2 A = cmulc(Pm, lerpIN(mc, lerpIndex(fb - tf)));
			float2 B = cmulc(Pm, lerpIN(mc, lerpIndex(fb - L*tf)));
			auto twist = [&](int off, float2 rotV, float stepMul) { // bx = b + off
				const int bc = min(b + off, M - 1);
				const float2 rotB = rotate ? rotV : make_float2(1.f, 0.f);
				const float2 Q = cmul(mine[(CH + mc)*G::RING + off], rotB);
				const float2 Px = IN(mc, off);
				const float2 TW = cmul(rotB, cmulc(Px, Q));
				const float eNow = cnorm(Px);
				// Prediction.energy of the previous hop: hop row-1's input, or the carried state
				const float2 up = above[mc*abovePitch + off];
				const float ePrev = (row > 0) ? cnorm(up) : up.x;
				const float den = fmaxf(ePrev, eNow) + 1e-15f;
				const float2 down = cmulc(Px, lerpIN(mc, lerpIndex(float(bc) - stepMul*tf)));
				const float2 rr = cmulc(TW, down);
				const float inv = __builtin_amdgcn_rcpf(den); // 1-ulp hardware reciprocal (an IEEE division costs ten instructions per record)
				return make_float2(rr.x*inv, rr.y*inv);
			};
			float2 Cc = twist(1, rot1, 1.0f), Dc = twist(L, rotL, float(L));
			const float2 zero = make_float2(0.f, 0.f);
			if (!(b > 0)) A = zero;
			if (!(b >= L)) B = zero;
			if (!(b < M - 1)) Cc = zero;
			if (!(b < M - L)) Dc = zero;
			if (FIRST) { // FOLD0: row 0's record carries the previous-hop part ready-made (lane select inside)
				float2 c1 = car1[0], cL = carL[0];
#pragma unroll
				for (int c = 1; c < CH; ++c) if (c == mc) { c1 = car1[c]; cL = carL[c]; }
				const float2 K = prevHopTerms(c1, Cc, cL, Dc);
				if (r == 0) { Cc = K; Dc = zero; }
			}
			f[0] = A.x; f[1] = A.y; f[2] = B.x; f[3] = B.y; f[4] = Cc.x; f[5] = Cc.y; f[6] = Dc.x; f[7] = Dc.y;
			f[8] = __int_as_float(mc);
			recordChannelFields<CH>(f, p, e, mc);
		}
#pragma unroll
		for (int j = 0; j < NCH; ++j) recs[((slot*BS + st)*NCH + j)*64 + ((row + st) & 63)] = make_float4(f[4*j], f[4*j + 1], f[4*j + 2], f[4*j + 3]);
		asm volatile("" ::: "memory");
		if (k == 0) ldsCount(&sync[slot]);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier(); // every lane has read its operands before the next block's lines are parked
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		landed(vNextBlock);
	};
	for (int n = n0 + 1; n < totalBlocks; n += 2) {
		step(n, 1, vO, vE);
		step(n + 1, 0, vE, vO);
	}
	asyncWait<0>(); // the requests that ran past the tile's last block: nothing of this wave stays in flight behind it
}

constexpr int kVocBlockSteps = 8, kVocBlocks = 3, kVocBlocksStaged = 2, kVocWaves = 16, kVocStagedProducers = 8, kVocOutBlocks = 4;
constexpr int kVocOutBlocksAligned = 3; // lag 8: a row's 16-bin line lies in exactly two result blocks
// results ring: [block][step][channel][kVocOutPitch] -- 66, not 64: the writer reads a row's values of steps 2 apart in adjacent
// lane groups, and 2*CH*64 float2 is a multiple of the 32 banks (an 8-way conflict on every writer read with the first layout)
constexpr int kVocOutPitch = 66;

__device__ __forceinline__ float2 selectPair(bool pick, float2 a, float2 b) { return make_float2(pick ? a.x : b.x, pick ? a.y : b.y); }
__device__ __forceinline__ float2 fromLaneBelow(float2 v, float2 lane0) { // lane k receives lane k-1's v; lane 0 keeps its `lane0`
	// DPP wave_shr:1 without bound_ctrl: a lane with no source lane keeps the old value of the destination register
	return make_float2(__int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0.x), __float_as_int(v.x), 0x138, 0xf, 0xf, false)),
	                   __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0.y), __float_as_int(v.y), 0x138, 0xf, 0xf, false)));
}

// ACROSS (single-hop tiles, the real-time calling pattern: every stream fires at most one hop per call): the 64 lanes of the
// recurrence wave are 64 STREAMS (up to acrossRows of them per workgroup) instead of 64 hops of one stream.  Every row is the
// first hop of its tile, so every record carries its previous-hop terms ready-made (FOLD0) and no lane needs another lane's
// output: no skew (lag 0), no DPP, M steps per launch.  kVocoderOne runs one chain per WAVE (64 lanes computing the same
// values); at 4096 streams that is four chain waves per SIMD and 1.57 ms per hop quantum.  Same records, same arithmetic:
// bit-identical to the other recurrence kernels.
// ALIGNED (with STAGED): the line-aligned producers and a wavefront lag of 8 bins (vocoderProduceAligned).
template <int CH, bool PLAIN, int L, bool STAGED, bool ROTL = false, bool ACROSS = false, bool ALIGNED = false>
__global__ __launch_bounds__(64*kVocWaves) __attribute__((amdgpu_waves_per_eu(4, 4))) void kVocoder(DevBatch d, int sBase, int hopBase, int acrossRows, int acrossStreams) {
	static_assert(!STAGED || (PLAIN && L <= 5), "staged producers: identity map, bounded windows");
	static_assert(!ALIGNED || (STAGED && L <= 4), "line-aligned producers: windows within the two lines around a row's bins");
	static_assert(!ACROSS || !STAGED, "rows that are streams gather their operands");
	static_assert(!ROTL || (!PLAIN && !STAGED), "the LDS copy of the rotation table serves the gathering producers of mapped tiles");
	constexpr int NF = 9 + 3*CH, NCH = (NF + 3)/4, BS = kVocBlockSteps, NB = STAGED ? kVocBlocksStaged : kVocBlocks;
	constexpr int NP = STAGED ? kVocStagedProducers : kVocWaves - 2;
	constexpr int lag = ACROSS ? 0 : (ALIGNED ? 8 : L + 1);
	constexpr int OB = ALIGNED ? kVocOutBlocksAligned : kVocOutBlocks; // result ring blocks
	static_assert(BS == 8 && L >= 1 && L <= 7 && lag <= 8, "history registers are indexed by step & 7");
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float4 *recs = reinterpret_cast<float4 *>(smemRaw);                 // [(slot*BS + st)*NCH + j][64 lanes]
	volatile int *sync = reinterpret_cast<volatile int *>(recs + NB*BS*NCH*64); // [0..NB) units produced, [NB] blocks consumed
	int *rowClass = const_cast<int *>(sync) + 16;                                  // [2][64]: the writer's two classes of rows
	HopDesc *hopsLds = reinterpret_cast<HopDesc *>(rowClass + 128);                // the tile's 64 hop descriptors
	float2 *outRing = reinterpret_cast<float2 *>(hopsLds + 64);                    // [OB][BS][CH][kVocOutPitch]: results on their way to HBM
	// sync words: [0..NB) units produced per slot, [NB] blocks consumed, [NB+1] result blocks ready, [NB+2] result blocks written

	// rows of the workgroup: hops 0 .. nh-1 of stream s, or (ACROSS) hop 0 of streams s .. s+nh-1
	const int s = ACROSS ? blockIdx.x*acrossRows : blockIdx.x, sg = sBase + s;
	const int nh = ACROSS ? min(acrossRows, acrossStreams - s) : d.nHops[s];
	if (nh <= 0) return;
	const int wave = threadIdx.x >> 6, k = threadIdx.x & 63;
	const int M = d.M;
	const int steps = M + lag*(nh - 1);
	const int chunks = (steps + 63) >> 6;
	const int totalBlocks = chunks*(64/BS);
	const CarriedOutput stOut = carriedOutput(d, sg);
	auto rowStream = [&](int row) { return ACROSS ? s + row : s; }; // sub-batch-local stream of a row
	auto rowHop = [&](int row) { return ACROSS ? 0 : row; };         // tile-local hop of a row

	// prologue (all waves): clear the hand-off words, cache the hop table
	if (threadIdx.x <= NB + 2) sync[threadIdx.x] = 0;
	if (threadIdx.x < 64) {
		if constexpr (ACROSS) {
			HopDesc hd{};
			const int row = threadIdx.x;
			if (row < nh && d.nHops[s + row] > 0) hd = d.hops[(size_t)(sg + row)*d.hopStride + hopBase];
			hopsLds[row] = hd; // streams without a hop in this call: flags == 0, all-zero records, nothing written
		} else {
			hopsLds[threadIdx.x] = d.hops[(size_t)sg*d.hopStride + hopBase + threadIdx.x];
		}
	}
	float2 *rotLds = outRing + (size_t)OB*BS*CH*kVocOutPitch; // [M] hop rotation table (ROTL; the staged kernels keep their windows here)
	if constexpr (ROTL) {
		for (int i = threadIdx.x; i < M; i += blockDim.x) rotLds[i] = d.rot[i];
	}
	__syncthreads();

	if (wave > 0) {
		// ---------------- producers ----------------
		if (wave == 4) {
			// ---------------- writer ----------------
			// Drains the consumer's results to HBM.  Per lane and step the consumer would issue one 8-byte store per channel
			// into 64 different cache lines (128 partial-line transactions per step, competing with the producers' loads);
			// here 8 lanes cover 16 bins of one row with 16-byte stores: one whole, ALIGNED 128-byte line (rows start on line
			// boundaries).  The first version stored whatever 8 bins a row had produced in the block, at 8-byte alignment, and
			// the partial lines went to HBM twice (rocprofv3 WRITE_SIZE 1.42 GB per launch for 0.79 GB of results); aligned
			// 64-byte halves still gave 1.04 GB.  Row r has produced line G = (n - ceil(lag*r/8) - 1)/2 completely at the end of
			// block n when n - ceil(lag*r/8) is odd, so the rows fall into two classes that store on alternate blocks; the
			// line's bins lie in ring blocks n-2..n (ring of four), and two extra passes after the last block flush the rows'
			// final lines.  Bins >= M of a line land in the rows' padding as zeros.
			int count[2] = {0, 0}; // rowClass[q][.]: the rows with ceil(lag*row/8) = q (mod 2)
			for (int r = 0; r < 64; ++r) { // every lane walks the same list; lane 0 records it
				const int q = ((lag*r + 7) >> 3) & 1;
				if (k == 0) rowClass[q*64 + count[q]] = r;
				++count[q];
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			const int g8 = k & 7, part = k >> 3; // eight consecutive lanes read eight consecutive rows of the ring (conflict-free); a store instruction still covers whole lines
			for (int n = 0; n <= totalBlocks + 1; ++n) {
				if (n < totalBlocks) {
					while (ldsPeek(&sync[NB + 1]) <= n) __builtin_amdgcn_s_sleep(2);
				}
				asm volatile("" ::: "memory");
				const int q = (n + 1) & 1; // rows with ceil(lag*row/8) = n - 1 (mod 2) complete a line with this block
				for (int pass = 0; 8*pass < count[q]; ++pass) {
					const int idx = 8*pass + g8;
					const int row = rowClass[q*64 + (idx < count[q] ? idx : 0)];
					const int G = (n - ((lag*row + 7) >> 3) - 1) >> 1;
					const int b = 16*G + 2*part;
					const bool ok = idx < count[q] && row < nh && G >= 0 && 16*G < M && (!ACROSS || (hopsLds[row].flags & HOP_ACTIVE));
					const int t0 = b + lag*row, t1 = t0 + 1; // the steps at which the two bins were produced
					const int r0 = t0 >= 0 ? (t0 >> 3)%OB : 0, r1 = t1 >= 0 ? (t1 >> 3)%OB : 0;
#pragma unroll
					for (int c = 0; c < CH; ++c) {
						float2 v0 = outRing[((r0*BS + (t0 & 7))*CH + c)*kVocOutPitch + row], v1 = outRing[((r1*BS + (t1 & 7))*CH + c)*kVocOutPitch + row];
						if (b >= M) v0 = make_float2(0.f, 0.f);
						if (b + 1 >= M) v1 = make_float2(0.f, 0.f);
						if (ok) {
							float2 *dst = d.OUT + rowOf(d, rowStream(row), rowHop(row), c) + b;
							dst[0] = v0;
							dst[1] = v1;
						}
					}
				}
				asm volatile("" ::: "memory");
				if (k == 0) ldsPost(&sync[NB + 2], n + 1);
			}
			return;
		}
		// 0..NP-1 over the producer waves.  Waves w and w+4 share a SIMD (tools/probes/wave_simd_map.hip).  The staged kernel
		// runs 8 producers on three SIMDs (waves 1,5,9 / 2,6,10 / 3,7) and leaves the recurrence wave's SIMD to it and the
		// writer: with two producers beside it (the first placement) the recurrence wave, which is the critical path once the
		// producers are light enough, lost issue slots to them -- 8.15 -> 7.45 ms per step.  (Earlier in the round, with
		// heavier producers, the same move changed nothing.)
		int pIndex = wave - 1 - (wave > 4);
		if (STAGED) pIndex = (wave & 3) ? ((wave < 8) ? pIndex : ((wave == 9) ? 6 : ((wave == 10) ? 7 : NP))) : NP;
		// aligned form: the wave that owns rows 0..7 (carried taps, FOLD0 -- the heaviest, and every block waits for the slowest producer)
		// on the SIMD that holds only two producers (waves 3, 7): recurrence 0.97 -> 0.95 ms in place, step -0.12 ms
		if (STAGED && ALIGNED && pIndex < NP) pIndex = (pIndex == 0) ? 2 : ((pIndex == 2) ? 0 : pIndex);
		if (pIndex >= NP) return;
		if constexpr (ALIGNED) {
			using G = AlignGeom<CH, L>;
			float2 *sbuf = outRing + (size_t)OB*BS*CH*kVocOutPitch + (size_t)pIndex*G::PER_PRODUCER;
			if (pIndex == 0) vocoderProduceAligned<CH, L, NB, true>(d, s, sg, nh, pIndex, k, totalBlocks, recs, sync, hopsLds, sbuf, stOut);
			else vocoderProduceAligned<CH, L, NB, false>(d, s, sg, nh, pIndex, k, totalBlocks, recs, sync, hopsLds, sbuf, stOut);
			return;
		} else if constexpr (STAGED) {
			using G = StageGeom<CH, L>;
			float2 *sbuf = outRing + (size_t)OB*BS*CH*kVocOutPitch + (size_t)pIndex*G::ROWS*G::ROWLEN;
			vocoderProduceStaged<CH, L, NB, NP>(d, s, sg, nh, pIndex, k, totalBlocks, recs, sync, hopsLds, sbuf, stOut);
			return;
		}
		const int st = k & 7, r = k >> 3; // 8 adjacent lanes = 8 consecutive bins of one row: 64-byte contiguous global loads
		for (int u = pIndex; u < totalBlocks*8; u += NP) {
			const int n = u >> 3, it = u & 7;
			const int slot = n%NB;
			const int row = 8*it + r;
			const int t = BS*n + st;
			const int b = t - lag*row;
			float f[NCH*4];
#pragma unroll
			for (int j = 0; j < NCH*4; ++j) f[j] = 0.0f;
			if (row < nh && b >= 0 && b < M && !SMST_SKIP_PRODUCER_MATH(d)) {
				if constexpr (ACROSS) {
					if (hopsLds[row].flags & HOP_ACTIVE) computeRecord<CH, PLAIN, false, false, NCH*4, ROTL, true>(d, hopsLds[row], hopsLds[row], s + row, sg + row, 0, b, f, rotLds);
				} else {
					computeRecord<CH, PLAIN, false, false, NCH*4, ROTL, true>(d, hopsLds[row], hopsLds[row > 0 ? row - 1 : 0], s, sg, row, b, f, rotLds);
				}
			}
			while (n - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2); // slot still being read (waited for AFTER the pass is computed)
			asm volatile("" ::: "memory");
#pragma unroll
			// lane rotation by st: the 8 lanes of a row (same row, 8 steps = 8 LDS rows a multiple of 4 KB apart) land in 8 different
			// 16-byte bank groups (a rotation by 2*st, the first version, used only four of them)
			for (int j = 0; j < NCH; ++j) recs[((slot*BS + st)*NCH + j)*64 + ((row + st) & 63)] = make_float4(f[4*j], f[4*j + 1], f[4*j + 2], f[4*j + 3]);
			asm volatile("" ::: "memory");
			if (k == 0) ldsCount(&sync[slot]); // LDS ops of a wave are in order: data first, then the count
		}
		return;
	}

	// ---------------- consumer (wave 0) ----------------
	__builtin_amdgcn_s_setprio(3);
	float2 h[8][CH]; // this lane's outputs of the last 8 steps
	// Previous-hop taps: lane k receives lane k-1's history registers by DPP.  Lane 0's previous hop is the carried state, which
	// its records have folded in (FOLD0): its taps are the constants (1, 0) and (0, 0).  A wave_shr:1 never writes lane 0, so the
	// constants are set ONCE, here, and each tap register is the `old` operand of the next DPP move into itself -- no LDS read,
	// no staging window and no register copy for lane 0 on the serial path.
	float2 tap1[CH], tapL[CH];
#pragma unroll
	for (int c = 0; c < CH; ++c) {
#pragma unroll
		for (int i = 0; i < 8; ++i) h[i][c] = make_float2(0.f, 0.f);
		tap1[c] = make_float2((ACROSS || k == 0) ? 1.f : 0.f, 0.f); // ACROSS: every lane is a first hop
		tapL[c] = make_float2(0.f, 0.f);
	}
	// the two hand-off words the NEXT block waits for are read during the current block's last step (an LDS round trip each,
	// 200 clock cycles, sat on the serial path at every block boundary -- cycle trace); the poll loops remain for the rare miss
	int seenProduced = ldsPeek(&sync[0]), seenWritten = 0;
	for (int n = 0; n < totalBlocks; ++n) {
		const int slot = n%NB;
		const int need = 8*(n/NB + 1);
		while (seenProduced < need) { __builtin_amdgcn_s_sleep(1); seenProduced = ldsPeek(&sync[slot]); }
		asm volatile("" ::: "memory");
		const float4 *blockRecs = recs + (size_t)slot*BS*NCH*64;
		while (n - seenWritten >= 2) { __builtin_amdgcn_s_sleep(1); seenWritten = ldsPeek(&sync[NB + 2]); } // the writer still owns this result slot
		asm volatile("" ::: "memory");
		float2 *blockOut = outRing + (size_t)(n%OB)*BS*CH*kVocOutPitch + k;
		float4 q[2][NCH]; // two register sets alternate, so the next step's record loads never overwrite live values
#pragma unroll
		for (int j = 0; j < NCH; ++j) q[0][j] = blockRecs[j*64 + k];
#pragma unroll
		for (int i = 0; i < BS; ++i) {
			if (SMST_CONSUMER_ONLY_ACKNOWLEDGES(d)) break; // experiment builds only
			if (i + 1 < BS) {
#pragma unroll
				for (int j = 0; j < NCH; ++j) q[(i + 1) & 1][j] = blockRecs[((i + 1)*NCH + j)*64 + ((k + (i + 1)) & 63)];
			} else { // last step: look at the next block's hand-off words now, their latency hides under this step
				seenProduced = ldsPeek(&sync[(n + 1)%NB]);
				seenWritten = ldsPeek(&sync[NB + 2]);
			}
			float f[NCH*4];
#pragma unroll
			for (int j = 0; j < NCH; ++j) { f[4*j] = q[i & 1][j].x; f[4*j + 1] = q[i & 1][j].y; f[4*j + 2] = q[i & 1][j].z; f[4*j + 3] = q[i & 1][j].w; }
			const int mc = __float_as_int(f[8]); // 0 .. CH-1: every record of the ring was written by a producer (all-zero outside the tile)
			// taps: own history (bins b-1, b-L), and lane k-1's history: it runs L+1 bins ahead, so ITS b-L and b-1 taps are this
			// lane's previous-hop taps at b+1 and b+L
#pragma unroll
			for (int c = 0; c < CH; ++c) {
				if constexpr (!ACROSS) {
					// lane k-1 finished its bin b+x (x = 1, L) lag - x steps ago
					tap1[c] = fromLaneBelow(h[(i + 17 - lag) & 7][c], tap1[c]);
					tapL[c] = fromLaneBelow(h[(i + 16 + L - lag) & 7][c], tapL[c]);
				}
			}
			// the maximum channel's taps: explicit per-component selects (v_cndmask) -- written as an `if` the compiler makes a branch of
			// it, with a register copy in front of every tap that must survive (14 moves against 8 selects)
			float2 o1 = h[(i + 7) & 7][0], oL = h[(i + 8 - L) & 7][0], p1 = tap1[0], pL = tapL[0];
#pragma unroll
			for (int c = 1; c < CH; ++c) {
				const bool pick = c == mc;
				o1 = selectPair(pick, h[(i + 7) & 7][c], o1);
				oL = selectPair(pick, h[(i + 8 - L) & 7][c], oL);
				p1 = selectPair(pick, tap1[c], p1);
				pL = selectPair(pick, tapL[c], pL);
			}
			const float2 pm = make_float2(f[9], f[10]); // mono: the channel's input; stereo: the maximum channel's fallback output (recordChannelFields)
			const float sm = f[11];
			float2 phi = prevHopTerms(p1, make_float2(f[4], f[5]), pL, make_float2(f[6], f[7])); // previous hop's part first (what FOLD0 records pre-compute)
			phi = cfma(oL, make_float2(f[2], f[3]), phi);
			phi = cfma(o1, make_float2(f[0], f[1]), phi); // the newest operand last: two dependent instructions behind it
			const float2 om = (CH == 2) ? makeOutputFb(phi, pm, sm) : makeOutput(phi, pm, sm); // :788
			if (CH == 2) { // one locked channel (:791-800), its makeOutput folded into the record
				const float2 olock = lockedOutput(om, f);
				// cells outside the tile (inactive hop, bin outside [0, M)) have all-zero records, which give exactly zero here
				const float2 oc0 = mc ? olock : om, oc1 = mc ? om : olock;
				h[i][0] = oc0;
				h[i][CH - 1] = oc1;
				blockOut[(i*CH)*kVocOutPitch] = oc0;
				blockOut[(i*CH + CH - 1)*kVocOutPitch] = oc1;
			} else {
#pragma unroll
				for (int c = 0; c < CH; ++c) {
					h[i][c] = om;
					blockOut[(i*CH + c)*kVocOutPitch] = om;
				}
			}
		}
		asm volatile("" ::: "memory");
		if (k == 0) { ldsPost(&sync[NB], n + 1); ldsPost(&sync[NB + 1], n + 1); } // record slot may be refilled; results may be written out
	}
}

// ------------------------------------------------------------------------------------------------------
// K3 fused, 3-8 channels (signalsmith-stretch.h:722-803 for any channel count).  Same organisation as kVocoder -- producer
// waves compute the records into an LDS ring, wave 0 runs the skewed wavefront, wave 4 drains the results with
// row-coalesced stores -- with three differences that the channel count forces:
//   * a record is 9 + 3*CH floats (36 for 8 channels), so a block is 4 steps instead of 8 (2 x 4 x 9 KiB of LDS);
//   * the consumer's history cannot live in registers (8 steps x CH complex values): each lane keeps its last output per
//     channel in registers (the b-1 tap) and everything else in an LDS ring [CH][16 bins][64 lanes] indexed by the bin,
//     which the next lane (the b+1 / b+L taps of the previous hop) and the writer read as well -- so there is no
//     separate result buffer;
//   * the producers gather (computeRecord), as the un-fused kPredictB does: same arithmetic, bit-identical results.
// It replaces kPredictB + kChain, whose records went through HBM (14 MB per stream and tile) and whose recurrence
// issued CH scattered 8-byte stores per lane and step -- every record prefetch then waited behind those stores
// (vmcnt counts both on gfx9): 4.4 us per step for 8 channels.
// ------------------------------------------------------------------------------------------------------
constexpr int kVocNBlockSteps = 4, kVocNBlocks = 2, kVocNRing = 16;

template <int CH, bool PLAIN, int L>
__global__ __launch_bounds__(64*kVocWaves) __attribute__((amdgpu_waves_per_eu(4, 4))) void kVocoderN(DevBatch d, int sBase, int hopBase) {
	constexpr int NF = 9 + 3*CH, NCH = (NF + 3)/4, BS = kVocNBlockSteps, NB = kVocNBlocks, R = kVocNRing, Rm = R - 1;
	constexpr int NP = kVocWaves - 2;
	constexpr int lag = L + 1;
	static_assert(CH >= 3 && CH <= kMaxChannels && L >= 1 && L + BS < R, "ring depth: a slot is rewritten R bins later, the oldest tap is L bins back");
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float4 *recs = reinterpret_cast<float4 *>(smemRaw);                  // [(slot*BS + st)*NCH + j][64 lanes]
	float2 *ring = reinterpret_cast<float2 *>(recs + NB*BS*NCH*64);      // [CH][R bins][64 lanes]: Band.output of the last R bins of every hop
	float2 *stage = ring + CH*R*64;                                      // [CH][128]: carried Band.output, 128-bin window
	volatile int *sync = reinterpret_cast<volatile int *>(stage + CH*128); // [0..NB) units produced, [NB] blocks consumed, [NB+1] result blocks ready, [NB+2] written
	HopDesc *hopsLds = reinterpret_cast<HopDesc *>(const_cast<int *>(sync) + 16);

	const int s = blockIdx.x, sg = sBase + s;
	const int nh = d.nHops[s];
	if (nh == 0) return;
	const int wave = threadIdx.x >> 6, k = threadIdx.x & 63;
	const int M = d.M;
	const int steps = M + lag*(nh - 1);
	const int chunks = (steps + 63) >> 6;
	const int totalBlocks = chunks*(64/BS);
	const CarriedOutput stOut = carriedOutput(d, sg);

	for (int i = threadIdx.x; i < CH*128; i += blockDim.x) {
		const int c = i >> 7, bb = i & 127;
		stage[i] = (bb < M) ? stOut[(size_t)c*M + bb] : make_float2(0.f, 0.f);
	}
	for (int i = threadIdx.x; i < CH*R*64; i += blockDim.x) ring[i] = make_float2(0.f, 0.f);
	if (threadIdx.x <= NB + 2) sync[threadIdx.x] = 0;
	if (threadIdx.x < 64) hopsLds[threadIdx.x] = d.hops[(size_t)sg*d.hopStride + hopBase + threadIdx.x];
	__syncthreads();

	if (wave > 0) {
		if (wave == 4) {
			// ---------------- writer: an ALIGNED group of 4 bins of a row = one 32-byte sector per channel, two lanes per row.
			// With block n row r has completed group n - ceil(lag*r/4) (the bins a row produced in the block itself straddle two
			// sectors, and partial sectors went to HBM twice -- see kVocoder's writer); one extra pass flushes the last groups.
			const int g = k >> 1, part = k & 1;
			for (int n = 0; n <= totalBlocks; ++n) {
				if (n < totalBlocks) {
					while (ldsPeek(&sync[NB + 1]) <= n) __builtin_amdgcn_s_sleep(2);
				}
				asm volatile("" ::: "memory");
#pragma unroll
				for (int pass = 0; pass < 2; ++pass) {
					const int row = 32*pass + g;
					const int grp = n - ((lag*row + 3) >> 2);
					const int b0 = 4*grp + 2*part;
					const bool ok = row < nh && grp >= 0 && 4*grp < M;
#pragma unroll
					for (int c = 0; c < CH; ++c) {
						float2 v0 = ring[(c*R + (b0 & Rm))*64 + row], v1 = ring[(c*R + ((b0 + 1) & Rm))*64 + row];
						if (b0 >= M) v0 = make_float2(0.f, 0.f); // not produced in this tile: the slot holds an older bin
						if (b0 + 1 >= M) v1 = make_float2(0.f, 0.f);
						if (ok) { // bins M .. M+2 of the last group land in the rows' padding
							float2 *dst = d.OUT + rowOf(d, s, row, c) + b0;
							dst[0] = v0;
							dst[1] = v1;
						}
					}
				}
				asm volatile("" ::: "memory");
				if (k == 0) ldsPost(&sync[NB + 2], n + 1);
			}
			return;
		}
		// ---------------- producers: 16 rows x 4 steps per wave-pass
		const int pIndex = wave - 1 - (wave > 4);
		const int st = k & (BS - 1), r = k/BS;
		constexpr int ROWS = 64/BS, UNITS = 64/ROWS; // passes per block
		for (int u = pIndex; u < totalBlocks*UNITS; u += NP) {
			const int n = u/UNITS, it = u - n*UNITS;
			const int slot = n%NB;
			const int row = ROWS*it + r;
			const int b = BS*n + st - lag*row;
			float f[NCH*4];
#pragma unroll
			for (int j = 0; j < NCH*4; ++j) f[j] = 0.0f;
			if (row < nh && b >= 0 && b < M && !SMST_SKIP_PRODUCER_MATH(d)) computeRecord<CH, PLAIN, false, false>(d, hopsLds[row], hopsLds[row > 0 ? row - 1 : 0], s, sg, row, b, f);
			// The records depend on feed-forward data only, so a pass is COMPUTED as soon as its wave is free and waits for its
			// slot just before it is stored.  With the wait in front (first version) one block was in production at a time: a
			// pass is two dependent rounds of gathers, 7 + 15 thousand cycles on a full tile (cycle trace), the 2-block ring let
			// 4 of the 14 producers work, and the recurrence wave waited 57 % of every block for records.
			while (n - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2); // slot still being read
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
	float2 pf[CH], own1[CH];
#pragma unroll
	for (int c = 0; c < CH; ++c) { pf[c] = make_float2(0.f, 0.f); own1[c] = make_float2(0.f, 0.f); }
	constexpr int UNITS = BS;
	for (int ch = 0; ch < chunks; ++ch) {
		const int tb = ch << 6;
		if (ch > 0) {
#pragma unroll
			for (int c = 0; c < CH; ++c) stage[c*128 + ((tb + 64 + k) & 127)] = pf[c];
		}
		{
			const int bb = tb + 128 + k;
			const int bc = (bb < M) ? bb : M - 1;
#pragma unroll
			for (int c = 0; c < CH; ++c) {
				float2 v = stOut[(size_t)c*M + bc];
				pf[c] = (bb < M) ? v : make_float2(0.f, 0.f);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		for (int blk = 0; blk < 64/BS; ++blk) {
			const int n = ch*(64/BS) + blk;
			const int slot = n%NB;
			const int need = UNITS*(n/NB + 1);
			while (ldsPeek(&sync[slot]) < need) __builtin_amdgcn_s_sleep(1);
			// the ring slots this block overwrites last held the bins of 16 steps ago; the writer's pass m reads bins down to
			// 4m - lag*row - 3, so it must have finished pass n - 3 (one block less slack than with unaligned groups)
			while (n - ldsPeek(&sync[NB + 2]) >= R/BS - 1) __builtin_amdgcn_s_sleep(1);
			asm volatile("" ::: "memory");
			const float4 *blockRecs = recs + (size_t)slot*BS*NCH*64;
#pragma unroll
			for (int i = 0; i < BS; ++i) {
				if (SMST_CONSUMER_ONLY_ACKNOWLEDGES(d)) break; // experiment builds only
				const int t = tb + blk*BS + i;
				float f[NCH*4];
#pragma unroll
				for (int j = 0; j < NCH; ++j) {
					const float4 q = blockRecs[(i*NCH + j)*64 + ((k + i) & 63)];
					f[4*j] = q.x; f[4*j + 1] = q.y; f[4*j + 2] = q.z; f[4*j + 3] = q.w;
				}
				const int b = t - kLag;
				int mc = __float_as_int(f[8]);
				mc = (mc < 0) ? 0 : ((mc > CH - 1) ? CH - 1 : mc);
				float2 o1 = own1[0], pm = make_float2(f[9], f[10]);
				float sm = f[11];
#pragma unroll
				for (int c = 1; c < CH; ++c) {
					if (c == mc) { o1 = own1[c]; pm = make_float2(f[9 + 3*c], f[10 + 3*c]); sm = f[11 + 3*c]; }
				}
				const int ringRow = mc*R;
				const float2 oL = ring[(ringRow + ((b - L) & Rm))*64 + k];
				const float2 p1 = (k == 0) ? stage[mc*128 + ((b + 1) & 127)] : ring[(ringRow + ((b + 1) & Rm))*64 + k - 1];
				const float2 pL = (k == 0) ? stage[mc*128 + ((b + L) & 127)] : ring[(ringRow + ((b + L) & Rm))*64 + k - 1];
				float2 phi = prevHopTerms(p1, make_float2(f[4], f[5]), pL, make_float2(f[6], f[7])); // previous hop's part first (what FOLD0 records pre-compute)
				phi = cfma(oL, make_float2(f[2], f[3]), phi);
				phi = cfma(o1, make_float2(f[0], f[1]), phi); // the newest operand last: two dependent instructions behind it
				const float2 om = makeOutput(phi, pm, sm); // :788
#pragma unroll
				for (int c = 0; c < CH; ++c) {
					const float2 pc = make_float2(f[9 + 3*c], f[10 + 3*c]);
					float2 oc = makeOutput(cmul(om, cmulc(pc, pm)), pc, f[11 + 3*c]); // channel lock, :791-800
					if (c == mc) oc = om;
					// cells outside the tile (inactive hop, bin outside [0, M)) have all-zero records, which give exactly zero here
					own1[c] = oc;
					ring[(c*R + (b & Rm))*64 + k] = oc;
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier(); // lane k+1 reads what lane k wrote lag-1 .. lag+L-1 steps ago
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			}
			asm volatile("" ::: "memory");
			if (k == 0) { ldsPost(&sync[NB], n + 1); ldsPost(&sync[NB + 1], n + 1); }
		}
	}
}

// ------------------------------------------------------------------------------------------------------
// K3 for single-hop tiles: the real-time calling pattern (one process() per 128-frame render quantum, web/web-wrapper.js:
// 255-315) fires at most ONE hop per stream and call.  The skewed wavefront then has one active lane per wave, and a
// 16-wave workgroup holding 127 KB of LDS per stream serialises the streams in rounds of 256 (4.4 ms for 1024 streams).
// Here a stream costs two waves and 12-30 KB: wave 1 computes the records of 64 consecutive bins per pass (lane = bin:
// every load is one contiguous row segment) and stores the finished results; wave 0 runs the bin recurrence of the single
// hop with its history in registers (every lane computes the same chain; lane 0 publishes).  All streams of a call are
// resident at once (1024 stereo streams: 12 waves per CU), so the latency of the hop is the length of ONE chain.
// Same records (computeRecord), same order of operations as kVocoder / kVocoderN: bit-identical results.
// ------------------------------------------------------------------------------------------------------
constexpr int kVocOneBlock = 64;

template <int CH, bool PLAIN, int L>
__global__ __launch_bounds__(128) void kVocoderOne(DevBatch d, int sBase, int hopBase) {
	constexpr int NF = 9 + 3*CH, NCH = (NF + 3)/4, BS = kVocOneBlock, NB = 2;
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float4 *recs = reinterpret_cast<float4 *>(smemRaw);                  // [slot][step][NCH]
	float2 *outRing = reinterpret_cast<float2 *>(recs + NB*BS*NCH);      // [2 blocks][CH][BS]: results on their way to HBM
	float2 *stage = outRing + 2*CH*BS;                                   // [CH][128]: carried Band.output, 128-bin window
	volatile int *sync = reinterpret_cast<volatile int *>(stage + CH*128); // [0] blocks produced, [1] blocks consumed (= result blocks ready)
	HopDesc *hopLds = reinterpret_cast<HopDesc *>(const_cast<int *>(sync) + 4);

	const int s = blockIdx.x, sg = sBase + s;
	if (d.nHops[s] == 0) return;
	const int wave = threadIdx.x >> 6, k = threadIdx.x & 63;
	const int M = d.M;
	const int totalBlocks = (M + BS - 1)/BS;
	const CarriedOutput stOut = carriedOutput(d, sg);
	for (int i = threadIdx.x; i < CH*128; i += blockDim.x) {
		const int c = i >> 7, bb = i & 127;
		stage[i] = (bb < M) ? stOut[(size_t)c*M + bb] : make_float2(0.f, 0.f);
	}
	if (threadIdx.x < 2) sync[threadIdx.x] = 0;
	if (threadIdx.x == 0) hopLds[0] = d.hops[(size_t)sg*d.hopStride + hopBase];
	__syncthreads();

	if (wave == 1) {
		// ---------------- producer + writer ----------------
		auto writeBlock = [&](int n) {
			while (ldsPeek(&sync[1]) <= n) __builtin_amdgcn_s_sleep(2);
			asm volatile("" ::: "memory");
			const int b = BS*n + k;
#pragma unroll
			for (int c = 0; c < CH; ++c) {
				const float2 v = outRing[((n & 1)*CH + c)*BS + k];
				if (b < M) d.OUT[rowOf(d, s, 0, c) + b] = v;
			}
		};
		for (int n = 0; n < totalBlocks; ++n) {
			// slot n % 2 last held block n - 2, which the consumer has finished once block n - 1's results could be written
			const int b = BS*n + k;
			float f[NCH*4];
#pragma unroll
			for (int j = 0; j < NCH*4; ++j) f[j] = 0.0f;
			if (b < M) computeRecord<CH, PLAIN, false, false>(d, hopLds[0], hopLds[0], s, sg, 0, b, f);
			float4 *dst = recs + ((size_t)(n % NB)*BS + k)*NCH;
#pragma unroll
			for (int j = 0; j < NCH; ++j) dst[j] = make_float4(f[4*j], f[4*j + 1], f[4*j + 2], f[4*j + 3]);
			asm volatile("" ::: "memory");
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			if (k == 0) ldsPost(&sync[0], n + 1);
			if (n > 0) writeBlock(n - 1);
		}
		writeBlock(totalBlocks - 1);
		return;
	}

	// ---------------- consumer (wave 0): every lane runs the chain of hop 0; lane 0 publishes ----------------
	__builtin_amdgcn_s_setprio(3);
	float2 pf[CH];
	float2 h[8][CH];
#pragma unroll
	for (int c = 0; c < CH; ++c) {
		pf[c] = make_float2(0.f, 0.f);
#pragma unroll
		for (int i = 0; i < 8; ++i) h[i][c] = make_float2(0.f, 0.f);
	}
	for (int n = 0; n < totalBlocks; ++n) {
		const int tb = n*BS;
		if (n > 0) {
#pragma unroll
			for (int c = 0; c < CH; ++c) stage[c*128 + ((tb + 64 + k) & 127)] = pf[c];
		}
		{
			const int bb = tb + 128 + k;
			const int bc = (bb < M) ? bb : M - 1;
#pragma unroll
			for (int c = 0; c < CH; ++c) {
				float2 v = stOut[(size_t)c*M + bc];
				pf[c] = (bb < M) ? v : make_float2(0.f, 0.f);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		while (ldsPeek(&sync[0]) <= n) __builtin_amdgcn_s_sleep(1);
		asm volatile("" ::: "memory");
		const float4 *blockRecs = recs + (size_t)(n % NB)*BS*NCH;
		float2 *blockOut = outRing + (size_t)(n & 1)*CH*BS;
		for (int i8 = 0; i8 < BS/8; ++i8) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const int step = i8*8 + i, b = tb + step;
				float f[NCH*4];
#pragma unroll
				for (int j = 0; j < NCH; ++j) {
					const float4 q = blockRecs[step*NCH + j];
					f[4*j] = q.x; f[4*j + 1] = q.y; f[4*j + 2] = q.z; f[4*j + 3] = q.w;
				}
				int mc = __float_as_int(f[8]);
				mc = (mc < 0) ? 0 : ((mc > CH - 1) ? CH - 1 : mc);
				float2 o1 = h[(i + 7) & 7][0], oL = h[(i + 8 - L) & 7][0];
				float2 p1 = stage[(b + 1) & 127], pL = stage[(b + L) & 127];
				float2 pm = make_float2(f[9], f[10]);
				float sm = f[11];
#pragma unroll
				for (int c = 1; c < CH; ++c) {
					const float2 p1c = stage[c*128 + ((b + 1) & 127)], pLc = stage[c*128 + ((b + L) & 127)];
					if (c == mc) {
						o1 = h[(i + 7) & 7][c]; oL = h[(i + 8 - L) & 7][c]; p1 = p1c; pL = pLc;
						if (CH != 2) { pm = make_float2(f[9 + 3*c], f[10 + 3*c]); sm = f[11 + 3*c]; } // stereo records lead with the maximum channel
					}
				}
				float2 phi = prevHopTerms(p1, make_float2(f[4], f[5]), pL, make_float2(f[6], f[7])); // previous hop's part first (what FOLD0 records pre-compute)
				phi = cfma(oL, make_float2(f[2], f[3]), phi);
				phi = cfma(o1, make_float2(f[0], f[1]), phi); // the newest operand last: two dependent instructions behind it
				const float2 om = (CH == 2) ? makeOutputFb(phi, pm, sm) : makeOutput(phi, pm, sm); // :788 (stereo records carry the fallback output in pm's place)
				const float2 olock = (CH == 2) ? lockedOutput(om, f) : om; // stereo: see recordChannelFields
#pragma unroll
				for (int c = 0; c < CH; ++c) {
					float2 oc;
					if constexpr (CH == 2) {
						oc = olock;
					} else {
						const float2 pc = make_float2(f[9 + 3*c], f[10 + 3*c]);
						oc = makeOutput(cmul(om, cmulc(pc, pm)), pc, f[11 + 3*c]); // channel lock, :791-800
					}
					if (c == mc) oc = om;
					h[i][c] = oc; // bins past the last one have all-zero records, which give exactly zero
					if (k == 0) blockOut[c*BS + step] = oc;
				}
			}
		}
		asm volatile("" ::: "memory");
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if (k == 0) ldsPost(&sync[1], n + 1);
	}
}

// ------------------------------------------------------------------------------------------------------
// K4a: synthesis.  One workgroup per (hop, channel, stream): inverse half-bin-shifted real FFT (gain N),
// multiply by the synthesis window, store the B-sample frame.  Replaces the copy at
// signalsmith-stretch.h:384-394 + stft.synthesiseStep (:397-399).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kSynth(DevBatch d, int sBase, int hopBase) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float2 *bufA = reinterpret_cast<float2 *>(smemRaw);
	float2 *bufB = bufA + d.M;
	const int k = blockIdx.x, c = blockIdx.y, s = blockIdx.z;
	const HopDesc hd = d.hops[(size_t)(sBase + s)*d.hopStride + hopBase + k];
	if (!(hd.flags & HOP_ACTIVE)) return;
	const int B = d.B, H = d.M, N = d.N, halfB = B/2;
	const float2 *X = d.OUT + rowOf(d, s, k, c);
	for (int j = threadIdx.x; j < H; j += blockDim.x) {
		int kk = 2*j;
		bufA[j] = (kk < H) ? X[kk] : cconj(X[N - 1 - kk]);
	}
	__syncthreads();
	float2 *res = fftLds<+1>(bufA, bufB, d.plan, d.twH);
	float *frame = d.frames + ((size_t)((size_t)s*d.T + k)*d.C + c)*(size_t)B;
	const float *__restrict__ win = d.window;
	for (int m = threadIdx.x; m < H; m += blockDim.x) {
		float2 v = cmulcPlain(res[m], d.halfTw[m]); // * e^{+i pi m / N}
		if (m < B - halfB) {
			int i = m + halfB;
			frame[i] = (2*v.x)*win[i];
		}
		if (m >= H - halfB) {
			int i = m - H + halfB;
			frame[i] = (2*v.y)*win[i];
		}
	}
}

// ------------------------------------------------------------------------------------------------------
// K4b: overlap-add as a gather + window-product normalisation + emission (stft.readOutput/moveOutput at
// signalsmith-stretch.h:406-415), and the new carry (the part of the ring that outlives the tile).
// Sums are formed oldest-frame-first, as the reference's ring accumulates them.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kEmit(DevBatch d, IoArgs io, int sBase, int tileIndex) {
	// four consecutive output samples per thread: the frame / window-product taps of a group are 16-byte loads
	// (dword alignment suffices on gfx9), and the two integer divisions are paid once per group
	const int s = blockIdx.z, sg = sBase + s, c = blockIdx.y;
	const EmitDesc ed = d.emit[(size_t)sg*d.emitStride + tileIndex];
	const int span = ed.nHi - ed.nLo;
	const int i0 = 4*(blockIdx.x*blockDim.x + threadIdx.x);
	const int CL = d.carryLen;
	const int total = span + CL;
	if (i0 >= total) return;
	const int B = d.B, I = d.I;
	const size_t carryRow = ((size_t)sg*d.C + c)*(size_t)CL;
	const float *carryWpOld = d.carryWp[d.carryCur] + (size_t)sg*CL;
	float sum[4], wp[4];
	if (i0 + 3 < CL && !d.halfState) {
		const float4 a = *reinterpret_cast<const float4 *>(d.carrySum[d.carryCur] + carryRow + i0), b = *reinterpret_cast<const float4 *>(carryWpOld + i0);
		sum[0] = a.x; sum[1] = a.y; sum[2] = a.z; sum[3] = a.w;
		wp[0] = b.x; wp[1] = b.y; wp[2] = b.z; wp[3] = b.w;
	} else {
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int i = i0 + j;
			sum[j] = (i < CL) ? loadCarrySum(d, d.carryCur, carryRow + i) : 0.0f;
			wp[j] = (i < CL) ? carryWpOld[i] : 1e-30f;
		}
	}
	if (ed.hopCount > 0) {
		// frames q with pos_q <= n < pos_q + B, pos_q = firstHopPos + q*I + delta; summed in ascending q per sample
		const int rel0 = ed.nLo + i0 - ed.firstHopPos - d.delta;
		int qHi = (rel0 + 3 >= 0) ? (rel0 + 3)/I : -1;
		const int qLo = (rel0 - B + 1 > 0) ? (rel0 - B + 1 + I - 1)/I : 0;
		if (qHi > ed.hopCount - 1) qHi = ed.hopCount - 1;
		// all covering frames' loads are issued before the first addition (a loop with a load per iteration costs one
		// memory round trip per frame); the additions then run in ascending q, the order the reference sums in
		constexpr int KMAX = 6;
		float4 f[KMAX], w[KMAX];
		bool fast[KMAX];
#pragma unroll
		for (int k = 0; k < KMAX; ++k) {
			const int q = qLo + k, idx0 = rel0 - q*I;
			fast[k] = q <= qHi && idx0 >= 0 && idx0 + 3 < B;
			const float *frame = d.frames + ((size_t)((size_t)s*d.T + (fast[k] ? q : 0))*d.C + c)*(size_t)B;
			f[k] = *reinterpret_cast<const float4 *>(frame + (fast[k] ? idx0 : 0));
			w[k] = *reinterpret_cast<const float4 *>(d.wprod + (fast[k] ? idx0 : 0));
		}
#pragma unroll
		for (int k = 0; k < KMAX; ++k) {
			const int q = qLo + k, idx0 = rel0 - q*I;
			if (fast[k]) {
				sum[0] += f[k].x; sum[1] += f[k].y; sum[2] += f[k].z; sum[3] += f[k].w;
				wp[0] += w[k].x; wp[1] += w[k].y; wp[2] += w[k].z; wp[3] += w[k].w;
			} else if (q <= qHi) { // a group that straddles a frame edge
				const float *frame = d.frames + ((size_t)((size_t)s*d.T + q)*d.C + c)*(size_t)B;
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const int idx = idx0 + j;
					if (idx >= 0 && idx < B) { sum[j] += frame[idx]; wp[j] += d.wprod[idx]; }
				}
			}
		}
		for (int q = qLo + KMAX; q <= qHi; ++q) { // more than KMAX covering frames (block/interval > 5): plain loop
			const int idx0 = rel0 - q*I;
			const float *frame = d.frames + ((size_t)((size_t)s*d.T + q)*d.C + c)*(size_t)B;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const int idx = idx0 + j;
				if (idx >= 0 && idx < B) { sum[j] += frame[idx]; wp[j] += d.wprod[idx]; }
			}
		}
	}
	float *out = io.out + (size_t)sg*io.outStreamStride + (size_t)c*io.outChannelStride + ed.nLo;
	if (i0 + 3 < span) {
		*reinterpret_cast<float4 *>(out + i0) = make_float4(sum[0]/wp[0], sum[1]/wp[1], sum[2]/wp[2], sum[3]/wp[3]);
	} else {
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int i = i0 + j;
			if (i < span) {
				out[i] = sum[j]/wp[j];
			} else if (i < total) {
				storeCarrySum(d, d.carryCur ^ 1, carryRow + (i - span), sum[j]);
				if (c == 0) d.carryWp[d.carryCur ^ 1][(size_t)sg*CL + (i - span)] = wp[j];
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------------
// State that outlives a tile: Band.input / Band.prevInput (= input of the last hop that analysed a new
// spectrum, signalsmith-stretch.h:806-811), Prediction.energy of the last hop (:707), pitch-estimate state.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kCarryFeed(DevBatch d, int sBase, int hopBase, int anyFormants) { // everything that does not depend on the recurrence
	const int s = blockIdx.z, sg = sBase + s, c = blockIdx.y;
	const int b = blockIdx.x*blockDim.x + threadIdx.x;
	const int nh = d.nHops[s];
	if (nh == 0 || b >= d.M) return;
	const int lastNew = d.lastNewHop[s];
	if (lastNew >= 0) {
		float2 v = d.Xcur[rowOf(d, s, lastNew, c) + b];
		d.stInput[stateRow(d, sg, c) + b] = v;
		d.stPrev[stateRow(d, sg, c) + b] = v;
	}
	{
		const HopDesc hl = d.hops[(size_t)sg*d.hopStride + hopBase + nh - 1];
		const bool plain = !(hl.flags & (HOP_MAPPED | HOP_FORMANTS));
		storeCarriedEnergy(d, stateRow(d, sg, c) + b, plain ? cnorm(inputRow(d, hl, s, sg, c)[b]) : d.PE[rowOf(d, s, nh - 1, c) + b].e);
	}
	if (anyFormants && b == 0 && c == 0) { // a serial walk over the tile's hops: skipped for tiles without formant processing
		float w = d.stFreq[2*sg], wt = d.stFreq[2*sg + 1];
		bool any = false;
		for (int j = 0; j < nh; ++j) {
			const HopDesc hj = d.hops[(size_t)sg*d.hopStride + hopBase + j];
			if (!(hj.flags & HOP_FORMANTS) || d.params[sg].formantBaseFreq > 0) continue;
			w += (d.est[((size_t)s*d.T + j)*2] - w)*0.25f;
			wt += (d.est[((size_t)s*d.T + j)*2 + 1] - wt)*0.25f;
			any = true;
		}
		if (any) { d.stFreq[2*sg] = w; d.stFreq[2*sg + 1] = wt; }
	}
}

// Band.output after the tile's last hop (signalsmith-stretch.h:788-800 leave it in the Band array): runs behind the recurrence
__global__ __launch_bounds__(256) void kCarryOut(DevBatch d, int sBase) {
	const int s = blockIdx.z, sg = sBase + s, c = blockIdx.y;
	const int b = blockIdx.x*blockDim.x + threadIdx.x;
	const int nh = d.nHops[s];
	if (nh == 0 || b >= d.M) return;
	storeCarriedOutput(d, stateRow(d, sg, c) + b, d.OUT[rowOf(d, s, nh - 1, c) + b]);
}

// Input history for the next call: the last B+I samples of (history ++ this call's input)  (copyInput, :215-229,:418)
__global__ __launch_bounds__(256) void kHistory(DevBatch d, IoArgs io) {
	const int sg = blockIdx.z, c = blockIdx.y;
	const int j = blockIdx.x*blockDim.x + threadIdx.x;
	const int HL = d.histLen;
	if (j >= HL) return;
	const int n = io.inSamples[sg];
	const size_t row = ((size_t)sg*d.C + c)*(size_t)HL;
	const int rel = n - HL + j;
	float v;
	if (rel >= 0) v = io.in[(size_t)sg*io.inStreamStride + (size_t)c*io.inChannelStride + rel];
	else v = d.hist[d.histCur][row + HL + rel];
	d.hist[d.histCur ^ 1][row + j] = v;
}

// Silence pass-through (signalsmith-stretch.h:252-267): outputs[c][i] = inputs[c][i % inputSamples] (or 0)
__global__ __launch_bounds__(256) void kPassThrough(DevBatch d, IoArgs io, const int *__restrict__ passFlags) {
	const int sg = blockIdx.z, c = blockIdx.y;
	if (!passFlags[sg]) return;
	const int nOut = io.outSamples[sg], nIn = io.inSamples[sg];
	const float *x = io.in + (size_t)sg*io.inStreamStride + (size_t)c*io.inChannelStride;
	float *y = io.out + (size_t)sg*io.outStreamStride + (size_t)c*io.outChannelStride;
	for (int i = blockIdx.x*blockDim.x + threadIdx.x; i < nOut; i += gridDim.x*blockDim.x) {
		y[i] = (nIn > 0) ? x[i%nIn] : 0.0f;
	}
}

// reset() / flush() / first silent block (signalsmith-stretch.h:49-60, :456-463, :244-251) for the selected streams in ONE
// launch.  Per-stream bit mask: 1 = stft.reset(0.1) (overlap-add sums and input history cleared, window products re-seeded,
// both halves of the double buffers), 2 / 4 / 8 = clear Band.input / .prevInput / .output.
__global__ __launch_bounds__(256) void kResetStreams(DevBatch d, const int *__restrict__ flags, int allBits, const float *__restrict__ seedWp) {
	const int sg = blockIdx.y;
	const int bits = flags ? flags[sg] : allBits;
	if (!bits) return;
	const int i = blockIdx.x*blockDim.x + threadIdx.x;
	const int CL = d.carryLen, HL = d.histLen, M = d.M, C = d.C;
	if (bits & 1) {
		if (i < CL) {
			const float w = seedWp[i];
			d.carryWp[0][(size_t)sg*CL + i] = w;
			d.carryWp[1][(size_t)sg*CL + i] = w;
		}
		for (int c = 0; c < C; ++c) {
			if (i < CL) {
				storeCarrySum(d, 0, ((size_t)sg*C + c)*CL + i, 0.0f);
				storeCarrySum(d, 1, ((size_t)sg*C + c)*CL + i, 0.0f);
			}
			if (i < HL) {
				d.hist[0][((size_t)sg*C + c)*HL + i] = 0.0f;
				d.hist[1][((size_t)sg*C + c)*HL + i] = 0.0f;
			}
		}
	}
	if (i < M && (bits & 14)) {
		const float2 zero = make_float2(0.f, 0.f);
		for (int c = 0; c < C; ++c) {
			const size_t o = stateRow(d, sg, c) + i;
			if (bits & 2) d.stInput[o] = zero;
			if (bits & 4) d.stPrev[o] = zero;
			if (bits & 8) storeCarriedOutput(d, o, zero);
		}
	}
}

// seek(): history = the last B+I samples of the (zero-padded) pre-roll  (signalsmith-stretch.h:140-158)
__global__ __launch_bounds__(256) void kSeekHistory(DevBatch d, IoArgs io, const int *__restrict__ seekFlags) {
	const int sg = blockIdx.z, c = blockIdx.y;
	const int j = blockIdx.x*blockDim.x + threadIdx.x;
	const int HL = d.histLen;
	if (j >= HL) return;
	const size_t row = ((size_t)sg*d.C + c)*(size_t)HL;
	float v;
	if (seekFlags[sg]) {
		const int n = io.inSamples[sg];
		const int rel = n - HL + j;
		v = (rel >= 0) ? io.in[(size_t)sg*io.inStreamStride + (size_t)c*io.inChannelStride + rel] : 0.0f;
	} else {
		v = d.hist[d.histCur][row + j];
	}
	d.hist[d.histCur ^ 1][row + j] = v;
}

// flush() tail (signalsmith-stretch.h:442-455): finishOutput(1) = running maximum of the window products from the
// read position, then out[i] = ring[i]/wp[i] - ring[2*tail-1-i]/wp[2*tail-1-i] for i < tail.  One thread per
// (stream): the running maximum is a serial scan over at most B entries.
__global__ __launch_bounds__(64) void kFlushTail(DevBatch d, IoArgs io, const int *__restrict__ tailOffset, const int *__restrict__ outOffset) {
	const int sg = blockIdx.x;
	const int tail = io.outSamples[sg];
	if (tail < 0) return; // stream not part of this flush
	const int off = tailOffset[sg]; // where the L1 read position sits relative to our carry (split mode: I - samplesSinceLast)
	const int CL = d.carryLen, B = d.B;
	float *wpRow = d.carryWp[d.carryCur] + (size_t)sg*CL;
	if (threadIdx.x == 0) {
		float mx = 0;
		for (int i = 0; i < B; ++i) {
			int idx = off + i;
			float wp = (idx < CL) ? wpRow[idx] : 1e-30f;
			mx = fmaxf(wp, mx);
			if (idx < CL) wpRow[idx] = wp + (mx - wp)*1.0f;
		}
	}
	__syncthreads();
	for (int c = 0; c < d.C; ++c) {
		const size_t sumRow = ((size_t)sg*d.C + c)*(size_t)CL;
		float *y = io.out + (size_t)sg*io.outStreamStride + (size_t)c*io.outChannelStride + outOffset[sg];
		for (int i = threadIdx.x; i < tail; i += blockDim.x) {
			int a = off + i, r = off + 2*tail - 1 - i;
			float va = (a < CL) ? loadCarrySum(d, d.carryCur, sumRow + a)/wpRow[a] : 0.0f;
			float vr = (r < CL) ? loadCarrySum(d, d.carryCur, sumRow + r)/wpRow[r] : 0.0f;
			y[i] = va - vr;
		}
	}
}

// outputSeek() pre-roll fold-back (signalsmith-stretch.h:198-203): negate, reverse, stft.addOutput
__global__ __launch_bounds__(256) void kAddPreRoll(DevBatch d, const float *__restrict__ preRoll, int length, const int *__restrict__ offsets) {
	const int sg = blockIdx.z, c = blockIdx.y;
	const int i = blockIdx.x*blockDim.x + threadIdx.x;
	if (i >= length) return;
	const int CL = d.carryLen;
	const int idx = offsets[sg] + i;
	if (idx >= CL) return;
	const float v = -preRoll[((size_t)sg*d.C + c)*(size_t)length + (length - 1 - i)];
	const size_t e = ((size_t)sg*d.C + c)*(size_t)CL + idx;
	storeCarrySum(d, d.carryCur, e, loadCarrySum(d, d.carryCur, e) + v*d.carryWp[d.carryCur][(size_t)sg*CL + idx]);
}

// Self-test of smst_complex.h (the packed-f32 helpers are inline assembly: their operand selects and negations are checked
// against the documented formulas on the device they ship for).  in: n triples (a, b, c) of complex values + one fraction each
// (7 floats); out: cmul, cmulc, cfma, clerp (8 floats).
__global__ __launch_bounds__(64) void kComplexSelfTest(const float *__restrict__ in, float *__restrict__ out, int n) {
	const int i = blockIdx.x*blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float *v = in + (size_t)7*i;
	const float2 a = make_float2(v[0], v[1]), b = make_float2(v[2], v[3]), c = make_float2(v[4], v[5]);
	const float2 r0 = cmul(a, b), r1 = cmulc(a, b), r2 = cfma(a, b, c), r3 = clerp(a, b, v[6]);
	float *o = out + (size_t)8*i;
	o[0] = r0.x; o[1] = r0.y; o[2] = r1.x; o[3] = r1.y; o[4] = r2.x; o[5] = r2.y; o[6] = r3.x; o[7] = r3.y;
}
void launchComplexSelfTest(const float *in, float *out, int n, hipStream_t st) {
	hipLaunchKernelGGL(kComplexSelfTest, dim3((n + 63)/64), dim3(64), 0, st, in, out, n);
}

// ------------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------------
static inline int divUp(int a, int b) { return (a + b - 1)/b; }

static std::atomic<long long> gLaunchCounts[LK_COUNT];
static const char *const kLaunchNames[LK_COUNT] = {
	"vocoder_aligned", "vocoder_staged", "vocoder_gather", "vocoder_n", "vocoder_one", "vocoder_across", "chain_unfused",
	"analyse_teams", "analyse_fast", "analyse_generic", "synth_teams", "synth_fast", "synth_generic", "synth_emit"};
static inline void countLaunch(LaunchKind k) { gLaunchCounts[k].fetch_add(1, std::memory_order_relaxed); }
long long launchCount(const char *name) {
	for (int i = 0; i < LK_COUNT; ++i) if (name && std::strcmp(name, kLaunchNames[i]) == 0) return gLaunchCounts[i].load(std::memory_order_relaxed);
	return -1;
}

void launchEnergy(const DevBatch &d, const IoArgs &io, int sBase, int nStreams, float *energyOut, hipStream_t st) {
	hipLaunchKernelGGL(kEnergy, dim3(nStreams, kEnergyParts), dim3(256), 256*sizeof(float), st, d, io, sBase, energyOut);
}
void launchAnalyse(const DevBatch &d, const IoArgs &io, int sBase, int nStreams, int hopBase, int tileHops, bool anyInCall, bool anyLate, hipStream_t st) {
	const dim3 grid(tileHops, d.C*2, nStreams);
	const size_t fastLds = ((size_t)d.M + d.M/16)*sizeof(float2);
	// persistent teams pay a 76-KB table copy per workgroup: only where every team gets a few frames
	const bool teams = !d.noFastFft && d.fftTeams && !d.fftLean && anyInCall && (d.M == 256*10 || d.M == 256*12) && (d.fftTeams == 2 || tileHops*d.C*2*nStreams >= 6*d.teamsGrid);
	if (teams) {
		const int jobs = tileHops*d.C*2*nStreams;
		const int wgs = std::max(8, std::min((jobs + 2)/3/8*8, d.teamsGrid)); // one workgroup per CU, a multiple of 8 (one residue class of the job order per XCD)
		const size_t lds = ((size_t)d.M + d.M/2 + d.M/32)*sizeof(float4) + 3*fastLds + 64; // window, first- and second-stage twiddles, a buffer per team, barrier words
		const WindowPad pad = windowPad(d.B, d.M);
		const bool slots = pad.lo == 0 && pad.hi == 0;
		if (d.M == 256*10) {
			if (slots) hipLaunchKernelGGL((kAnalyseTeams<10, 3, true>), dim3(wgs), dim3(768), lds, st, d, io, d.hops, sBase, hopBase, tileHops, nStreams);
			else hipLaunchKernelGGL((kAnalyseTeams<10, 3, false>), dim3(wgs), dim3(768), lds, st, d, io, d.hops, sBase, hopBase, tileHops, nStreams);
		} else {
			if (slots) hipLaunchKernelGGL((kAnalyseTeams<12, 3, true>), dim3(wgs), dim3(768), lds, st, d, io, d.hops, sBase, hopBase, tileHops, nStreams);
			else hipLaunchKernelGGL((kAnalyseTeams<12, 3, false>), dim3(wgs), dim3(768), lds, st, d, io, d.hops, sBase, hopBase, tileHops, nStreams);
		}
		countLaunch(LK_ANALYSE_TEAMS);
		if (!anyLate) return;
	}
	const int lateOnly = teams ? 1 : 0; // the frames whose windows reach into the carried history
	if (!d.noFastFft && d.M%256 == 0 && (d.M/256 == 10 || d.M/256 == 12 || d.M/256 == 20 || d.M/256 == 24)) countLaunch(LK_ANALYSE_FAST); else countLaunch(LK_ANALYSE_GENERIC);
	if (!d.noFastFft) { // every preset: presetCheaper at 44.1 / 48 kHz, presetDefault at 44.1 / 48 kHz, presetCheaper at 88.2 / 96 kHz, presetDefault at 88.2 / 96 kHz
		if (d.M == 256*10) { if (d.fftLean) hipLaunchKernelGGL((kAnalyseFast<10, true>), grid, dim3(256), fastLds, st, d, io, sBase, hopBase, lateOnly); else hipLaunchKernelGGL((kAnalyseFast<10, false>), grid, dim3(256), fastLds, st, d, io, sBase, hopBase, lateOnly); return; }
		if (d.M == 256*12) { if (d.fftLean) hipLaunchKernelGGL((kAnalyseFast<12, true>), grid, dim3(256), fastLds, st, d, io, sBase, hopBase, lateOnly); else hipLaunchKernelGGL((kAnalyseFast<12, false>), grid, dim3(256), fastLds, st, d, io, sBase, hopBase, lateOnly); return; }
		if (d.M == 256*20) { if (d.fftLean) hipLaunchKernelGGL((kAnalyseFast<20, true>), grid, dim3(320), fastLds, st, d, io, sBase, hopBase, lateOnly); else hipLaunchKernelGGL((kAnalyseFast<20, false>), grid, dim3(320), fastLds, st, d, io, sBase, hopBase, lateOnly); return; }
		if (d.M == 256*24) { if (d.fftLean) hipLaunchKernelGGL((kAnalyseFast<24, true>), grid, dim3(384), fastLds, st, d, io, sBase, hopBase, lateOnly); else hipLaunchKernelGGL((kAnalyseFast<24, false>), grid, dim3(384), fastLds, st, d, io, sBase, hopBase, lateOnly); return; }
	}
	size_t lds = 2*(size_t)d.M*sizeof(float2);
	hipLaunchKernelGGL(kAnalyse, grid, dim3(256), lds, st, d, io, sBase, hopBase);
}
// returns true if pass A (the (P, E) rows) has been done here: tiles without formant processing, presets' plan sizes
bool launchFeed(const DevBatch &d, int sBase, int nStreams, int hopBase, int tileHops, bool anyFormants, hipStream_t st) {
	if (d.feedSerial) { // bin-by-bin evaluation (SMST_FEED_SERIAL=1)
		hipLaunchKernelGGL(kFeedEnergy, dim3(divUp(d.M, 64), nStreams), dim3(256), 64*65*sizeof(float), st, d, sBase, hopBase);
		hipLaunchKernelGGL(kFeedSerial, dim3(nStreams), dim3(64), 0, st, d, sBase, hopBase);
		return false;
	}
	const size_t ldsA = (size_t)2*d.M*sizeof(float) + (size_t)(d.M/2 + 2)*sizeof(float2) + 264*sizeof(ScanMap) + 264*sizeof(int);
	const size_t ldsC = (size_t)2*d.M*sizeof(float) + 264*sizeof(ScanMap);
	const int perThread = divUp(d.M, 256); // bins per thread: in registers up to 24 (M <= 6144), through LDS beyond
	const bool fusePassA = !anyFormants && perThread <= 24 && !d.noFeedFusion;
	if (fusePassA) {
		if (perThread <= 16) hipLaunchKernelGGL((kFeedScanA<16, true>), dim3(tileHops, nStreams), dim3(256), ldsA, st, d, sBase, hopBase);
		else hipLaunchKernelGGL((kFeedScanA<24, true>), dim3(tileHops, nStreams), dim3(256), ldsA, st, d, sBase, hopBase);
		return true;
	}
	if (perThread <= 16) hipLaunchKernelGGL(kFeedScanA<16>, dim3(tileHops, nStreams), dim3(256), ldsA, st, d, sBase, hopBase);
	else if (perThread <= 24) hipLaunchKernelGGL(kFeedScanA<24>, dim3(tileHops, nStreams), dim3(256), ldsA, st, d, sBase, hopBase);
	else hipLaunchKernelGGL(kFeedScanA<0>, dim3(tileHops, nStreams), dim3(256), ldsA, st, d, sBase, hopBase);
	if (anyFormants) {
		hipLaunchKernelGGL(kFeedFreq, dim3(divUp(nStreams, 64)), dim3(64), 0, st, d, sBase, nStreams, hopBase);
		if (!d.noFeedFusion) { // tiles with formant processing: pass A at the end of the envelope kernel, the ratios still in LDS
			if (perThread <= 16) hipLaunchKernelGGL((kFeedScanC<16, true>), dim3(tileHops, nStreams), dim3(256), ldsC, st, d, sBase, hopBase);
			else if (perThread <= 24) hipLaunchKernelGGL((kFeedScanC<24, true>), dim3(tileHops, nStreams), dim3(256), ldsC, st, d, sBase, hopBase);
			else hipLaunchKernelGGL((kFeedScanC<0, true>), dim3(tileHops, nStreams), dim3(256), ldsC, st, d, sBase, hopBase);
			return true;
		}
		if (perThread <= 16) hipLaunchKernelGGL(kFeedScanC<16>, dim3(tileHops, nStreams), dim3(256), ldsC, st, d, sBase, hopBase);
		else if (perThread <= 24) hipLaunchKernelGGL(kFeedScanC<24>, dim3(tileHops, nStreams), dim3(256), ldsC, st, d, sBase, hopBase);
		else hipLaunchKernelGGL(kFeedScanC<0>, dim3(tileHops, nStreams), dim3(256), ldsC, st, d, sBase, hopBase);
	}
	return false;
}
template <int CH>
static void launchPredictT(const DevBatch &d, int sBase, int nStreams, int hopBase, int tileHops, bool plain, bool passADone, hipStream_t st) {
	const dim3 grid(divUp(d.M + d.lag*(d.T - 1), 8), nStreams);
	const size_t lds = (size_t)8*((9 + 3*CH + 3)/4)*65*sizeof(float4);
	if (plain) {
		hipLaunchKernelGGL((kPredictB<CH, true>), grid, dim3(256), lds, st, d, sBase, hopBase);
	} else {
		if (!passADone) hipLaunchKernelGGL(kPredictA, dim3(divUp(d.M, 256), tileHops, nStreams), dim3(256), 0, st, d, sBase, hopBase);
		hipLaunchKernelGGL((kPredictB<CH, false>), grid, dim3(256), lds, st, d, sBase, hopBase);
	}
}
// mono / stereo: pass A (only with a pitch map or formants) on the feed-forward stream ...
void launchPredictFused(const DevBatch &d, int sBase, int nStreams, int hopBase, int tileHops, bool plain, bool passADone, hipStream_t st) {
	if (!plain && !passADone) hipLaunchKernelGGL(kPredictA, dim3(divUp(d.M, 256), tileHops, nStreams), dim3(256), 0, st, d, sBase, hopBase);
}
// ... and the fused producer/consumer recurrence
template <int CH, int L>
static void launchVocoderTL(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, bool bounded, hipStream_t st) {
	constexpr int NCH = (9 + 3*CH + 3)/4;
	const size_t fixed = 64 + 128*sizeof(int) + 64*sizeof(HopDesc) + (size_t)kVocOutBlocks*kVocBlockSteps*CH*kVocOutPitch*sizeof(float2);
	const size_t lds = (size_t)kVocBlocks*kVocBlockSteps*NCH*64*sizeof(float4) + fixed;
	// line-aligned producers where they were measured to pay: L = 4 (presetDefault at 48 / 96 kHz: step 14.9 -> 14.5 ms).  At L = 3
	// (presetCheaper) the 8-bin lag costs 9 % more wavefront steps than lag 4 and the two forms tie (10.75 / 10.80 ms per step), so that
	// geometry stays on the staged producers; SMST_ALIGN_ALL=1 takes the aligned form wherever it is valid (L <= 4), for the A/B.
	if constexpr (L <= 4) {
		if (plain && bounded && !d.noStage && !d.noAlign && d.M%16 == 0 && !d.halfState && (L == 4 || d.alignAll)) { // (fp16 state: the staged producers below)
			using G = AlignGeom<CH, L>;
			const size_t fixedA = 64 + 128*sizeof(int) + 64*sizeof(HopDesc) + (size_t)kVocOutBlocksAligned*kVocBlockSteps*CH*kVocOutPitch*sizeof(float2);
			const size_t ldsAligned = (size_t)kVocBlocksStaged*kVocBlockSteps*NCH*64*sizeof(float4) + fixedA + (size_t)kVocStagedProducers*G::PER_PRODUCER*sizeof(float2);
			hipLaunchKernelGGL((kVocoder<CH, true, L, true, false, false, true>), dim3(nStreams), dim3(64*kVocWaves), ldsAligned, st, d, sBase, hopBase, 0, 0);
			countLaunch(LK_VOC_ALIGNED);
			return;
		}
	}
	if constexpr (L <= 5) {
		if (plain && bounded && !d.noStage) {
			using G = StageGeom<CH, L>;
			const size_t ldsStaged = (size_t)kVocBlocksStaged*kVocBlockSteps*NCH*64*sizeof(float4) + fixed + (size_t)kVocStagedProducers*G::ROWS*G::ROWLEN*sizeof(float2);
			hipLaunchKernelGGL((kVocoder<CH, true, L, true>), dim3(nStreams), dim3(64*kVocWaves), ldsStaged, st, d, sBase, hopBase, 0, 0);
			countLaunch(LK_VOC_STAGED);
			return;
		}
	}
	countLaunch(LK_VOC_GATHER);
	if (plain) { hipLaunchKernelGGL((kVocoder<CH, true, L, false>), dim3(nStreams), dim3(64*kVocWaves), lds, st, d, sBase, hopBase, 0, 0); return; }
	// mapped tiles: the hop rotation table beside the rings when the CU's 160 KB hold it (presetDefault: 3073 bins, 24 KB)
	const size_t ldsRot = lds + (size_t)d.M*sizeof(float2);
	if (ldsRot <= (size_t)160*1024) hipLaunchKernelGGL((kVocoder<CH, false, L, false, true>), dim3(nStreams), dim3(64*kVocWaves), ldsRot, st, d, sBase, hopBase, 0, 0);
	else hipLaunchKernelGGL((kVocoder<CH, false, L, false>), dim3(nStreams), dim3(64*kVocWaves), lds, st, d, sBase, hopBase, 0, 0);
}
// single-hop tiles, mono / stereo: rows of the recurrence wave are streams (kVocoder ACROSS).  Rows per workgroup: enough to cover the
// streams with one workgroup per CU (a multiple of 8: a producer pass is 8 rows x 8 steps), at most 64
template <int CH, int L>
static void launchVocoderAcrossL(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st) {
	constexpr int NCH = (9 + 3*CH + 3)/4;
	const size_t fixed = 64 + 128*sizeof(int) + 64*sizeof(HopDesc) + (size_t)kVocOutBlocks*kVocBlockSteps*CH*kVocOutPitch*sizeof(float2);
	const size_t lds = (size_t)kVocBlocks*kVocBlockSteps*NCH*64*sizeof(float4) + fixed;
	int rows = ((nStreams + 255)/256 + 7) & ~7;
	rows = rows < 8 ? 8 : (rows > 64 ? 64 : rows);
	const dim3 grid((nStreams + rows - 1)/rows);
	if (plain) { hipLaunchKernelGGL((kVocoder<CH, true, L, false, false, true>), grid, dim3(64*kVocWaves), lds, st, d, sBase, hopBase, rows, nStreams); return; }
	const size_t ldsRot = lds + (size_t)d.M*sizeof(float2);
	if (ldsRot <= (size_t)160*1024) hipLaunchKernelGGL((kVocoder<CH, false, L, false, true, true>), grid, dim3(64*kVocWaves), ldsRot, st, d, sBase, hopBase, rows, nStreams);
	else hipLaunchKernelGGL((kVocoder<CH, false, L, false, false, true>), grid, dim3(64*kVocWaves), lds, st, d, sBase, hopBase, rows, nStreams);
}
template <int CH>
static void launchVocoderAcrossT(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st) {
	switch (d.L) {
	case 2: launchVocoderAcrossL<CH, 2>(d, sBase, nStreams, hopBase, plain, st); break;
	case 3: launchVocoderAcrossL<CH, 3>(d, sBase, nStreams, hopBase, plain, st); break;
	case 4: launchVocoderAcrossL<CH, 4>(d, sBase, nStreams, hopBase, plain, st); break;
	default: launchVocoderAcrossL<CH, 5>(d, sBase, nStreams, hopBase, plain, st); break;
	}
}
bool acrossSupported(const DevBatch &d) { return d.C <= 2 && d.lag == d.L + 1 && d.L >= 2 && d.L <= 5; }
void launchVocoderAcross(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st) {
	countLaunch(LK_VOC_ACROSS);
	if (d.C == 1) launchVocoderAcrossT<1>(d, sBase, nStreams, hopBase, plain, st);
	else launchVocoderAcrossT<2>(d, sBase, nStreams, hopBase, plain, st);
}
template <int CH>
static void launchVocoderT(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, bool bounded, hipStream_t st) {
	switch (d.L) { // longVerticalStep = round(fftSamples/interval): 4 (presetDefault @48k), 5 (@44.1k), 3 (presetCheaper)
	case 3: launchVocoderTL<CH, 3>(d, sBase, nStreams, hopBase, plain, bounded, st); break;
	case 4: launchVocoderTL<CH, 4>(d, sBase, nStreams, hopBase, plain, bounded, st); break;
	case 5: launchVocoderTL<CH, 5>(d, sBase, nStreams, hopBase, plain, bounded, st); break;
	case 2: launchVocoderTL<CH, 2>(d, sBase, nStreams, hopBase, plain, bounded, st); break;
	case 6: launchVocoderTL<CH, 6>(d, sBase, nStreams, hopBase, plain, bounded, st); break;
	default: launchVocoderTL<CH, 7>(d, sBase, nStreams, hopBase, plain, bounded, st); break; // only reached with L == 7 (see fusedSupported)
	}
}
template <int CH, int L>
static void launchVocoderNL(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st) {
	constexpr int NCH = (9 + 3*CH + 3)/4;
	const size_t lds = (size_t)kVocNBlocks*kVocNBlockSteps*NCH*64*sizeof(float4) + (size_t)CH*kVocNRing*64*sizeof(float2)
	                   + (size_t)CH*128*sizeof(float2) + 64 + 64*sizeof(HopDesc);
	if (plain) hipLaunchKernelGGL((kVocoderN<CH, true, L>), dim3(nStreams), dim3(64*kVocWaves), lds, st, d, sBase, hopBase);
	else hipLaunchKernelGGL((kVocoderN<CH, false, L>), dim3(nStreams), dim3(64*kVocWaves), lds, st, d, sBase, hopBase);
}
template <int CH>
static void launchVocoderN(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st) {
	switch (d.L) { // longVerticalStep: 3 (presetCheaper), 4 / 5 (presetDefault at 48 / 44.1 kHz); fusedSupported(): 2 <= L <= 5 here
	case 2: launchVocoderNL<CH, 2>(d, sBase, nStreams, hopBase, plain, st); break;
	case 3: launchVocoderNL<CH, 3>(d, sBase, nStreams, hopBase, plain, st); break;
	case 4: launchVocoderNL<CH, 4>(d, sBase, nStreams, hopBase, plain, st); break;
	default: launchVocoderNL<CH, 5>(d, sBase, nStreams, hopBase, plain, st); break;
	}
}
template <int CH, int L>
static void launchVocoderOneL(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st) {
	constexpr int NCH = (9 + 3*CH + 3)/4;
	const size_t lds = (size_t)2*kVocOneBlock*NCH*sizeof(float4) + (size_t)2*CH*kVocOneBlock*sizeof(float2) + (size_t)CH*128*sizeof(float2) + 16 + sizeof(HopDesc);
	if (plain) hipLaunchKernelGGL((kVocoderOne<CH, true, L>), dim3(nStreams), dim3(128), lds, st, d, sBase, hopBase);
	else hipLaunchKernelGGL((kVocoderOne<CH, false, L>), dim3(nStreams), dim3(128), lds, st, d, sBase, hopBase);
}
template <int CH>
static void launchVocoderOneT(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st) {
	switch (d.L) {
	case 2: launchVocoderOneL<CH, 2>(d, sBase, nStreams, hopBase, plain, st); break;
	case 3: launchVocoderOneL<CH, 3>(d, sBase, nStreams, hopBase, plain, st); break;
	case 4: launchVocoderOneL<CH, 4>(d, sBase, nStreams, hopBase, plain, st); break;
	default: launchVocoderOneL<CH, 5>(d, sBase, nStreams, hopBase, plain, st); break;
	}
}
// single-hop tiles (every stream fires at most one hop): see kVocoderOne.  Same geometries as the fused 3-8 channel kernel.
bool singleHopSupported(const DevBatch &d) { return d.lag == d.L + 1 && d.L >= 2 && d.L <= 5; }
void launchVocoderOne(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st) {
	countLaunch(LK_VOC_ONE);
	switch (d.C) {
	case 1: launchVocoderOneT<1>(d, sBase, nStreams, hopBase, plain, st); return;
	case 2: launchVocoderOneT<2>(d, sBase, nStreams, hopBase, plain, st); return;
	case 3: launchVocoderOneT<3>(d, sBase, nStreams, hopBase, plain, st); return;
	case 4: launchVocoderOneT<4>(d, sBase, nStreams, hopBase, plain, st); return;
	case 5: launchVocoderOneT<5>(d, sBase, nStreams, hopBase, plain, st); return;
	case 6: launchVocoderOneT<6>(d, sBase, nStreams, hopBase, plain, st); return;
	case 7: launchVocoderOneT<7>(d, sBase, nStreams, hopBase, plain, st); return;
	default: launchVocoderOneT<8>(d, sBase, nStreams, hopBase, plain, st); return;
	}
}
bool fusedSupported(const DevBatch &d) {
	return d.lag == d.L + 1 && d.L >= 2 && d.L <= (d.C <= 2 ? 7 : 5); // other geometries: kPredictB + kChain (records through HBM)
}
void launchVocoder(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, bool bounded, hipStream_t st) {
	if (d.C > 2) countLaunch(LK_VOC_N);
	switch (d.C) {
	case 1: launchVocoderT<1>(d, sBase, nStreams, hopBase, plain, bounded, st); return;
	case 2: launchVocoderT<2>(d, sBase, nStreams, hopBase, plain, bounded, st); return;
	case 3: launchVocoderN<3>(d, sBase, nStreams, hopBase, plain, st); return;
	case 4: launchVocoderN<4>(d, sBase, nStreams, hopBase, plain, st); return;
	case 5: launchVocoderN<5>(d, sBase, nStreams, hopBase, plain, st); return;
	case 6: launchVocoderN<6>(d, sBase, nStreams, hopBase, plain, st); return;
	case 7: launchVocoderN<7>(d, sBase, nStreams, hopBase, plain, st); return;
	default: launchVocoderN<8>(d, sBase, nStreams, hopBase, plain, st); return;
	}
}
void launchPredict(const DevBatch &d, int sBase, int nStreams, int hopBase, int tileHops, bool plain, bool passADone, hipStream_t st) {
	switch (d.C) {
	case 1: launchPredictT<1>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 2: launchPredictT<2>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 3: launchPredictT<3>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 4: launchPredictT<4>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 5: launchPredictT<5>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 6: launchPredictT<6>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 7: launchPredictT<7>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	default: launchPredictT<8>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	}
}
template <int CH>
static void launchChainT(const DevBatch &d, int sBase, int nStreams, int hopBase, hipStream_t st) {
	size_t lds = ((size_t)CH*d.ringSlots*64 + (size_t)CH*128)*sizeof(float2);
	hipLaunchKernelGGL(kChain<CH>, dim3(nStreams), dim3(64), lds, st, d, sBase, hopBase);
}
void launchChain(const DevBatch &d, int sBase, int nStreams, int hopBase, hipStream_t st) {
	countLaunch(LK_CHAIN_UNFUSED);
	switch (d.C) {
	case 1: launchChainT<1>(d, sBase, nStreams, hopBase, st); break;
	case 2: launchChainT<2>(d, sBase, nStreams, hopBase, st); break;
	case 3: launchChainT<3>(d, sBase, nStreams, hopBase, st); break;
	case 4: launchChainT<4>(d, sBase, nStreams, hopBase, st); break;
	case 5: launchChainT<5>(d, sBase, nStreams, hopBase, st); break;
	case 6: launchChainT<6>(d, sBase, nStreams, hopBase, st); break;
	case 7: launchChainT<7>(d, sBase, nStreams, hopBase, st); break;
	default: launchChainT<8>(d, sBase, nStreams, hopBase, st); break;
	}
}
void launchSynth(const DevBatch &d, int sBase, int nStreams, int hopBase, int tileHops, hipStream_t st) {
	const dim3 grid(tileHops, d.C, nStreams);
	const size_t fastLds = ((size_t)d.M + d.M/16)*sizeof(float2);
	if (!d.noFastFft && d.fftTeams && !d.fftLean && (d.M == 256*10 || d.M == 256*12) && (d.fftTeams == 2 || tileHops*d.C*nStreams >= 6*d.teamsGrid)) {
		const int jobs = tileHops*d.C*nStreams;
		const int wgs = std::max(8, std::min((jobs + 2)/3/8*8, d.teamsGrid));
		const size_t lds = ((size_t)d.M + d.M/2 + d.M/32)*sizeof(float4) + 3*fastLds + 64;
		if (d.M == 256*10) hipLaunchKernelGGL((kSynthTeams<10, 3>), dim3(wgs), dim3(768), lds, st, d, d.hops, sBase, hopBase, tileHops, nStreams);
		else hipLaunchKernelGGL((kSynthTeams<12, 3>), dim3(wgs), dim3(768), lds, st, d, d.hops, sBase, hopBase, tileHops, nStreams);
		countLaunch(LK_SYNTH_TEAMS);
		return;
	}
	if (!d.noFastFft && d.M%256 == 0 && (d.M/256 == 10 || d.M/256 == 12 || d.M/256 == 20 || d.M/256 == 24)) countLaunch(LK_SYNTH_FAST); else countLaunch(LK_SYNTH_GENERIC);
	if (!d.noFastFft) {
		if (d.M == 256*10) { if (d.fftLean) hipLaunchKernelGGL((kSynthFast<10, true>), grid, dim3(256), fastLds, st, d, sBase, hopBase); else hipLaunchKernelGGL((kSynthFast<10, false>), grid, dim3(256), fastLds, st, d, sBase, hopBase); return; }
		if (d.M == 256*12) { if (d.fftLean) hipLaunchKernelGGL((kSynthFast<12, true>), grid, dim3(256), fastLds, st, d, sBase, hopBase); else hipLaunchKernelGGL((kSynthFast<12, false>), grid, dim3(256), fastLds, st, d, sBase, hopBase); return; }
		if (d.M == 256*20) { if (d.fftLean) hipLaunchKernelGGL((kSynthFast<20, true>), grid, dim3(320), fastLds, st, d, sBase, hopBase); else hipLaunchKernelGGL((kSynthFast<20, false>), grid, dim3(320), fastLds, st, d, sBase, hopBase); return; }
		if (d.M == 256*24) { if (d.fftLean) hipLaunchKernelGGL((kSynthFast<24, true>), grid, dim3(384), fastLds, st, d, sBase, hopBase); else hipLaunchKernelGGL((kSynthFast<24, false>), grid, dim3(384), fastLds, st, d, sBase, hopBase); return; }
	}
	size_t lds = 2*(size_t)d.M*sizeof(float2);
	hipLaunchKernelGGL(kSynth, grid, dim3(256), lds, st, d, sBase, hopBase);
}
bool synthEmitApplies(const DevBatch &d, int nStreams, int tileHops) {
	if (d.noFastFft || !d.fftTeams || d.fftLean || !d.synthEmit || !(d.M == 256*10 || d.M == 256*12)) return false;
	if (!(d.delta == 0 || d.delta == d.I)) return false;
	const int QN = d.M == 256*10 ? 3 : 4, SLOTS = d.M == 256*10 ? 8 : 6; // the presets' block / interval ratios (2.5 and 4)
	if (QN*d.I < d.B || d.I > 256*SLOTS || d.B > d.N) return false;
	// one (stream, channel) per team, its hops in sequence.  Measured on 256 CUs (profiles/r4_synth_emit_sweep.txt): ahead of the two
	// kernels from 32 stereo streams on, at every batch size up to 1024 -- also where the last round of teams is mostly empty
	return d.synthEmit == 2 || (tileHops >= 8 && nStreams*d.C >= 64);
}
void launchEmitProducts(const DevBatch &d, int sBase, int nStreams, int tileIndex, hipStream_t st) {
	hipLaunchKernelGGL(kEmitProducts, dim3(divUp(d.wpHeadLen + d.carryLen, 256), nStreams), dim3(256), 0, st, d, sBase, tileIndex);
}
void launchSynthEmit(const DevBatch &d, const IoArgs &io, int sBase, int nStreams, int tileIndex, hipStream_t st) {
	const int items = nStreams*d.C;
	const int wgs = std::min(divUp(items, 2), d.teamsGrid);
	const size_t fastLds = ((size_t)d.M + d.M/16)*sizeof(float2);
	const size_t lds = ((size_t)d.M + d.M/2 + d.M/32)*sizeof(float4) + 2*fastLds + 64;
	const bool split = d.delta != 0;
	if (d.M == 256*10) {
		if (split) hipLaunchKernelGGL((kSynthEmitTeams<10, 3, 8, true>), dim3(wgs), dim3(512), lds, st, d, io, sBase, tileIndex, nStreams);
		else hipLaunchKernelGGL((kSynthEmitTeams<10, 3, 8, false>), dim3(wgs), dim3(512), lds, st, d, io, sBase, tileIndex, nStreams);
	} else {
		if (split) hipLaunchKernelGGL((kSynthEmitTeams<12, 4, 6, true>), dim3(wgs), dim3(512), lds, st, d, io, sBase, tileIndex, nStreams);
		else hipLaunchKernelGGL((kSynthEmitTeams<12, 4, 6, false>), dim3(wgs), dim3(512), lds, st, d, io, sBase, tileIndex, nStreams);
	}
	countLaunch(LK_SYNTH_EMIT);
}
void launchEmit(const DevBatch &d, const IoArgs &io, int sBase, int nStreams, int tileIndex, int maxSpan, hipStream_t st) {
	hipLaunchKernelGGL(kEmit, dim3(divUp(divUp(maxSpan + d.carryLen, 4), 256), d.C, nStreams), dim3(256), 0, st, d, io, sBase, tileIndex);
}
void launchCarryFeed(const DevBatch &d, int sBase, int nStreams, int hopBase, bool anyFormants, hipStream_t st) {
	hipLaunchKernelGGL(kCarryFeed, dim3(divUp(d.M, 256), d.C, nStreams), dim3(256), 0, st, d, sBase, hopBase, anyFormants ? 1 : 0);
}
void launchCarryOut(const DevBatch &d, int sBase, int nStreams, hipStream_t st) {
	hipLaunchKernelGGL(kCarryOut, dim3(divUp(d.M, 256), d.C, nStreams), dim3(256), 0, st, d, sBase);
}
void launchHistory(const DevBatch &d, const IoArgs &io, hipStream_t st) {
	hipLaunchKernelGGL(kHistory, dim3(divUp(d.histLen, 256), d.C, d.S), dim3(256), 0, st, d, io);
}
void launchPassThrough(const DevBatch &d, const IoArgs &io, const int *passFlags, int maxOut, hipStream_t st) {
	int bx = divUp(maxOut, 256);
	if (bx > 64) bx = 64;
	if (bx < 1) bx = 1;
	hipLaunchKernelGGL(kPassThrough, dim3(bx, d.C, d.S), dim3(256), 0, st, d, io, passFlags);
}
void launchResetStreams(const DevBatch &d, const int *flags, int allBits, const float *seedWp, hipStream_t st) {
	const int span = d.carryLen > d.M ? (d.carryLen > d.histLen ? d.carryLen : d.histLen) : (d.M > d.histLen ? d.M : d.histLen);
	hipLaunchKernelGGL(kResetStreams, dim3(divUp(span, 256), d.S), dim3(256), 0, st, d, flags, allBits, seedWp);
}
void launchSeekHistory(const DevBatch &d, const IoArgs &io, const int *seekFlags, hipStream_t st) {
	hipLaunchKernelGGL(kSeekHistory, dim3(divUp(d.histLen, 256), d.C, d.S), dim3(256), 0, st, d, io, seekFlags);
}
void launchFlushTail(const DevBatch &d, const IoArgs &io, const int *tailOffset, const int *outOffset, hipStream_t st) {
	hipLaunchKernelGGL(kFlushTail, dim3(d.S), dim3(64), 0, st, d, io, tailOffset, outOffset);
}

void launchAddPreRoll(const DevBatch &d, const float *preRoll, int length, const int *offsets, hipStream_t st) {
	hipLaunchKernelGGL(kAddPreRoll, dim3(divUp(length, 256), d.C, d.S), dim3(256), 0, st, d, preRoll, length, offsets);
}

} // namespace smst
