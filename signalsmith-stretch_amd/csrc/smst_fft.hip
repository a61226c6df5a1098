// Analysis and synthesis transforms, overlap-add and emission (K1, K4a, K4b, K5) and their launchers.  See smst_kernels_common.h.
#include "smst_kernels_common.h"

namespace smst {

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
	if (!(hd.flags & HOP_ACTIVE) || !(hd.flags & HOP_NEW_SPECTRUM) || (hd.flags & HOP_PREANALYSED)) return;
	if (which == 1 && !(hd.flags & HOP_REANALYSE_PREV)) return;
	if (lateOnly && analysisWindowInCall(d.B, d.M, d.I, hd.inputOffset, which, io.inSamples[sBase + s])) return; // kAnalyseTeams has taken this frame
	const int B = d.B, H = d.M, halfB = B/2, N = d.N;
	const int base = hd.inputOffset - (which ? d.I : 0) - B;
	const float *x = io.in + (size_t)(sBase + s)*io.inStreamStride + (size_t)c*io.inChannelStride;
	const float *hist = d.hist + ((size_t)(sBase + s)*d.C + c)*(size_t)d.histPitch + d.histBase[d.histCur][sBase + s] + d.histLen;
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
		if (!(hd.flags & HOP_ACTIVE) || !(hd.flags & HOP_NEW_SPECTRUM) || (hd.flags & HOP_PREANALYSED) || (which && !(hd.flags & HOP_REANALYSE_PREV)) || !analysisWindowInCall(B, H, d.I, hd.inputOffset, which, io.inSamples[sBase + bc.s])) continue;
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
	const float *carryWpOld = d.carryWp[d.carryCur] + carryWpRow(d, sg) + d.carryBase[d.carryCur][sg];
	const int i = blockIdx.x*blockDim.x + threadIdx.x;
	if (i == 0) d.carryBase[d.carryCur ^ 1][sg] = 0; // the new carry is written from the front of its rows (here and by kSynthEmitTeams)
	if (i < d.wpHeadLen) d.wpHead[(size_t)sg*d.wpHeadLen + i] = windowProductAt(d, ed, carryWpOld, i);
	else if (i - d.wpHeadLen < CL) d.carryWp[d.carryCur ^ 1][carryWpRow(d, sg) + (i - d.wpHeadLen)] = windowProductAt(d, ed, carryWpOld, span + (i - d.wpHeadLen));
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
		const int carryFrom = d.carryBase[d.carryCur][sg];
		const size_t carryRow = carrySumRow(d, sg, c);
		const float *wpOld = d.carryWp[d.carryCur] + carryWpRow(d, sg) + carryFrom;
		const float *wpHead = d.wpHead + (size_t)sg*d.wpHeadLen; // (kEmitProducts; it also writes the new carry's products)
		float *out = io.out + (size_t)sg*io.outStreamStride + (size_t)c*io.outChannelStride;
		auto carryAt = [&](int i) { return i < CL ? loadCarrySum(d, d.carryCur, carryRow + carryFrom + i) : 0.0f; };
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
		// split computation: only blocks whose interval is complete are in the tile (the one in flight runs with a later call), so a call's
		// last tile may end up to an interval behind its last hop's interval: those samples leave like any other interval
		const bool trailing = SPLIT && ed.firstHopPos + cnt*I < ed.nHi;
		if (trailing) emitInterval(cnt);
		const int n0 = ed.firstHopPos + (cnt + (trailing ? 1 : 0))*I; // what the ring still holds: the start of the next tile's sums
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
	// grid (stream, part): partial sums, the host adds the kEnergyParts partials of a stream (short inputs are launched with fewer parts:
	// the slots that are left hold zeros)
	const int s = blockIdx.x, part = blockIdx.y, parts = gridDim.y;
	if (part == 0 && (int)threadIdx.x >= parts && (int)threadIdx.x < kEnergyParts) energyOut[(size_t)(sBase + s)*kEnergyParts + threadIdx.x] = 0.0f;
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
	if (threadIdx.x == 0) energyOut[(size_t)(sBase + s)*kEnergyParts + part] = red[0];
}

// ------------------------------------------------------------------------------------------------------
// K1: analysis.  One workgroup per (hop, channel x {current, previous}, stream):
// gather B samples ending at the hop's input offset from the contiguous per-channel sample block (or the
// carried history for negative indices), multiply by the analysis window, fold into the N/2-point complex
// sequence of the half-bin-shifted real FFT, Stockham FFT in LDS, write the M = N/2 bins.
// Replaces stft.analyseStep at signalsmith-stretch.h:337,:359 (+ the copies :344-350,:366-372).
// ------------------------------------------------------------------------------------------------------
// BIG (blocks whose two ping-pong buffers do not fit a CU's LDS: presetDefault / presetCheaper at 176.4 / 192 kHz, 12288 / 10240 bins): one buffer
// in LDS, the other in memory -- the frame's own output row, which nobody reads before the kernel has finished.  The Stockham passes then
// alternate between LDS and HBM (L2 in practice: 96 KB per workgroup): slower than the LDS-resident form, the same arithmetic in the same order.
template <bool BIG>
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
	if (!(hd.flags & HOP_ACTIVE) || !(hd.flags & HOP_NEW_SPECTRUM) || (hd.flags & HOP_PREANALYSED)) return;
	if (which == 1 && !(hd.flags & HOP_REANALYSE_PREV)) return;

	const int B = d.B, H = d.M, halfB = B/2;
	const int base = hd.inputOffset - (which ? d.I : 0) - B; // index of block element 0 in the call's input
	const float *x = io.in + (size_t)(sBase + s)*io.inStreamStride + (size_t)c*io.inChannelStride;
	const float *hist = d.hist + ((size_t)(sBase + s)*d.C + c)*(size_t)d.histPitch + d.histBase[d.histCur][sBase + s];
	const float *__restrict__ win = d.window;
	float2 *dst = (which ? d.Xprev : d.Xcur) + rowOf(d, s, k, c);
	if (BIG) bufB = dst;

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
	if (BIG && res == dst) { // the last pass ended in the output row: through LDS once more, the final order is a permutation of it
		for (int j = threadIdx.x; j < H; j += blockDim.x) bufA[j] = dst[j];
		__syncthreads();
		res = bufA;
	}
	const int N = d.N;
	for (int j = threadIdx.x; j < H; j += blockDim.x) {
		float2 u = res[j];
		int kk = 2*j;
		if (kk < H) dst[kk] = u;
		else dst[N - 1 - kk] = cconj(u);
	}
}

// ------------------------------------------------------------------------------------------------------
// K4a: synthesis.  One workgroup per (hop, channel, stream): inverse half-bin-shifted real FFT (gain N),
// multiply by the synthesis window, store the B-sample frame.  Replaces the copy at
// signalsmith-stretch.h:384-394 + stft.synthesiseStep (:397-399).
// ------------------------------------------------------------------------------------------------------
template <bool BIG> // BIG: the second buffer in memory (DevBatch::fftScratch, a row per frame), see kAnalyse
__global__ __launch_bounds__(256) void kSynth(DevBatch d, int sBase, int hopBase) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float2 *bufA = reinterpret_cast<float2 *>(smemRaw);
	float2 *bufB = bufA + d.M;
	const int k = blockIdx.x, c = blockIdx.y, s = blockIdx.z;
	if (BIG) bufB = d.fftScratch + rowOf(d, s, k, c);
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
// KMAX = the frames whose loads are issued together: ceil((B + 3)/I) cover a group of four samples (3 for presetCheaper, 5 for presetDefault);
// a slot beyond that is a load instruction for nothing, and the kernel lives on the number of those (EXPERIMENTS.md 6.10).
template <int KMAX>
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
	const int carryFrom = d.carryBase[d.carryCur][sg];
	const size_t carryRow = carrySumRow(d, sg, c);
	const float *carryWpOld = d.carryWp[d.carryCur] + carryWpRow(d, sg) + carryFrom;
	if (i0 == 0 && c == 0) d.carryBase[d.carryCur ^ 1][sg] = 0; // the new carry is written from the front of its rows
	float sum[4], wp[4];
	if (i0 + 3 < CL && !d.halfState) {
		const float4 a = *reinterpret_cast<const float4 *>(d.carrySum[d.carryCur] + carryRow + carryFrom + i0), b = *reinterpret_cast<const float4 *>(carryWpOld + i0);
		sum[0] = a.x; sum[1] = a.y; sum[2] = a.z; sum[3] = a.w;
		wp[0] = b.x; wp[1] = b.y; wp[2] = b.z; wp[3] = b.w;
	} else {
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int i = i0 + j;
			sum[j] = (i < CL) ? loadCarrySum(d, d.carryCur, carryRow + carryFrom + i) : 0.0f;
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
				if (c == 0) d.carryWp[d.carryCur ^ 1][carryWpRow(d, sg) + (i - span)] = wp[j];
			}
		}
	}
}

// A call in which no stream fires a hop (most quanta of the 128-frame real-time pattern): its output is the front of the carried sums over
// their window products -- kEmit's quotient -- and the carry that is left BEGINS LATER in the same rows instead of being copied
// (B+I entries per row and call otherwise).  One workgroup per stream, so nobody reads the window's beginning while it moves.
__global__ __launch_bounds__(256) void kEmitCarried(DevBatch d, IoArgs io) {
	const int sg = blockIdx.x;
	const EmitDesc ed = d.emit[(size_t)sg*d.emitStride];
	const int span = ed.nHi - ed.nLo, CL = d.carryLen, from = d.carryBase[d.carryCur][sg];
	const float *wpRow = d.carryWp[d.carryCur] + carryWpRow(d, sg) + from;
	for (int c = 0; c < d.C; ++c) {
		const size_t row = carrySumRow(d, sg, c) + from;
		float *out = io.out + (size_t)sg*io.outStreamStride + (size_t)c*io.outChannelStride + ed.nLo;
		for (int i = threadIdx.x; i < span; i += blockDim.x) {
			const float sum = (i < CL) ? loadCarrySum(d, d.carryCur, row + i) : 0.0f, wp = (i < CL) ? wpRow[i] : 1e-30f;
			out[i] = sum/wp;
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) d.carryBase[d.carryCur][sg] = from + span;
}

// ------------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------------
void launchEnergy(const DevBatch &d, const IoArgs &io, int sBase, int nStreams, int maxSamples, float *energyOut, hipStream_t st) {
	// a part per 1024 samples of the longest input: a 128-frame quantum of 4096 streams is 4096 workgroups, not 65536 (56 -> 8 us)
	const int parts = std::max(1, std::min(kEnergyParts, maxSamples/1024));
	hipLaunchKernelGGL(kEnergy, dim3(nStreams, parts), dim3(256), 256*sizeof(float), st, d, io, sBase, energyOut);
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
	if (fftNeedsScratch(d.M)) hipLaunchKernelGGL(kAnalyse<true>, grid, dim3(256), lds/2, st, d, io, sBase, hopBase);
	else hipLaunchKernelGGL(kAnalyse<false>, grid, dim3(256), lds, st, d, io, sBase, hopBase);
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
	if (fftNeedsScratch(d.M)) hipLaunchKernelGGL(kSynth<true>, grid, dim3(256), lds/2, st, d, sBase, hopBase);
	else hipLaunchKernelGGL(kSynth<false>, grid, dim3(256), lds, st, d, sBase, hopBase);
}
bool synthEmitApplies(const DevBatch &d, int nStreams, int tileHops) {
	if (d.noFastFft || !d.fftTeams || d.fftLean || !d.synthEmit || !(d.M == 256*10 || d.M == 256*12)) return false;
	if (!(d.delta == 0 || d.delta == d.I)) return false;
	const int QN = d.M == 256*10 ? 3 : 4, SLOTS = d.M == 256*10 ? 8 : 6; // the presets' block / interval ratios (2.5 and 4)
	if (QN*d.I < d.B || d.I > 256*SLOTS || d.B > d.N) return false;
	// the teams read wpHead[off + q*I + r] for q <= QN, r < I, with off = the samples in front of a tile's first hop (< I: a block begins at
	// the latest I - 1 samples into a call, smst_engine.cpp): inside the (ceil(B/I) + 2)*I floats per stream only if QN >= ceil(B/I) -- stated, not assumed
	if ((QN + 2)*d.I > d.wpHeadLen) return false;
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
	const dim3 grid(divUp(divUp(maxSpan + d.carryLen, 4), 256), d.C, nStreams);
	const int covering = divUp(d.B + 3, d.I); // frames that can cover a group of four output samples
	if (covering <= 3) hipLaunchKernelGGL(kEmit<3>, grid, dim3(256), 0, st, d, io, sBase, tileIndex);
	else if (covering <= 4) hipLaunchKernelGGL(kEmit<4>, grid, dim3(256), 0, st, d, io, sBase, tileIndex);
	else if (covering <= 5) hipLaunchKernelGGL(kEmit<5>, grid, dim3(256), 0, st, d, io, sBase, tileIndex);
	else hipLaunchKernelGGL(kEmit<6>, grid, dim3(256), 0, st, d, io, sBase, tileIndex); // (more covering frames: the kernel's plain loop takes the rest)
}
void launchEmitCarried(const DevBatch &d, const IoArgs &io, hipStream_t st) {
	hipLaunchKernelGGL(kEmitCarried, dim3(d.S), dim3(256), 0, st, d, io);
	countLaunch(LK_EMIT_CARRIED);
}

} // namespace smst
