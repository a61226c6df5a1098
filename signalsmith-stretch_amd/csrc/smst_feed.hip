// The feed-forward passes over a hop's spectrum (K2b-e: energies, smoothing, peaks, frequency map, formant envelope) and their launcher.
#include "smst_recurrence.h"

namespace smst {

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
	const StreamParams prm = d.paramsPeaks[sg], prmF0 = d.paramsForm0[sg], prmF2 = d.paramsForm2[sg]; // findPeaks / updateFormants(0) / (2) may see different live values (smst_device.h)
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
					if (mapped) pk[(size_t)nPeaks*64] = make_float2(avgBand, freqToBandDev(mapFreqDev(d, prm, sg, avgFreq), Nf));
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
		const bool autoBase = formants && prmF0.formantBaseFreq <= 0;
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
		float freqEstimate = freqToBandDev(prmF0.formantBaseFreq, Nf); // freqToBand, :982
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
				float outputF = prmF2.formantCompensation ? mapFreqDev(d, prmF2, sg, inputF) : inputF;
				// invMapFormant, :920-925
				if (outputF*prmF2.invFormantMultiplier > prmF2.freqTonalityLimit) outputF = mulAdd2(1 - prmF2.formantMultiplier, prmF2.freqTonalityLimit, outputF);
				else outputF = outputF*prmF2.invFormantMultiplier;
				const float inputE = sT[(size_t)b*64];
				float band = freqToBandDev(outputF, Nf);
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

// The formant envelope of one hop (2 x (down, up) max-decay, 2 x (down, up) min-grow over the channel-summed energies in `en`, :987-1006) and the
// per-bin energy ratio (:1018-1033) -- into `en` in place of the energies (RATIO_TO_LDS: pass A follows in the same kernel) or to the tile's
// ratio rows.  One workgroup of 256 threads; `sm` and `maps` are scratch.  Shared by kFeedScanC and the one-pass form of kFeedScanA.
template <int NMAX, bool FUSE_PE>
__device__ __forceinline__ void formantEnvelopeAndRatio(const DevBatch &d, const StreamParams &prm, int s, int sg, int k, float freqEstimate, float *en, float *sm, ScanMap *maps) {
	const int M = d.M, t = threadIdx.x;
	const float Nf = float(d.N);
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
		if (outputF*prm.invFormantMultiplier > prm.freqTonalityLimit) outputF = mulAdd2(1 - prm.formantMultiplier, prm.freqTonalityLimit, outputF); // invMapFormant, :920-925
		else outputF = outputF*prm.invFormantMultiplier;
		const float inputE = sm[b];
		float band = freqToBandDev(outputF, Nf);
		float targetE = 0;
		if (!(band < 0)) { // getFormant, :1009-1016 (entries M and M+1 of the metric are zero)
			band = fminf(band, float(M));
			const int fl = (int)floorf(band);
			const float fr = band - fl;
			const float low = (fl < M) ? sm[fl] : 0.0f, high = (fl + 1 < M) ? sm[fl + 1] : 0.0f;
			targetE = low + (high - low)*fr;
		}
		if constexpr (FUSE_PE) en[b] = targetE/(inputE + 1e-30f); // the energies are dead: the ratios take their place in LDS
		else { ratio[b] = targetE/(inputE + 1e-30f); if (d.envelope) d.envelope[((size_t)s*d.T + k)*M + b] = inputE; }
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
// FUSE_FORM (tiles WITH formant processing in which no stream estimates its base frequency -- BASELINE config 4 gives 200 Hz): the formant
// envelope, the energy ratio and pass A follow in the SAME kernel, from the channel-summed energies that are still in LDS.  The pitch
// estimate is what forces three kernels otherwise (it is smoothed from hop to hop: a serial walk over the tile's hops, kFeedFreq, between the
// energies and the envelope); with a given base frequency the spectra are read once instead of twice.  Same code (formantEnvelopeAndRatio,
// feedPredictionRows), same values: bit-identical to kFeedScanA + kFeedFreq + kFeedScanC.
template <int NMAX, bool FUSE_PE = false, bool FUSE_FORM = false> // NMAX: bins per thread held in registers during the smoothing passes; 0: through LDS (any M)
__global__ __launch_bounds__(256) void kFeedScanA(DevBatch d, int sBase, int hopBase) {
	static_assert(!FUSE_PE || NMAX > 0, "pass A is folded into the register form only");
	static_assert(!(FUSE_PE && FUSE_FORM), "FUSE_PE: tiles without formant processing; FUSE_FORM: tiles with it");
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	const int k = blockIdx.x, s = blockIdx.y, sg = sBase + s;
	if (k >= d.nHops[s]) return;
	const HopDesc hd = d.hops[(size_t)sg*d.hopStride + hopBase + k];
	const bool mapped = hd.flags & HOP_MAPPED, formants = hd.flags & HOP_FORMANTS;
	if (!mapped && !formants) {
		if constexpr (FUSE_PE || FUSE_FORM) feedPredictionRows(d, hd, s, sg, k, false, [](int bb) { return make_float2(float(bb), 1.0f); }, false, nullptr);
		return;
	}
	const int M = d.M, C = d.C, t = threadIdx.x;
	const float Nf = float(d.N);
	float *en = reinterpret_cast<float *>(smemRaw);         // [M] channel-summed energy
	float *sm = en + M;                                       // [M] smoothed
	float2 *pk = reinterpret_cast<float2 *>(sm + M);          // [M/2 + 2] peaks
	ScanMap *maps = reinterpret_cast<ScanMap *>(pk + M/2 + 2); // [264]
	int *counts = reinterpret_cast<int *>(maps + 264);         // [264]
	const StreamParams prm = d.paramsPeaks[sg];
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
				pk[idx++] = make_float2(avgBand, freqToBandDev(mapFreqDev(d, prm, sg, avgFreq), Nf));
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
	if constexpr (FUSE_FORM) {
		// (every stream of the tile has a base frequency: the host's condition for this form)
		__syncthreads(); // the map row is complete (its stores are visible to the workgroup behind the barrier); `sm` / `cover` and the peaks are dead
		if (formants) {
			formantEnvelopeAndRatio<NMAX, true>(d, d.paramsForm2[sg], s, sg, k, freqToBandDev(d.paramsForm0[sg].formantBaseFreq, Nf), en, sm, maps);
			__syncthreads();
		}
		const float2 *mapRowIn = d.map + ((size_t)s*d.T + k)*M;
		feedPredictionRows(d, hd, s, sg, k, mapped, [&](int bb) { return mapRowIn[bb]; }, false, formants ? en : nullptr);
		return;
	}
	if (formants && d.paramsForm0[sg].formantBaseFreq <= 0) {
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
	const StreamParams prm = d.paramsForm0[sg];
	const float Nf = float(d.N);
	float w = d.stFreq[2*sg], wt = d.stFreq[2*sg + 1];
	for (int j = 0; j < nh; ++j) {
		const HopDesc hj = d.hops[(size_t)sg*d.hopStride + hopBase + j];
		float fe = freqToBandDev(prm.formantBaseFreq, Nf); // freqToBand, :982
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
	const StreamParams prm = d.paramsForm2[sg];
	feedEnergyToLds(d, hd, s, sg, en);
	__syncthreads();
	formantEnvelopeAndRatio<NMAX, FUSE_PE>(d, prm, s, sg, k, d.freqEst[(size_t)s*d.T + k], en, sm, maps);
	if constexpr (FUSE_PE) {
		__syncthreads();
		const float2 *mapRowIn = d.map + ((size_t)s*d.T + k)*M;
		feedPredictionRows(d, hd, s, sg, k, (hd.flags & HOP_MAPPED) != 0, [&](int bb) { return mapRowIn[bb]; }, false, en);
	}
}

// ------------------------------------------------------------------------------------------------------
// host-side launcher
// ------------------------------------------------------------------------------------------------------
// returns true if pass A (the (P, E) rows) has been done here: tiles without formant processing, presets' plan sizes
bool launchFeed(const DevBatch &d, int sBase, int nStreams, int hopBase, int tileHops, bool anyFormants, bool anyEstimatedBase, hipStream_t st) {
	if (d.feedSerial) { // bin-by-bin evaluation (SMST_FEED_SERIAL=1)
		hipLaunchKernelGGL(kFeedEnergy, dim3(divUp(d.M, 64), nStreams), dim3(256), 64*65*sizeof(float), st, d, sBase, hopBase);
		hipLaunchKernelGGL(kFeedSerial, dim3(nStreams), dim3(64), 0, st, d, sBase, hopBase);
		return false;
	}
	const size_t ldsA = (size_t)2*d.M*sizeof(float) + (size_t)(d.M/2 + 2)*sizeof(float2) + 264*sizeof(ScanMap) + 264*sizeof(int);
	const size_t ldsC = (size_t)2*d.M*sizeof(float) + 264*sizeof(ScanMap);
	const int perThread = divUp(d.M, 256); // bins per thread: in registers up to 24 (M <= 6144), through LDS beyond
	const bool fusePassA = !anyFormants && perThread <= 24 && d.noFeedFusion != 1;
	if (fusePassA) {
		if (perThread <= 16) hipLaunchKernelGGL((kFeedScanA<16, true>), dim3(tileHops, nStreams), dim3(256), ldsA, st, d, sBase, hopBase);
		else hipLaunchKernelGGL((kFeedScanA<24, true>), dim3(tileHops, nStreams), dim3(256), ldsA, st, d, sBase, hopBase);
		return true;
	}
	if (anyFormants && !anyEstimatedBase && perThread <= 24 && d.noFeedFusion == 0) { // formant tiles, every base frequency given: ONE pass over the spectra
		if (perThread <= 16) hipLaunchKernelGGL((kFeedScanA<16, false, true>), dim3(tileHops, nStreams), dim3(256), ldsA, st, d, sBase, hopBase);
		else hipLaunchKernelGGL((kFeedScanA<24, false, true>), dim3(tileHops, nStreams), dim3(256), ldsA, st, d, sBase, hopBase);
		countLaunch(LK_FEED_ONE_PASS);
		return true;
	}
	if (perThread <= 16) hipLaunchKernelGGL(kFeedScanA<16>, dim3(tileHops, nStreams), dim3(256), ldsA, st, d, sBase, hopBase);
	else if (perThread <= 24) hipLaunchKernelGGL(kFeedScanA<24>, dim3(tileHops, nStreams), dim3(256), ldsA, st, d, sBase, hopBase);
	else hipLaunchKernelGGL(kFeedScanA<0>, dim3(tileHops, nStreams), dim3(256), ldsA, st, d, sBase, hopBase);
	if (anyFormants) {
		hipLaunchKernelGGL(kFeedFreq, dim3(divUp(nStreams, 64)), dim3(64), 0, st, d, sBase, nStreams, hopBase);
		if (d.noFeedFusion != 1) { // tiles with formant processing: pass A at the end of the envelope kernel, the ratios still in LDS (SMST_NO_FEED_FUSION=2: this form even where one pass would do)
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

} // namespace smst
