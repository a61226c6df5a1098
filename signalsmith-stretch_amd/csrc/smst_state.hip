// State that outlives a tile or a call (carry, history, reset, seek, flush, pre-roll, pass-through), the launch counters, the self-test of smst_complex.h.
#include "smst_recurrence.h"

namespace smst {

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
			if (!(hj.flags & HOP_FORMANTS) || d.paramsForm0[sg].formantBaseFreq > 0) continue;
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

// Input history for the next call: the last B+I samples of (history ++ this call's input)  (copyInput, :215-229,:418).  The window
// slides: a short call appends its input behind the window (n samples written instead of B+I copied -- the 128-frame real-time
// pattern spent a quarter of its no-hop quanta on that copy); when the row is full the window moves back to its front.  Moving is
// safe in place: it happens only when base + n > B+I, so what is read lies behind what is written.
__global__ __launch_bounds__(256) void kHistory(DevBatch d, IoArgs io) {
	const int sg = blockIdx.z, c = blockIdx.y;
	const int j = blockIdx.x*blockDim.x + threadIdx.x;
	const int HL = d.histLen;
	const int n = io.inSamples[sg], base = d.histBase[d.histCur][sg];
	const bool append = base + HL + n <= d.histPitch;
	if (j == 0 && c == 0) d.histBase[d.histCur ^ 1][sg] = append ? base + n : 0;
	float *row = d.hist + ((size_t)sg*d.C + c)*(size_t)d.histPitch;
	const float *x = io.in + (size_t)sg*io.inStreamStride + (size_t)c*io.inChannelStride;
	if (append) {
		if (j < n) row[base + HL + j] = x[j];
		return;
	}
	if (j >= HL) return;
	const int rel = n - HL + j;
	row[j] = (rel >= 0) ? x[rel] : row[base + HL + rel];
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
__global__ __launch_bounds__(256) void kResetStreams(DevBatch d, const int *__restrict__ flags, int allBits, const float *__restrict__ seedWp, const int *__restrict__ keep) {
	const int sg = blockIdx.y;
	const int bits = flags ? flags[sg] : allBits;
	if (!bits) return;
	const int i = blockIdx.x*blockDim.x + threadIdx.x;
	const int CL = d.carryLen, HL = d.histLen, M = d.M, C = d.C;
	if (bits & 1) {
		// split computation, between two interval boundaries: the samples up to the end of the interval are read from the stashed ring, which
		// stft.reset() does not touch (:407-415); the real ring -- re-seeded -- begins behind them
		// (the engine settles the carry before a reset that keeps samples: the window it reads from begins at the front of its rows, so no
		// thread reads what another one writes)
		const int kp = keep ? keep[sg] : 0, cur = d.carryCur, CP = d.carryPitch;
		if (i < CP) {
			const float w = (i < kp) ? d.carryWp[cur][carryWpRow(d, sg) + i] : (i < CL ? seedWp[i - kp] : 1e-30f);
			d.carryWp[0][carryWpRow(d, sg) + i] = w;
			d.carryWp[1][carryWpRow(d, sg) + i] = w;
		}
		if (i == 0) d.carryBase[0][sg] = d.carryBase[1][sg] = 0;
		for (int c = 0; c < C; ++c) {
			if (i < CP) {
				const float v = (i < kp) ? loadCarrySum(d, cur, carrySumRow(d, sg, c) + i) : 0.0f;
				storeCarrySum(d, 0, carrySumRow(d, sg, c) + i, v);
				storeCarrySum(d, 1, carrySumRow(d, sg, c) + i, v);
			}
			if (i < HL) { // (the whole row: the window returns to its front)
				d.hist[((size_t)sg*C + c)*d.histPitch + i] = 0.0f;
				d.hist[((size_t)sg*C + c)*d.histPitch + HL + i] = 0.0f;
			}
		}
		if (i == 0) d.histBase[0][sg] = d.histBase[1][sg] = 0;
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

// split computation: the block in flight was analysed when it started (:293, :332-373); its spectra wait in [S][C][Mp] buffers and enter the
// tile that finally runs the block as row 0 (Xcur: Band.input, Xprev: the re-analysed Band.prevInput)
__global__ __launch_bounds__(256) void kPendingToTile(DevBatch d, int sBase, const float2 *__restrict__ pendIn, const float2 *__restrict__ pendPrev) {
	const int s = blockIdx.z, sg = sBase + s, c = blockIdx.y;
	const int b = blockIdx.x*blockDim.x + threadIdx.x;
	if (d.nHops[s] == 0 || b >= d.M || !(d.hops[(size_t)sg*d.hopStride].flags & HOP_PREANALYSED)) return; // (such a block is always its stream's first hop of the call)
	const size_t src = ((size_t)sg*d.C + c)*(size_t)d.Mp + b, dst = rowOf(d, s, 0, c) + b;
	d.Xcur[dst] = pendIn[src];
	d.Xprev[dst] = pendPrev[src];
}
// ... and a flush() between two of its synthesis steps (:397-399): the channels that had not been synthesised add nothing afterwards
// (stft.reset() cleared the spectrum they would have been made from)
__global__ __launch_bounds__(256) void kMaskOutRows(DevBatch d, int sBase, const int *__restrict__ synthChannels) {
	const int s = blockIdx.z, sg = sBase + s, c = blockIdx.y;
	const int b = blockIdx.x*blockDim.x + threadIdx.x;
	const int from = synthChannels[sg];
	if (d.nHops[s] == 0 || b >= d.M || from < 0 || c < from) return;
	d.OUT[rowOf(d, s, 0, c) + b] = make_float2(0.f, 0.f);
}

// seek(): history = the last B+I samples of the (zero-padded) pre-roll  (signalsmith-stretch.h:140-158), written at the front of the row
__global__ __launch_bounds__(256) void kSeekHistory(DevBatch d, IoArgs io, const int *__restrict__ seekFlags) {
	const int sg = blockIdx.z, c = blockIdx.y;
	const int j = blockIdx.x*blockDim.x + threadIdx.x;
	const int HL = d.histLen;
	if (j == 0 && c == 0) d.histBase[d.histCur ^ 1][sg] = seekFlags[sg] ? 0 : d.histBase[d.histCur][sg];
	if (j >= HL || !seekFlags[sg]) return;
	const int n = io.inSamples[sg];
	const int rel = n - HL + j;
	d.hist[((size_t)sg*d.C + c)*(size_t)d.histPitch + j] = (rel >= 0) ? io.in[(size_t)sg*io.inStreamStride + (size_t)c*io.inChannelStride + rel] : 0.0f;
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
	float *wpRow = d.carryWp[d.carryCur] + carryWpRow(d, sg); // (settled by the engine: the window begins at the front of its rows)
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
		const size_t sumRow = carrySumRow(d, sg, c);
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
	const size_t e = carrySumRow(d, sg, c) + idx; // (settled by the engine: the window begins at the front of its rows)
	storeCarrySum(d, d.carryCur, e, loadCarrySum(d, d.carryCur, e) + v*d.carryWp[d.carryCur][carryWpRow(d, sg) + idx]);
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

static std::atomic<long long> gLaunchCounts[LK_COUNT];
static const char *const kLaunchNames[LK_COUNT] = {
	"vocoder_aligned", "vocoder_staged", "vocoder_gather", "vocoder_n", "vocoder_one", "vocoder_across", "vocoder_continuous", "chain_unfused",
	"analyse_teams", "analyse_fast", "analyse_generic", "synth_teams", "synth_fast", "synth_generic", "synth_emit", "emit_carried", "feed_one_pass"};
void countLaunch(LaunchKind k) { gLaunchCounts[k].fetch_add(1, std::memory_order_relaxed); }
long long launchCount(const char *name) {
	for (int i = 0; i < LK_COUNT; ++i) if (name && std::strcmp(name, kLaunchNames[i]) == 0) return gLaunchCounts[i].load(std::memory_order_relaxed);
	return -1;
}

void launchCarryFeed(const DevBatch &d, int sBase, int nStreams, int hopBase, bool anyFormants, hipStream_t st) {
	hipLaunchKernelGGL(kCarryFeed, dim3(divUp(d.M, 256), d.C, nStreams), dim3(256), 0, st, d, sBase, hopBase, anyFormants ? 1 : 0);
}
void launchCarryOut(const DevBatch &d, int sBase, int nStreams, hipStream_t st) {
	hipLaunchKernelGGL(kCarryOut, dim3(divUp(d.M, 256), d.C, nStreams), dim3(256), 0, st, d, sBase);
}
void launchHistory(const DevBatch &d, const IoArgs &io, int span, hipStream_t st) { // span: the most elements any row writes (its input if it appends, B+I if its window moves)
	hipLaunchKernelGGL(kHistory, dim3(divUp(span < 1 ? 1 : span, 256), d.C, d.S), dim3(256), 0, st, d, io);
}
void launchPassThrough(const DevBatch &d, const IoArgs &io, const int *passFlags, int maxOut, hipStream_t st) {
	int bx = divUp(maxOut, 256);
	if (bx > 64) bx = 64;
	if (bx < 1) bx = 1;
	hipLaunchKernelGGL(kPassThrough, dim3(bx, d.C, d.S), dim3(256), 0, st, d, io, passFlags);
}
void launchResetStreams(const DevBatch &d, const int *flags, int allBits, const float *seedWp, hipStream_t st, const int *keep) {
	const int span = d.carryPitch > d.M ? d.carryPitch : d.M; // (carryPitch > histLen)
	hipLaunchKernelGGL(kResetStreams, dim3(divUp(span, 256), d.S), dim3(256), 0, st, d, flags, allBits, seedWp, keep);
}
void launchPendingToTile(const DevBatch &d, int sBase, int nStreams, const float2 *pendIn, const float2 *pendPrev, hipStream_t st) {
	hipLaunchKernelGGL(kPendingToTile, dim3(divUp(d.M, 256), d.C, nStreams), dim3(256), 0, st, d, sBase, pendIn, pendPrev);
}
void launchMaskOutRows(const DevBatch &d, int sBase, int nStreams, const int *synthChannels, hipStream_t st) {
	hipLaunchKernelGGL(kMaskOutRows, dim3(divUp(d.M, 256), d.C, nStreams), dim3(256), 0, st, d, sBase, synthChannels);
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
