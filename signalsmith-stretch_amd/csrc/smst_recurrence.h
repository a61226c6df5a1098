// Device code shared by the recurrence kernels and the feed-forward passes (smst_vocoder.hip, smst_vocoder_n.hip, smst_feed.hip): row lookup,
// interpolated band access, the per-(hop, channel, bin) record that carries everything the bin recurrence needs that does not depend on it,
// makeOutput, the LDS hand-off words.
#pragma once
#include "smst_kernels_common.h"

namespace smst {

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

// freqToBand (f * fftSamples - 0.5, signalsmith-stretch.h:519-522 through stft.freqToBin) and the two frequency maps with the reference's
// roundings: a product and a sum, each rounded.  Written as one expression they contract to a fused multiply-add on the device
// (-ffp-contract=on), which is MORE accurate -- and moves an interpolation position near bin 3000 by an ulp of 2.4e-4 bins against the x86
// build of the reference: on a steep formant envelope (a chirp's) that was 1e-4 of the energy ratio (round 6: case_formant_stages).
__device__ __forceinline__ float freqToBandDev(float freq, float Nf) { return __fadd_rn(__fmul_rn(freq, Nf), -0.5f); }
__device__ __forceinline__ float mulAdd2(float a, float b, float c) { return __fadd_rn(__fmul_rn(a, b), c); } // a*b + c, two roundings
__device__ __forceinline__ float mapFreqDev(const DevBatch &d, const StreamParams &p, int sGlobal, float freq) { // :850-856
	if (p.hasCustomMap) {
		const float *t = d.mapTable + ((size_t)sGlobal*kMapSlots + p.mapSlot)*d.mapTableLen; // row pitch: the longest table of the batch
		const int n = p.mapLen;                                      // this stream's own knots
		float pos = __fadd_rn(__fmul_rn(__fmul_rn(freq, 2.0f), float(n)), -0.5f);
		if (pos <= 0) return mulAdd2(t[1] - t[0], pos, t[0]);
		if (pos >= n - 1) return mulAdd2(t[n - 1] - t[n - 2], pos - (n - 1), t[n - 1]);
		int lo = (int)floorf(pos);
		float fr = pos - lo;
		return mulAdd2(t[lo + 1] - t[lo], fr, t[lo]);
	}
	if (freq > p.freqTonalityLimit) return mulAdd2(p.freqMultiplier - 1, p.freqTonalityLimit, freq);
	return freq*p.freqMultiplier;
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


// Per-channel fields of a record (floats 9..).  Mono: {P, sqrt(E)}.  3 and more channels: see the branch below.  STEREO: the one locked channel's
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
__device__ __forceinline__ void recordChannelFields(float (&f)[NFLOATS], const float2 (&p)[CH], const float (&e)[CH], int mc, float2 PmN = float2{0.f, 0.f}, float eMN = 0.0f) {
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
	} else if constexpr (CH == 1) {
		f[9] = p[0].x; f[10] = p[0].y; f[11] = __builtin_amdgcn_sqrtf(e[0]); // 1-ulp hardware square root
	} else {
		// 3 and more channels (round 6): the stereo record's idea for EVERY locked channel.  9..11 as there (the maximum channel's fallback output
		// and sqrt(E_m)); per channel c two floats X_c = the lock's twist P_c conj(P_m) scaled to sqrt(E_c) / sqrt(E_m |P_c conj(P_m)|^2) -- or, where that
		// norm is below the noise floor (:598-601), the channel's own input scaled to sqrt(E_c), flagged in bit 8 + c of the word that carries the
		// maximum channel.  The recurrence wave then forms  out_c = weak_c ? X_c : out_m X_c : one complex multiply and a select per channel instead
		// of two multiplies, two norms, a reciprocal square root and four selects (~450 -> ~200 instructions per step at 8 channels, on the one wave
		// whose chain is the kernel's serial path).  The norm is E_m |T|^2 instead of |out_m T|^2, as in the stereo record.
		// (PmN, eMN = p[mc], e[mc] as the caller's arg-max loop tracked them: a second chain of selects over the arrays here is turned into a
		// dynamically indexed private array by the compiler -- scratch memory.  Callers with 1 or 2 channels leave them out.)
		const float2 Pm = PmN;
		const float eM = eMN;
		const float sM = __builtin_amdgcn_sqrtf(eM);
		const float2 Fb = cscale(Pm, sM*__builtin_amdgcn_rsqf(cnorm(Pm) + 1e-15f));
		f[9] = Fb.x; f[10] = Fb.y; f[11] = sM;
		unsigned word = unsigned(mc);
#pragma unroll
		for (int c = 0; c < CH; ++c) {
			const float2 T = cmulc(p[c], Pm);
			const float nT = eM*cnorm(T);
			const bool weak = nT <= 1e-15f;
			const float g = __builtin_amdgcn_sqrtf(e[c])*__builtin_amdgcn_rsqf(weak ? cnorm(p[c]) + 1e-15f : nT);
			f[12 + 2*c] = (weak ? p[c].x : T.x)*g;
			f[13 + 2*c] = (weak ? p[c].y : T.y)*g;
			if (weak) word |= 256u << c;
		}
		f[8] = __int_as_float(int(word));
	}
}
// 3 and more channels: channel c's output from the maximum channel's (see recordChannelFields); `word` = the record's float 8 as an integer
template <int NFLOATS>
__device__ __forceinline__ float2 lockedOutputN(float2 om, const float (&f)[NFLOATS], int c, unsigned word) {
	const float2 X = make_float2(f[12 + 2*c], f[13 + 2*c]);
	const float2 t = cmul(om, X);
	const bool weak = (word >> (8 + c)) & 1u;
	return make_float2(weak ? X.x : t.x, weak ? X.y : t.y);
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
	const float *row;    // inside a mapped tile: third float of hop k-1's 12-byte entries (stride 3); a valid row also when `carried`
	const float2 *input; // inside a plain tile: hop k-1's input spectrum; likewise
	// ONE 8-byte load at a selected address, the value picked out of it afterwards -- a three-way branch with a load in every arm ended
	// the basic block at every call: the record producers of kVocoderN waited for memory nine times per record where two or three
	// round trips are what the data dependences ask for (ISA of the round-3 build; EXPERIMENTS.md 4.11).  The load may reach 4 bytes
	// past the element (the next float of the row): the engine allocates stEnergy and PE with that slack.
	struct Raw { float x, y; bool odd; }; // the 8 bytes at the element, not looked at yet
	template <bool PLAIN>
	__device__ __forceinline__ Raw load(const DevBatch &d, int bc) const {
		const bool half = d.halfState;
		const size_t ci = carriedBase + bc;
		const char *const state = reinterpret_cast<const char *>(d.stEnergy);
		const char *const fromState = state + (half ? (ci & ~size_t(1))*2 : ci*4); // fp16: the dword that holds the element
		const char *const fromTile = PLAIN ? reinterpret_cast<const char *>(input + bc) : reinterpret_cast<const char *>(row + (size_t)bc*3);
		const float *const address = reinterpret_cast<const float *>(carried ? fromState : fromTile);
		return Raw{address[0], address[1], bool(ci & 1)};
	}
	// (called once EVERY load of the record's second round trip has been issued: the empty asm statements keep the two loads
	// unconditional -- the compiler otherwise sinks the second one into a conditional of its own and waits for it there -- and nothing
	// is scheduled across them)
	template <bool PLAIN>
	__device__ __forceinline__ float value(const DevBatch &d, Raw r) const {
		keepUnconditional(r.x);
		keepUnconditional(r.y);
		const unsigned bits = (unsigned)__float_as_int(r.x);
		const unsigned short hbits = r.odd ? (unsigned short)(bits >> 16) : (unsigned short)(bits & 0xffffu);
		const float a = float(__builtin_bit_cast(half_t, hbits));
		const float asState = d.halfState ? a*a : r.x;
		return carried ? asState : (PLAIN ? cnorm(make_float2(r.x, r.y)) : r.x);
	}
};
// coefficient multiplying the previous hop's final output at bin bx (bx = b+1 or b+L), see the record description -- in two parts: every
// load and what follows from it except the previous hop's energy (twistParts), and the division by max(E_prev, E_now) (twistFinish),
// so that a record's two twists have ALL their loads in flight before the first of them is waited for
struct TwistParts { float2 r; float eNow; PrevEnergy::Raw ePrev; };
template <int CH, bool PLAIN>
__device__ __forceinline__ TwistParts twistParts(const RecordSource<CH, PLAIN> &src, int mc, int bx, float2 mp, bool rotate, const float2 *in,
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
	const float2 down = cmulc(Px, lerpBand(in, lerpIndex(mp.x - stepMul*tfDown), M));
	TwistParts t;
	t.r = cmulc(TW, down);
	t.eNow = pe.z;
	t.ePrev = prevE.template load<PLAIN>(d, bc);
	return t;
}
template <bool PLAIN>
__device__ __forceinline__ float2 twistFinish(const DevBatch &d, const PrevEnergy &prevE, const TwistParts &t) {
	const float ePrev = prevE.template value<PLAIN>(d, t.ePrev);
	const float den = fmaxf(ePrev, t.eNow) + 1e-15f; // :716
	const float inv = __builtin_amdgcn_rcpf(den); // 1-ulp hardware reciprocal (an IEEE division costs ten instructions per record)
	return make_float2(t.r.x*inv, t.r.y*inv);
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
	// Split computation: a flush() between two chunks of this block's main prediction zeroed the bins the earlier chunks had computed
	// (:458-463) and the later chunks started from those zeros (HopDesc.startBin; 0 in every other hop).  An all-zero record gives
	// exactly zero in every recurrence kernel (outputs and history alike) -- so the bins below startBin keep the zeros the caller put
	// into `f`, and kVocoder's gathering form, kVocoderN, kVocoderOne, ACROSS and kPredictB + kChain all honour it the same way.
	if (b < hd.startBin) return;
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
		float4 pr = *reinterpret_cast<const float4 *>(d.map + ((size_t)s*d.T + k)*M + ci);
		// (only the positions are used here; with the gradients dead the compiler loads the two positions as two separate dwords -- two gather
		// instructions over the same lines, and the gathering producers live on the number of requests their L1 takes: EXPERIMENTS.md 6.9)
		keepUnconditional(pr.y);
		keepUnconditional(pr.w);
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
		prevE.input = inputRow(d, hp, s, sg, cm);                                                           // (k == 0: some valid row, unused)
		prevE.row = reinterpret_cast<const float *>(d.PE + rowOf(d, s, k > 0 ? k - 1 : 0, cm)) + 2;
		const float2 zero = make_float2(0.f, 0.f);
		A = cmulc(Pcm, lerpBand(in, lerpIndex(mp.x - tfUp), M));
		B = cmulc(Pcm, lerpBand(in, lerpIndex(mp.x - L*tfUp), M));
		const TwistParts t1 = twistParts<CH, PLAIN>(src, cm, b + 1, mp1, rotate, in, pv, prevE, tfDn, 1.0f, rot);
		const TwistParts tL = twistParts<CH, PLAIN>(src, cm, b + L, mpL, rotate, in, pv, prevE, tfDn, float(L), rot);
		Cc = twistFinish<PLAIN>(d, prevE, t1);
		Dc = twistFinish<PLAIN>(d, prevE, tL);
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
	recordChannelFields<CH>(f, p, e, mc, Pm, eMax);
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

// Hand-off words in LDS: relaxed workgroup-scope atomics.  (A `volatile` access makes the backend drain EVERY
// outstanding memory operation -- s_waitcnt vmcnt(0) -- around it, which serialised the producers' prefetch loads
// behind each poll.)  Ordering against the data they guard comes from the in-order LDS pipe plus compiler barriers.
__device__ __forceinline__ int ldsPeek(volatile int *p) { return __hip_atomic_load(const_cast<int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ldsPost(volatile int *p, int v) { __hip_atomic_store(const_cast<int *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ldsCount(volatile int *p) { (void)__hip_atomic_fetch_add(const_cast<int *>(p), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

constexpr int kVocWaves = 16; // waves of a recurrence workgroup (kVocoder, kVocoderN)

} // namespace smst
