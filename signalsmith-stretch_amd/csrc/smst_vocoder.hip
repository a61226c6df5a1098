// K3 fused, mono / stereo: the bin recurrence and its record producers in one kernel (kVocoder; ACROSS form for single-hop tiles) and its launchers.
#include "smst_vocoder_common.h"

namespace smst {

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
	// (The wave index as a scalar -- what depends on the wave and the block number alone then runs on the scalar unit -- for the line-aligned
	// producers only: 251 -> 244 vector instructions per block there.  The staged producers' loop came out SLOWER with it -- presetCheaper
	// 65.5 -> 55.8 Gsamples/s, presetDefault at 44.1 kHz 38.2 -> 33.1, profiles/r6_presets_scalar_wave_regression.json: found by the preset table)
	const int wave = ALIGNED ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : int(threadIdx.x >> 6), k = threadIdx.x & 63;
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
		if (!ACROSS && d.vocWide) {
			// WIDE passes (round 6): 4 rows x 16 steps -- two blocks of the three-block ring at once.  The gathering producers are what a mapped
			// tile's recurrence waits for (configs 3 / 4b: the recurrence wave alone needs 13 of config 3's 30 ms, the producers 29, of which
			// 10 are their requests to the L1: EXPERIMENTS.md 6.9), and a request is one (instruction, line): with 16 consecutive bins of
			// a row in one load a line is asked for once where two passes of 8 bins asked twice.  Same records, same slots.
			static_assert(NB >= 2, "a wide pass fills two blocks of the ring");
			const int st16 = k & 15, r4 = k >> 4;
			const int half = st16 >> 3, st = st16 & 7;
			for (int u = pIndex; u < (totalBlocks/2)*16; u += NP) { // (totalBlocks is a multiple of 8)
				const int pair = u >> 4, it = u & 15;
				const int row = 4*it + r4;
				const int b = 2*BS*pair + st16 - lag*row;
				float f[NCH*4];
#pragma unroll
				for (int j = 0; j < NCH*4; ++j) f[j] = 0.0f;
				if (row < nh && b >= 0 && b < M && !SMST_SKIP_PRODUCER_MATH(d))
					computeRecord<CH, PLAIN, false, false, NCH*4, ROTL, true>(d, hopsLds[row], hopsLds[row > 0 ? row - 1 : 0], s, sg, row, b, f, rotLds);
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					const int n = 2*pair + h, slot = n%NB;
					while (n - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2); // slot still being read
					asm volatile("" ::: "memory");
					if (half == h) {
#pragma unroll
						for (int j = 0; j < NCH; ++j) recs[((slot*BS + st)*NCH + j)*64 + ((row + st) & 63)] = make_float4(f[4*j], f[4*j + 1], f[4*j + 2], f[4*j + 3]);
					}
					asm volatile("" ::: "memory");
					if (k == 0) ldsCount(&sync[slot]); // LDS ops of a wave are in order: data first, then the count
				}
			}
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
	const int units = (!STAGED && !ACROSS && d.vocWide) ? 16 : 8; // producer passes that make up a block
	int seenProduced = ldsPeek(&sync[0]), seenWritten = 0;
	for (int n = 0; n < totalBlocks; ++n) {
		const int slot = n%NB;
		const int need = units*(n/NB + 1);
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
// host-side launchers
// ------------------------------------------------------------------------------------------------------
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
bool fusedSupported(const DevBatch &d) {
	return d.C <= kMaxFusedChannels && d.lag == d.L + 1 && d.L >= 2 && d.L <= (d.C <= 2 ? 7 : 5); // other geometries / more channels: kPredictB + kChain (records through HBM)
}
void launchVocoder(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, bool bounded, hipStream_t st) {
	switch (d.C) {
	case 1: launchVocoderT<1>(d, sBase, nStreams, hopBase, plain, bounded, st); return;
	case 2: launchVocoderT<2>(d, sBase, nStreams, hopBase, plain, bounded, st); return;
	default: launchVocoderMany(d, sBase, nStreams, hopBase, plain, st); return; // 3-8 channels: kVocoderN
	}
}

} // namespace smst
