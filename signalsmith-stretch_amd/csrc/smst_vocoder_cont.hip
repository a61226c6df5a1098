// K3 fused, mono / stereo, CONTINUOUS form: the skewed wavefront of kVocoder's line-aligned producers (smst_vocoder.hip) run through all
// tiles of a call without draining (signalsmith-stretch.h:642-660 rotation, :714-716, :722-803 main prediction + channel lock,
// makeOutput :596-603 -- the same lines kVocoder replaces, the same records, the same arithmetic in the same order: bit-identical).
//
// The tile form fills and drains the wavefront once per 64-hop tile: M + 8*63 steps of which 8*63 (14 % at 3072 bins) run with part of
// the 64 lanes idle -- and every one of those steps costs what a full one costs (the producer waves are the critical path).  Here lane r
// takes hops r, r + 64, r + 128, ... of the call one after the other: its bins form ONE virtual row
//     [hop r: bins 0 .. M-1][16 zero bins][hop r + 64: bins 0 .. M-1][16 zero bins] ...
// of period P = M/8 + 2 blocks, and the lag-8 wavefront slides over the virtual rows exactly as it slides over a tile's rows:
//   * the zero line between two hops is what the reference reads below bin 0 and above bin M-1 (:548-551), so the producers' two-line
//     buffers need no special case at a hop boundary -- the window slides from the last line of one hop over the zero line into the
//     first line of the next (P is even: a row keeps its block parity);
//   * records of the 16 gap bins are all-zero, which gives exactly zero outputs: a lane enters its next hop with a zero history, as the
//     reference's bins below 0 are (:748, :756);
//   * lane k-1 runs 8 virtual bins ahead of lane k as before (DPP wave_shr:1); lane 0's previous hop is lane 63's hop of the tile
//     before, which that lane finished P - 63 blocks before lane 0 needs the same bins: its taps are folded into lane 0's records by the
//     producer (FOLD0) from the OUT rows in memory -- written by this workgroup's own writer wave at least P - 67 blocks earlier (the
//     writer publishes how far its stores have COMPLETED, sync word NB + 3), or by the launch before.
// A launch covers the global blocks [n0, n1) = one period (the last launch of a call: until the last row has finished): the analysis of
// tile t+1 and the synthesis of tile t-1 still overlap it tile by tile.  Between two launches the recurrence wave's last eight outputs
// per lane travel through `save` (they are its history registers AND the result-ring block the writer still needs); the producers
// warm their line buffers up over the four blocks in front of n0.  Tile t is complete when launch t+1 has run.
// 3150 blocks instead of 3576 on a 500-hop call -- and, measured, the same 6.4 ms per step: a block of a tile's ramp costs in proportion to
// the producer waves that are active in it, so the tile form never paid for its idle lanes (EXPERIMENTS.md 6.1).  Opt-in: SMST_CONTINUOUS=1.
#include "smst_vocoder_common.h"

namespace smst {

struct RowInfo { float tf; unsigned flags; int inSrc, prevSrc; }; // what the kernel needs of a HopDesc, 16 bytes: [tile parity][row] in LDS

// tile-relative position of a row at a block: m = n - row.  rel: 0 = the tile before the launch's own (tile - 1), 1 = its own, 2 = the
// one after (reached only by requests that run ahead and in the last launch's extra blocks: never valid).  mm: block within the period.
struct TilePos { int rel, mm; };

// The producer waves are the kernel's critical path and they are bound by what a wave can ISSUE (EXPERIMENTS.md 4.2): the first form of
// this function located every row at every block with compares and a multiply (m = n - row -> tile, block within the period), kept its
// cursors packed, and took 440 vector instructions per block where the tile form takes 255 -- 8.3 ms per step against 6.4 with 12 % fewer
// blocks.  Now: the wave index is made scalar (everything that depends on the block number and the wave alone runs on the scalar unit), a
// lane's position is two counters that step once per block, and whatever changes only at a hop boundary sits behind a rarely taken branch.
template <int CH, int L, bool FIRST>
__device__ __forceinline__ void contProduce(const DevBatch &d, const ContArgs &a, int s, int sg, int it, int k, int n0, int n1,
                                            float4 *recs, volatile int *sync, const RowInfo *rowInfo, float2 *sbuf) {
	using G = AlignGeom<CH, L>;
	constexpr int NF = 9 + 3*CH, NCH = (NF + 3)/4, BS = 8, NB = kVocBlocksStaged;
	static_assert(2*L <= 8 && 7 + L <= 11, "the windows must fit the two lines around the row's bins");
	const int M = d.M, P = a.period, MB = M >> 3, LP = P >> 1;
	const int PQ0 = P*a.tile, PQ1 = PQ0 + P;
	// (set-up only: where row `row` stands at block n = m + row)
	auto posOf = [&](int m) { TilePos t; t.rel = (m >= PQ0 ? 1 : 0) + (m >= PQ1 ? 1 : 0); t.mm = m - PQ0 + P - P*t.rel; return t; };
	auto infoOf = [&](int rel, int row) { // rel 0 / 1: parity of tile - 1 + rel
		RowInfo r = rowInfo[(((a.tile + 1 + rel) & 1) << 6) + row];
		if (rel > 1) r.flags = 0;
		return r;
	};
	// the row of spectrum `array` (0 .. CH-1: Band.input of that channel, CH .. 2CH-1: Band.prevInput) of the hop (rel, row); d.rot for a hop that does not exist
	auto rowBase = [&](int rel, int row, int array, bool &valid) -> const float2 * {
		const RowInfo ri = infoOf(rel, row);
		valid = (ri.flags & HOP_ACTIVE) != 0;
		if (!valid) return d.rot;
		const int r2 = rel & 1;
		const int c = array < CH ? array : array - CH;
		if (array < CH) {
			if (ri.inSrc >= 0) return (r2 ? a.Xcur[1] : a.Xcur[0]) + rowOf(d, s, ri.inSrc, c);
			return d.stInput + stateRow(d, sg, c);
		}
		if (ri.prevSrc == SRC_REANALYSED) return (r2 ? a.Xprev[1] : a.Xprev[0]) + rowOf(d, s, row, c);
		if (ri.prevSrc >= 0) return (r2 ? a.Xcur[1] : a.Xcur[0]) + rowOf(d, s, ri.prevSrc, c);
		// SRC_STATE: Band.prevInput as the hop before left it.  Every hop of a continuous run analyses a new spectrum (the engine's condition),
		// so that is the input of the tile before's last row -- or, in the run's first tile, the carried state
		if (a.tile - 1 + rel == 0) return d.stPrev + stateRow(d, sg, c);
		if (rel == 1) return a.Xcur[0] + rowOf(d, s, kTileHops - 1, c);
		valid = false; // (tile - 2: only the first row of tile - 1 could ask, and it has finished before this launch)
		return d.rot;
	};
	float2 *xbuf = sbuf + 8*G::ROWLEN;
	for (int i = k; i < G::PER_PRODUCER/2; i += 64) reinterpret_cast<float4 *>(sbuf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const int nS = n0 - 4; // warm-up: two parks per row fill its two-line buffer (no records, no hand-off words)
	// ---- this lane's line pieces: [parity of the rows][load].  Rows of parity `par` take a new line in the blocks n with n + 1 = par (mod 2)
	// (their m = n - row is odd there).  Per piece: the cursor of the NEXT request (lptr; line lline of its hop), whether that hop exists,
	// and whether the request issued last was for a line that exists (read by the park that follows it).
	const float2 *lptr[2][G::LOADS];
	int llds[2][G::LOADS], linc[2][G::LOADS]; // linc: 16 bins per request while the hop exists, 0 (the cursor rests on the rotation table) while it does not
	bool lreq[2][G::LOADS];
	auto pieceLocalRow = [&](int par, int i) { return 2*(((k + 64*i) >> 3)/(2*CH)) + par; };
	auto pieceArray = [&](int i) { return ((k + 64*i) >> 3)%(2*CH); };
#pragma unroll
	for (int par = 0; par < 2; ++par) {
#pragma unroll
		for (int i = 0; i < G::LOADS; ++i) {
			const int prow = 8*it + pieceLocalRow(par, i);
			llds[par][i] = pieceLocalRow(par, i)*G::ROWLEN + pieceArray(i)*G::RING + 2*(k & 7);
			const int nF = nS + 1 - par;           // the first block at or after nS in which this parity parks
			const TilePos t = posOf(nF - prow);    // mm odd
			int line = (t.mm + 1) >> 1, rel = t.rel;
			if (line == LP) { line = 0; ++rel; }   // the last gap block brings line 0 of the next hop
			bool valid;
			const float2 *base = rowBase(rel, prow, pieceArray(i), valid);
			lptr[par][i] = base + 2*(k & 7) + (valid ? 16*line : 0);
			linc[par][i] = valid ? 16 : 0;
			lreq[par][i] = false;
		}
	}
	const int st = k & 7, r = k >> 3, row = 8*it + r;
	const int xc = (k >> 3) < CH ? (k >> 3) : 0, xpiece = k & 7;
	const bool xlane = k < 8*CH;
	// the hop above the wave's first row, by the tile that row is in (rel 0 / 1): the row before it in the same tile -- FIRST: the tile
	// before's last row, or above the run's very first hop the carried Prediction.energy (one source per launch: row 0 works in tile `a.tile`)
	const float2 *xsrcRel[2] = {d.rot, d.rot};
	int xkindRel[2] = {0, 0}; // 0 nothing (zeros), 1 carried energies, 2 a hop's input
	if (FIRST) {
		if (a.tile == 0) { xsrcRel[1] = reinterpret_cast<const float2 *>(d.stEnergy + stateRow(d, sg, xc)); xkindRel[1] = 1; }
		else { xsrcRel[1] = a.Xcur[0] + rowOf(d, s, kTileHops - 1, xc); xkindRel[1] = 2; } // (the tile before is full: a later one exists)
	} else {
#pragma unroll
		for (int rel = 0; rel < 2; ++rel) {
			bool hv;
			const float2 *base = rowBase(rel, 8*it - 1, xc, hv);
			if (hv) { xsrcRel[rel] = base; xkindRel[rel] = 2; }
		}
	}
	// FIRST: the taps of row 0's records (FOLD0) -- the carried Band.output in the run's first tile, the tile before's last row afterwards
	const float2 *tapBase[CH];
#pragma unroll
	for (int c = 0; c < CH; ++c) tapBase[c] = (a.tile == 0) ? static_cast<const float2 *>(carriedOutput(d, sg).base) + (size_t)c*M : a.OUT[0] + rowOf(d, s, kTileHops - 1, c);
	Async16 vE[G::LOADS], vO[G::LOADS], xv;
	Async8 rotNext1, rotNextL, carNext1[CH], carNextL[CH];
	float2 rot1 = make_float2(1.f, 0.f), rotL = rot1;
	float2 car1[CH], carL[CH];
#pragma unroll
	for (int c = 0; c < CH; ++c) car1[c] = carL[c] = make_float2(0.f, 0.f);
	// ---- positions, stepped once per block.  Scalar: the wave's first row (mmA, relA).  Per lane: the lane's own row (mmL, relL).
	int mmA, relA, mmL, relL;
	{ const TilePos t = posOf(nS - 8*it); mmA = t.mm; relA = t.rel; }
	{ const TilePos t = posOf(nS - row); mmL = t.mm; relL = t.rel; }
	auto hopWords = [&]() { return reinterpret_cast<const float2 *>(rowInfo)[2*((((a.tile + 1 + relL) & 1) << 6) + row)]; };
	float2 tfFlags = hopWords();
	int xKind = 0;     // of the piece staged for the block at hand
	int xKindNext = 0; // ... and for the one after it (set by issueSmall)

	// Block nPark = n + 2's lines of parity par (n: the block the counters stand at): request at the cursor, move it on by one line.  Only
	// in the eight blocks per period in which some row of the wave asks for the zero line between two hops is there anything to decide:
	// that request brings nothing and the cursor jumps to the next hop's first line (a wave-uniform branch around it all).
	auto issueLines = [&](int par, Async16 (&v)[G::LOADS]) {
		// A row with local index pr stands at block mmA - pr of its period (mod P; in the tile before the wave's first row's if that is
		// negative) and asks for the line of the block two on: the zero line when that block is MB - 1 = P - 3, i.e. mmA - pr = -5 (mod P)
		const bool window = mmA >= P - 5 || mmA <= 2; // scalar
#pragma unroll
		for (int i = 0; i < G::LOADS; ++i) {
			bool zeroLine = false;
			int dRow = 0;
			if (window) { dRow = mmA - pieceLocalRow(par, i); zeroLine = dRow == P - 5 || dRow == -5; }
			asyncLoad16(v[i], zeroLine ? d.rot : lptr[par][i]);
			lreq[par][i] = linc[par][i] != 0 && !zeroLine;
			lptr[par][i] += linc[par][i];
			if (window && zeroLine) { // the piece's next request (two blocks on) is line 0 of the hop after
				const int prow = 8*it + pieceLocalRow(par, i);
				bool hv;
				const float2 *base = rowBase((dRow >= 0 ? relA : relA - 1) + 1, prow, pieceArray(i), hv);
				lptr[par][i] = base + 2*(k & 7);
				linc[par][i] = hv ? 16 : 0;
			}
		}
	};
	// the small loads of the block AFTER the one the counters stand at: the 16 bins of the hop above the wave's first row, the rotation
	// factors of the lane's two previous-hop bins, FIRST: the taps of row 0.  Request counts do not depend on the data (the waits count requests).
	auto issueSmall = [&](int nn) {
		const int mmA1 = (mmA + 1 == P) ? 0 : mmA + 1, relA1 = relA + (mmA + 1 == P ? 1 : 0); // scalar
		const int x0 = BS*mmA1 + 2*xpiece, xcl = min(x0, M - 2);
		const int kind = relA1 <= 1 ? (relA1 ? xkindRel[1] : xkindRel[0]) : 0;
		const float2 *xsrc = relA1 ? xsrcRel[1] : xsrcRel[0];
		xKindNext = kind;
		// (FIRST, carried energies: one 16-byte request brings four floats, the first two are the piece's -- stEnergy has the slack)
		const float2 *xaddr = (kind == 2) ? xsrc + xcl : ((kind == 1) ? reinterpret_cast<const float2 *>(reinterpret_cast<const float *>(xsrc) + xcl) : d.rot);
		asyncLoad16(xv, xaddr);
		const int mmL1 = (mmL + 1 == P) ? 0 : mmL + 1;
		const int b = BS*mmL1 + st;
		asyncLoad8(rotNext1, d.rot + min(b + 1, M - 1));
		asyncLoad8(rotNextL, d.rot + min(b + L, M - 1));
		if (FIRST) { // row 0 (lanes 0..7; the other lanes request the same in-range values and never use them): its position is the wave's (mmA1, relA1)
			const int b0 = min(BS*mmA1 + st, M - 1);
			if (a.tile > 0 && relA1 == 1) {
				// The rows of tile - 1 are written by this workgroup's writer wave or by the launch before.  Row 0 at block nn reads bins up to
				// 8 mm + 11 of lane 63's hop, which that lane produced by block nn - P + 65 and the writer stored with the line's completion one
				// block later: the stores through block nn - P + 66 must have COMPLETED (the writer publishes that every 16 blocks and is never
				// more than a few blocks behind the recurrence: with P >= 128 this never waits for anything still to come).  A launch's first
				// look happens in its warm-up (n0 - 4 + 4 is a multiple of 16 only by chance): `flushed` starts at n0, which covers every
				// request before block n0 + P - 67
				if ((nn & 15) == 0) { // (the word moves every 16 blocks, three blocks behind the writer: cover the requests up to the next look)
					while (ldsPeek(&sync[NB + 3]) < nn + 15 - (P - 67)) __builtin_amdgcn_s_sleep(2);
					asm volatile("" ::: "memory");
				}
			}
#pragma unroll
			for (int c = 0; c < CH; ++c) { // (outside its own tile row 0 is in gap bins: whatever these bring is never used)
				asyncLoad8(carNext1[c], tapBase[c] + min(b0 + 1, M - 1));
				asyncLoad8(carNextL[c], tapBase[c] + min(b0 + L, M - 1));
			}
		}
	};
	auto landed = [&](Async16 (&v)[G::LOADS]) {
		asyncWait<G::LOADS>();
#pragma unroll
		for (int i = 0; i < G::LOADS; ++i) asyncArrived(v[i]);
		asyncArrived(xv);
		asyncArrived(rotNext1);
		asyncArrived(rotNextL);
		if (FIRST) {
#pragma unroll
			for (int c = 0; c < CH; ++c) { asyncArrived(carNext1[c]); asyncArrived(carNextL[c]); }
		}
	};
	auto park = [&](int par, Async16 (&v)[G::LOADS]) {
		if (FIRST) {
#pragma unroll
			for (int c = 0; c < CH; ++c) { car1[c] = asyncValue(carNext1[c]); carL[c] = asyncValue(carNextL[c]); }
		}
#pragma unroll
		for (int i = 0; i < G::LOADS; ++i) { // upper line -> lower half, the new line -> upper half (this lane's piece of both)
			float4 *lower = reinterpret_cast<float4 *>(sbuf + llds[par][i]), *upper = lower + 8;
			*lower = *upper;
			*upper = lreq[par][i] ? asyncValue(v[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
		}
		if (xlane) {
			const int x0 = BS*mmA + 2*xpiece;
			float4 piece = asyncValue(xv);
			if (FIRST && xKind == 1) piece = make_float4(piece.x, 0.f, piece.y, 0.f); // two carried energies, staged as (E, 0) pairs
			*reinterpret_cast<float4 *>(xbuf + xc*16 + 2*xpiece) = (xKind != 0 && x0 + 1 < M) ? piece : make_float4(0.f, 0.f, 0.f, 0.f);
		}
		rot1 = asyncValue(rotNext1);
		rotL = asyncValue(rotNextL);
	};
	auto step = [&](int n, int par, Async16 (&v)[G::LOADS], Async16 (&vNextBlock)[G::LOADS]) {
		xKind = xKindNext;
		park(par, v);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		issueSmall(n + 1);
		issueLines(par, v);
		if (n >= n0) {
			const int slot = n%NB;
			while (n - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2); // slot still being read
			asm volatile("" ::: "memory");
			const unsigned flags = relL <= 1 ? unsigned(__float_as_int(tfFlags.y)) : 0u;
			const int b = BS*mmL + st;
			float f[NCH*4];
#pragma unroll
			for (int j = 0; j < NCH*4; ++j) f[j] = 0.0f;
			if ((flags & HOP_ACTIVE) && mmL < MB) {
				// the arithmetic of vocoderProduceAligned (= computeRecord<CH, true, false, false>), operands from the line buffers: this row's
				// buffer holds the lines (j-1, j) in its even blocks (b0 = 16j) and (j, j+1) in its odd ones; the row above runs 8 bins ahead
				const bool rotate = flags & HOP_NEW_SPECTRUM;
				const float tf = tfFlags.x;
				const bool odd = mmL & 1;
				const float2 *mine = sbuf + r*G::ROWLEN + (odd ? 8 : 16) + st;
				const float2 *above = (r > 0) ? sbuf + (r - 1)*G::ROWLEN + (odd ? 8 : 0) + st : xbuf + st;
				const int abovePitch = (r > 0) ? G::RING : 16;
				auto IN = [&](int c, int off) { return mine[c*G::RING + off]; };
				auto lerpIN = [&](int c, LerpIndex li) {
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
				const bool energyAbove = FIRST && xKind == 1 && r == 0; // the carried Prediction.energy only above the run's very first hop
				auto twist = [&](int off, float2 rotV, float stepMul) {
					const int bc = min(b + off, M - 1);
					const float2 rotB = rotate ? rotV : make_float2(1.f, 0.f);
					const float2 Q = cmul(mine[(CH + mc)*G::RING + off], rotB);
					const float2 Px = IN(mc, off);
					const float2 TW = cmul(rotB, cmulc(Px, Q));
					const float eNow = cnorm(Px);
					const float2 up = above[mc*abovePitch + off];
					const float ePrev = energyAbove ? up.x : cnorm(up);
					const float den = fmaxf(ePrev, eNow) + 1e-15f;
					const float2 down = cmulc(Px, lerpIN(mc, lerpIndex(float(bc) - stepMul*tf)));
					const float2 rr = cmulc(TW, down);
					const float inv = __builtin_amdgcn_rcpf(den);
					return make_float2(rr.x*inv, rr.y*inv);
				};
				float2 Cc = twist(1, rot1, 1.0f), Dc = twist(L, rotL, float(L));
				const float2 zero = make_float2(0.f, 0.f);
				if (!(b > 0)) A = zero;
				if (!(b >= L)) B = zero;
				if (!(b < M - 1)) Cc = zero;
				if (!(b < M - L)) Dc = zero;
				if (FIRST) { // FOLD0: row 0's record carries the previous-hop part ready-made
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
		}
		// on to the next block
		if (++mmA == P) { mmA = 0; ++relA; }
		{ const bool wrap = mmL + 1 == P; mmL = wrap ? 0 : mmL + 1; relL += wrap ? 1 : 0; }
		tfFlags = hopWords(); // (tf, flags) of the lane's hop in the next block: 8 bytes out of LDS, asked for a block ahead of their use
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier(); // every lane has read its operands before the next block's lines are parked
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		landed(vNextBlock);
	};
	// nS is even: its parking rows have parity 1 (register set "O"), block nS + 1 the others.  issueSmall(n) is called with the counters
	// standing at block n - 1
	{ // the small loads of block nS itself: step the counters back by one block for the call
		const int keepA = mmA, keepRA = relA, keepL = mmL, keepRL = relL;
		if (mmA == 0) { mmA = P - 1; --relA; } else --mmA;
		if (mmL == 0) { mmL = P - 1; --relL; } else --mmL;
		issueSmall(nS);
		if (mmA == 0) { mmA = P - 1; --relA; } else --mmA;
		issueLines(1, vO); // (the lines of block nS: the counters two blocks before it)
		if (++mmA == P) { mmA = 0; ++relA; }
		issueLines(0, vE); // (block nS + 1)
		mmA = keepA; relA = keepRA; mmL = keepL; relL = keepRL;
	}
	landed(vO);
	for (int n = nS; n < n1; n += 2) {
		step(n, 1, vO, vE);
		step(n + 1, 0, vE, vO);
	}
	asyncWait<0>(); // the requests that ran past the last block: nothing of this wave stays in flight behind it
}

// Twelve waves, not kVocoder's sixteen: wave 0 the recurrence, wave 4 the writer, eight producers on waves 1-3, 5-7, 9, 10 (wave w runs on
// SIMD w % 4) -- the gathering form's other six waves have no work here, and three waves per SIMD leave every wave 168 registers instead
// of 128 (the first build of this kernel spilled 17 of them to scratch: the row cursors and hop bookkeeping of the continuous form).
constexpr int kContWaves = 12;
template <int CH, int L>
__global__ __launch_bounds__(64*kContWaves) __attribute__((amdgpu_waves_per_eu(3, 3))) void kVocoderCont(DevBatch d, ContArgs a, int sBase) {
	using G = AlignGeom<CH, L>;
	constexpr int NF = 9 + 3*CH, NCH = (NF + 3)/4, BS = kVocBlockSteps, NB = kVocBlocksStaged, NP = kVocStagedProducers, OB = kVocOutBlocksAligned;
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float4 *recs = reinterpret_cast<float4 *>(smemRaw);                            // [(slot*BS + st)*NCH + j][64 lanes]
	volatile int *sync = reinterpret_cast<volatile int *>(recs + NB*BS*NCH*64);    // [0..NB) units produced per slot, [NB] blocks consumed, [NB+1] result blocks ready,
	                                                                               // [NB+2] result blocks written, [NB+3] blocks whose stores have completed -- absolute block numbers
	RowInfo *rowInfo = reinterpret_cast<RowInfo *>(const_cast<int *>(sync) + 16);  // [tile parity][64]
	float2 *outRing = reinterpret_cast<float2 *>(rowInfo + 128);                   // [OB][BS][CH][kVocOutPitch]
	float2 *lines = outRing + (size_t)OB*BS*CH*kVocOutPitch;                       // the producers' line buffers

	const int s = blockIdx.x, sg = sBase + s;
	const int M = d.M, P = a.period, ML = M >> 4;
	const int nhPrev = a.tile > 0 ? a.tileInfo[(size_t)(a.tile - 1)*a.tileStride + s] : 0;
	const int nhCur = a.tile < a.nTiles ? a.tileInfo[(size_t)a.tile*a.tileStride + s] : 0;
	if (nhPrev == 0 && nhCur == 0) return;
	const int n0 = a.n0;
	const int n1 = nhCur > 0 ? a.n1 : min(a.n1, P*a.tile + 62); // (only rows finishing tile - 1: row r is done after block P*tile - 3 + r)
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), k = threadIdx.x & 63; // (scalar: what depends on it and the block number alone runs on the scalar unit)
	const int PQ0 = P*a.tile, PQ1 = PQ0 + P;
	auto posOf = [&](int m) { TilePos t; t.rel = (m >= PQ0 ? 1 : 0) + (m >= PQ1 ? 1 : 0); t.mm = m - PQ0 + P - P*t.rel; return t; };

	if (threadIdx.x < 16) sync[threadIdx.x] = (threadIdx.x >= NB) ? n0 : 0;
	if (threadIdx.x < 128) {
		const int rel = threadIdx.x >> 6, row = threadIdx.x & 63, tile = a.tile - 1 + rel;
		RowInfo ri{1.0f, 0u, SRC_STATE, SRC_STATE};
		if (row < (rel ? nhCur : nhPrev)) {
			const HopDesc hd = d.hops[(size_t)sg*d.hopStride + a.hopBase + (size_t)tile*kTileHops + row];
			ri.tf = hd.timeFactor; ri.flags = hd.flags; ri.inSrc = hd.inSrc; ri.prevSrc = hd.prevSrc;
		}
		rowInfo[((tile & 1) << 6) + row] = ri;
	}
	__syncthreads();

	if (wave > 0) {
		if (wave == a.writerWave) {
			// ---------------- writer: whole aligned 128-byte lines of OUT, as kVocoder's (lag 8: rows of one parity complete a line per block) ----------------
			const int g8 = k & 7, part = k >> 3;
			// which of this lane's rows hold a hop, per tile of the launch: bit rel*8 + pass*2 + par for row 2*(8*pass + g8) + par
			unsigned active = 0;
#pragma unroll
			for (int rel = 0; rel < 2; ++rel) {
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const int row = 2*(8*(j >> 1) + g8) + (j & 1);
					if (rowInfo[(((a.tile + 1 + rel) & 1) << 6) + row].flags & HOP_ACTIVE) active |= 1u << (rel*8 + j);
				}
			}
			float2 *const outS[2] = {a.OUT[0] + rowOf(d, s, 0, 0), a.OUT[1] + rowOf(d, s, 0, 0)}; // the stream's first row in both workspaces (scalar)
			const int Mp = d.Mp;
			int mm0 = posOf(n0).mm, rel0 = posOf(n0).rel; // row 0's position (scalar), stepped once per block
			const int i0 = (2*part) & 7;
			for (int n = n0; n < n1; ++n) {
				while (ldsPeek(&sync[NB + 1]) <= n) __builtin_amdgcn_s_sleep(2);
				asm volatile("" ::: "memory");
				const int par = (n + 1) & 1; // the rows whose m = n - row is odd
				const int rbNow = n%OB, rbBefore = (n + OB - 1)%OB; // (scalar; n0 - 1 >= -1)
				const int rb = (part < 4) ? rbBefore : rbNow;      // bins 16G .. 16G+7 came with the block before
#pragma unroll
				for (int pass = 0; pass < 4; ++pass) {
					const int row = 2*(8*pass + g8) + par;
					int mm = mm0 - row, rel = rel0;
					if (mm < 0) { mm += P; --rel; }
					const int G16 = (mm - 1) >> 1;                   // the line whose upper half this block produced
					const bool ok = rel >= 0 && rel <= 1 && ((active >> (rel*8 + pass*2 + par)) & 1u) && G16 < ML;
					const int b = 16*G16 + 2*part;
#pragma unroll
					for (int c = 0; c < CH; ++c) {
						const float2 v0 = outRing[((rb*BS + i0)*CH + c)*kVocOutPitch + row], v1 = outRing[((rb*BS + i0 + 1)*CH + c)*kVocOutPitch + row];
						if (ok) {
							float2 *dst = (rel ? outS[1] : outS[0]) + ((row*CH + c)*Mp + b); // (a row's offset within its stream: 32 bits)
							dst[0] = v0;
							dst[1] = v1;
						}
					}
				}
				asm volatile("" ::: "memory");
				// What the producer of row 0 needs before it reads these rows back: how far the stores have COMPLETED.  This wave issues nothing
				// but stores to memory (at most 4 passes x CH x 2 per block) and stores complete in order, so "all but the youngest 3 blocks'
				// worth" is a wait by count that never stalls in practice.  (The first form drained them all -- s_waitcnt vmcnt(0) -- every 16
				// blocks: a few microseconds in which the recurrence ran into the writer's two-block ring, +11 % on the whole kernel.)
				asyncWait<3*4*CH*2>();
				if (k == 0) { ldsPost(&sync[NB + 2], n + 1); if ((n & 15) == 15) ldsPost(&sync[NB + 3], n - 2); }
				if (++mm0 == P) { mm0 = 0; ++rel0; }
			}
			return;
		}
		// the producers' placement of the tile form: 8 waves on three SIMDs (1,5,9 / 2,6,10 / 3,7), the recurrence wave's SIMD left to it and
		// the writer; the wave with rows 0..7 (carried taps, FOLD0: the heaviest) on the SIMD that holds two producers
		int pIndex = wave - 1 - (wave > 4);
		pIndex = ((wave & 3) && wave != 11) ? ((wave < 8) ? pIndex : ((wave == 9) ? 6 : ((wave == 10) ? 7 : NP))) : NP;
		if (pIndex < NP) pIndex = (pIndex == 0) ? 2 : ((pIndex == 2) ? 0 : pIndex);
		if (pIndex >= NP) return;
		float2 *sbuf = lines + (size_t)pIndex*G::PER_PRODUCER;
		if (pIndex == 0) contProduce<CH, L, true>(d, a, s, sg, pIndex, k, n0, n1, recs, sync, rowInfo, sbuf);
		else contProduce<CH, L, false>(d, a, s, sg, pIndex, k, n0, n1, recs, sync, rowInfo, sbuf);
		return;
	}

	// ---------------- consumer (wave 0): kVocoder's, over the global blocks [n0, n1) ----------------
	__builtin_amdgcn_s_setprio(3);
	float2 h[8][CH];
	float2 tap1[CH], tapL[CH];
	float2 *mySave = a.save + ((size_t)sg*BS*CH)*64 + k;
#pragma unroll
	for (int c = 0; c < CH; ++c) {
		tap1[c] = make_float2(k == 0 ? 1.f : 0.f, 0.f); // lane 0: the constant taps of FOLD0 records; the others: overwritten by every DPP move
		tapL[c] = make_float2(0.f, 0.f);
	}
	{
		// the history of the launch before: this lane's last eight outputs -- which are also the result-ring block n0 - 1, of which the
		// writer still needs the rows that had completed half a line
		float2 *ringBefore = outRing + (size_t)(((n0 - 1)%OB + OB)%OB)*BS*CH*kVocOutPitch + k;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
#pragma unroll
			for (int c = 0; c < CH; ++c) {
				h[i][c] = (a.tile > 0) ? mySave[(size_t)(i*CH + c)*64] : make_float2(0.f, 0.f);
				ringBefore[(i*CH + c)*kVocOutPitch] = h[i][c];
			}
		}
	}
	constexpr int lag = 8;
	int seenProduced = ldsPeek(&sync[n0%NB]), seenWritten = n0;
	for (int n = n0; n < n1; ++n) {
		const int slot = n%NB;
		const int need = 8*((n - n0)/NB + 1);
		while (seenProduced < need) { __builtin_amdgcn_s_sleep(1); seenProduced = ldsPeek(&sync[slot]); }
		asm volatile("" ::: "memory");
		const float4 *blockRecs = recs + (size_t)slot*BS*NCH*64;
		while (n - seenWritten >= 2) { __builtin_amdgcn_s_sleep(1); seenWritten = ldsPeek(&sync[NB + 2]); } // the writer still owns this result slot
		asm volatile("" ::: "memory");
		float2 *blockOut = outRing + (size_t)(n%OB)*BS*CH*kVocOutPitch + k;
		float4 q[2][NCH];
#pragma unroll
		for (int j = 0; j < NCH; ++j) q[0][j] = blockRecs[j*64 + k];
#pragma unroll
		for (int i = 0; i < BS; ++i) {
			if (i + 1 < BS) {
#pragma unroll
				for (int j = 0; j < NCH; ++j) q[(i + 1) & 1][j] = blockRecs[((i + 1)*NCH + j)*64 + ((k + (i + 1)) & 63)];
			} else {
				seenProduced = ldsPeek(&sync[(n + 1)%NB]);
				seenWritten = ldsPeek(&sync[NB + 2]);
			}
			float f[NCH*4];
#pragma unroll
			for (int j = 0; j < NCH; ++j) { f[4*j] = q[i & 1][j].x; f[4*j + 1] = q[i & 1][j].y; f[4*j + 2] = q[i & 1][j].z; f[4*j + 3] = q[i & 1][j].w; }
			const int mc = __float_as_int(f[8]);
#pragma unroll
			for (int c = 0; c < CH; ++c) {
				tap1[c] = fromLaneBelow(h[(i + 17 - lag) & 7][c], tap1[c]);
				tapL[c] = fromLaneBelow(h[(i + 16 + L - lag) & 7][c], tapL[c]);
			}
			float2 o1 = h[(i + 7) & 7][0], oL = h[(i + 8 - L) & 7][0], p1 = tap1[0], pL = tapL[0];
#pragma unroll
			for (int c = 1; c < CH; ++c) {
				const bool pick = c == mc;
				o1 = selectPair(pick, h[(i + 7) & 7][c], o1);
				oL = selectPair(pick, h[(i + 8 - L) & 7][c], oL);
				p1 = selectPair(pick, tap1[c], p1);
				pL = selectPair(pick, tapL[c], pL);
			}
			const float2 pm = make_float2(f[9], f[10]);
			const float sm = f[11];
			float2 phi = prevHopTerms(p1, make_float2(f[4], f[5]), pL, make_float2(f[6], f[7]));
			phi = cfma(oL, make_float2(f[2], f[3]), phi);
			phi = cfma(o1, make_float2(f[0], f[1]), phi);
			const float2 om = (CH == 2) ? makeOutputFb(phi, pm, sm) : makeOutput(phi, pm, sm);
			if (CH == 2) {
				const float2 olock = lockedOutput(om, f);
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
		if (k == 0) { ldsPost(&sync[NB], n + 1); ldsPost(&sync[NB + 1], n + 1); }
	}
#pragma unroll
	for (int i = 0; i < 8; ++i) {
#pragma unroll
		for (int c = 0; c < CH; ++c) mySave[(size_t)(i*CH + c)*64] = h[i][c];
	}
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
bool continuousSupported(const DevBatch &d) {
	// (M >= 1024: a period of at least 128 blocks -- a launch then sees two tiles per row at most, and row 0 reads lane 63's rows long after they were stored)
	return d.C <= 2 && d.L >= 2 && d.L <= 4 && d.lag == d.L + 1 && d.M%16 == 0 && d.M >= 1024 && !d.halfState && !d.noStage && !d.noAlign;
}
int continuousPeriod(const DevBatch &d) { return d.M/8 + 2; }

template <int CH, int L>
static void launchContL(const DevBatch &d, const ContArgs &a, int sBase, int nStreams, hipStream_t st) {
	using G = AlignGeom<CH, L>;
	constexpr int NCH = (9 + 3*CH + 3)/4;
	const size_t lds = (size_t)kVocBlocksStaged*kVocBlockSteps*NCH*64*sizeof(float4) + 64 + 128*sizeof(RowInfo)
	                   + (size_t)kVocOutBlocksAligned*kVocBlockSteps*CH*kVocOutPitch*sizeof(float2) + (size_t)kVocStagedProducers*G::PER_PRODUCER*sizeof(float2);
	hipLaunchKernelGGL((kVocoderCont<CH, L>), dim3(nStreams), dim3(64*kContWaves), lds, st, d, a, sBase);
}
template <int CH>
static void launchContT(const DevBatch &d, const ContArgs &a, int sBase, int nStreams, hipStream_t st) {
	switch (d.L) {
	case 2: launchContL<CH, 2>(d, a, sBase, nStreams, st); break;
	case 3: launchContL<CH, 3>(d, a, sBase, nStreams, st); break;
	default: launchContL<CH, 4>(d, a, sBase, nStreams, st); break;
	}
}
void launchVocoderContinuous(const DevBatch &d, const ContArgs &a, int sBase, int nStreams, hipStream_t st) {
	countLaunch(LK_VOC_CONT);
	if (d.C == 1) launchContT<1>(d, a, sBase, nStreams, st);
	else launchContT<2>(d, a, sBase, nStreams, st);
}

} // namespace smst
