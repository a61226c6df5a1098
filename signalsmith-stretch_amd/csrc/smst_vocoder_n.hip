// K3 for 3-8 channels (kVocoderN), for single-hop tiles (kVocoderOne), and the un-fused form with its record kernels (kPredictA / kPredictB + kChain),
// with their launchers.
#include "smst_recurrence.h"

namespace smst {

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

// One workgroup = 8 wavefront steps x all 64 hops of a stream.  Reads are coalesced along the bin index (8 lanes
// per row); the 512 records are transposed through LDS so that the stores to the skewed array are contiguous 1-KiB
// rows (the scattered 16-byte stores of the first version ran at 1.3 TB/s and dominated the whole pipeline).
// (Used for more than 2 channels; mono/stereo use the fused kVocoder below, which never writes records to HBM.)
template <int CH, bool PLAIN>
__global__ __launch_bounds__(256) void kPredictB(DevBatch d, int sBase, int hopBase) {
	constexpr int NF = recordFloats(CH), NCH = (NF + 3)/4;
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

template <int CH>
__global__ __launch_bounds__(64) void kChain(DevBatch d, int sBase, int hopBase) {
	constexpr int NF = recordFloats(CH), NCH = (NF + 3)/4;
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
				const unsigned word = unsigned(__float_as_int(f[8])); // the maximum channel; 3+ channels: bits 8.. flag the channels whose lock falls back to their input
				int mc = int(word & 255u);
				mc = (mc > CH - 1) ? CH - 1 : mc; // records of out-of-range steps are not initialised
				float2 o1 = own1[0];
				const float2 pm = make_float2(f[9], f[10]); // mono: the input; 2+ channels: the maximum channel's fallback output (recordChannelFields)
				const float sm = f[11];
#pragma unroll
				for (int c = 1; c < CH; ++c) {
					if (c == mc) o1 = own1[c];
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
				const float2 om = (CH >= 2) ? makeOutputFb(phi, pm, sm) : makeOutput(phi, pm, sm); // :788 (records of 2+ channels carry the fallback output in pm's place)
				const float2 olock = (CH == 2) ? lockedOutput(om, f) : om; // stereo: see recordChannelFields
#pragma unroll
				for (int c = 0; c < CH; ++c) {
					float2 oc;
					if constexpr (CH == 2) oc = olock;
					else if constexpr (CH == 1) oc = om;
					else oc = lockedOutputN(om, f, c, word); // channel lock, :791-800, pre-scaled by the record's producer
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
// K3 fused, 3-8 channels (signalsmith-stretch.h:722-803 for any channel count).  Same organisation as kVocoder -- producer
// waves compute the records into an LDS ring, wave 0 runs the skewed wavefront, wave 4 drains the results with
// row-coalesced stores -- with three differences that the channel count forces:
//   * a record is 12 + 2*CH floats (28 for 8 channels; 9 + 3*CH = 33 until round 6), so a block is 4 steps instead of 8 (2 x 4 x 7 KiB of LDS);
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
	constexpr int NF = recordFloats(CH), NCH = (NF + 3)/4, BS = kVocNBlockSteps, NB = kVocNBlocks, R = kVocNRing, Rm = R - 1;
	constexpr int NP = kVocWaves - 2;
	constexpr int lag = L + 1;
	static_assert(CH >= 3 && CH <= kMaxFusedChannels && L >= 1 && L + BS < R, "ring depth: a slot is rewritten R bins later, the oldest tap is L bins back");
	static_assert(NB == 2, "the wide producer passes fill both blocks of the ring");
	extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
	float4 *recs = reinterpret_cast<float4 *>(smemRaw);                  // [(slot*BS + st)*NCH + j][64 lanes]
	float2 *ring = reinterpret_cast<float2 *>(recs + NB*BS*NCH*64);      // [CH][R bins][64 lanes]: Band.output of the last R bins of every hop
	float2 *stage = ring + CH*R*64;                                      // [CH][128]: carried Band.output, 128-bin window
	volatile int *sync = reinterpret_cast<volatile int *>(stage + CH*128); // [0..NB) units produced, [NB] blocks consumed, [NB+1] result blocks ready, [NB+2] written
	HopDesc *hopsLds = reinterpret_cast<HopDesc *>(const_cast<int *>(sync) + 16);
	int *rowClass = reinterpret_cast<int *>(hopsLds + 64);                 // [4][64]: the writer's classes of rows (two for half lines, four for whole lines), then its slabs

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
			if (d.vocNHalfLines == 2 && (M & 15) == 0) {
				// WHOLE LINES (round 6, second step): a row's 16 bins = one 128-byte line per request, eight lanes per row, 8 rows per store
				// instruction.  A row completes an aligned line every fourth block (4-bin group index n - ceil(lag*row/4) = 3 mod 4): four classes
				// of 16 rows, one of which stores per block; the other three copy their fresh group out of the ring into a slab of LDS (the ring
				// is only 16 bins deep).  Six slabs of [16 rows][CH][4 bins] do: a line's first group waits three passes (three slabs in rotation),
				// the second two (two), the third one -- and a pass reads the slabs it frees before it fills them (LDS operations of a wave
				// execute in order).  24 KB at 8 channels, which the shorter records of round 6 left free.
				float4 *slabs = reinterpret_cast<float4 *>(rowClass + 4*64);
				constexpr int SL = 16*CH*2; // float4s per slab
				{
					int c0 = 0, c1 = 0, c2 = 0, c3 = 0; // (scalars: a counter array indexed by the class would live in scratch memory)
					for (int r = 0; r < 64; ++r) { // every lane walks the same list; lane 0 records it
						const int q = ((lag*r + 3) >> 2) & 3;
						const int at = (q == 0) ? c0 : ((q == 1) ? c1 : ((q == 2) ? c2 : c3));
						if (k == 0) rowClass[q*64 + at] = r; // 16 rows per class for every lag from 2 to 7
						c0 += q == 0; c1 += q == 1; c2 += q == 2; c3 += q == 3;
					}
				}
				static_assert(lag >= 2 && lag <= 7, "the whole-line writer's four classes of 16 rows");
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				for (int n = 0; n <= totalBlocks; ++n) {
					if (n < totalBlocks) {
						while (ldsPeek(&sync[NB + 1]) <= n) __builtin_amdgcn_s_sleep(2);
					}
					asm volatile("" ::: "memory");
					const int base0 = (n%3)*SL, base1 = (3 + (n & 1))*SL, base2 = 5*SL; // where a group 0 / 1 / 2 filed in THIS pass goes -- and where the line stored in this pass finds its own
					{
						const int oct = k & 7, rho = k >> 3, g = oct >> 1, part = oct & 1;
						const int base = (g == 0) ? base0 : ((g == 1) ? base1 : base2);
						const int cls = (n + 1) & 3; // ceil(lag*row/4) = n - 3 (mod 4)
#pragma unroll
						for (int j = 0; j < 2; ++j) {
							const int idx = 8*j + rho;
							const int row = rowClass[cls*64 + idx];
							const int grp = n - ((lag*row + 3) >> 2);
							const int b = 4*grp + 2*part; // lanes 6-7: this block's group, still in the ring
							const bool ok = row < nh && grp >= 3 && 4*grp < M; // (M is a multiple of 16: the line lies below M)
#pragma unroll
							for (int c = 0; c < CH; ++c) {
								float4 val = slabs[base + (idx*CH + c)*2 + part];
								if (g == 3) {
									const float2 v0 = ring[(c*R + (b & Rm))*64 + row], v1 = ring[(c*R + ((b + 1) & Rm))*64 + row];
									val = make_float4(v0.x, v0.y, v1.x, v1.y);
								}
								if (ok) *reinterpret_cast<float4 *>(d.OUT + rowOf(d, s, row, c) + 4*(grp - 3) + 2*oct) = val;
							}
						}
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier(); // every lane has read the slabs this pass frees (the hardware's lanes run in lockstep; the CPU stand-in's do not)
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					{
						const int idx = k >> 2, part = (k >> 1) & 1, odd = k & 1;
#pragma unroll
						for (int g = 0; g < 3; ++g) {
							const int row = rowClass[((n - g) & 3)*64 + idx];
							const int b = 4*(n - ((lag*row + 3) >> 2)) + 2*part;
							const int base = (g == 0) ? base0 : ((g == 1) ? base1 : base2);
#pragma unroll
							for (int cc = 0; cc < (CH + 1)/2; ++cc) {
								const int c = 2*cc + odd;
								if (c < CH) {
									const float2 v0 = ring[(c*R + (b & Rm))*64 + row], v1 = ring[(c*R + ((b + 1) & Rm))*64 + row];
									slabs[base + (idx*CH + c)*2 + part] = make_float4(v0.x, v0.y, v1.x, v1.y);
								}
							}
						}
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier(); // ... and filed its groups before the next pass reads them
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					if (k == 0) ldsPost(&sync[NB + 2], n + 1);
				}
				return;
			}
			if (d.vocNHalfLines && (M & 7) == 0) {
				// HALF LINES (round 6).  The kernel lives on the number of requests its CU's L1 takes (loads and stores alike, hits included:
				// EXPERIMENTS.md 6.9 -- the stores of 32-byte sectors cost 12 of its 42 ms per step, 7 of them for the count of lines alone), so a
				// row's 8 bins = 64 bytes go out in ONE request: four lanes per row, 16 rows per store instruction.  A row completes an aligned
				// 8-bin group every other block (when its 4-bin group index n - ceil(lag*row/4) is odd), so the rows fall into two classes that
				// store on alternate blocks; on the block in between the row's lanes 0-1 take the first four bins out of the ring into
				// registers -- the ring keeps its depth and the hand-shake with the recurrence wave its slack.
				int count[2] = {0, 0};
				for (int r = 0; r < 64; ++r) { // every lane walks the same list; lane 0 records it
					const int q = ((lag*r + 3) >> 2) & 1;
					if (k == 0) rowClass[q*64 + count[q]] = r;
					++count[q];
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				constexpr int PASSES = 2; // 32 rows per class for every lag from 2 to 7 (ceil(lag*row/4) is odd for exactly half of the 64 rows)
				static_assert(lag >= 2 && lag <= 7, "the half-line writer's two classes of rows");
				const int rho = k >> 2, quarter = k & 3;
				float4 held[PASSES][CH];
#pragma unroll
				for (int j = 0; j < PASSES; ++j) {
#pragma unroll
					for (int c = 0; c < CH; ++c) held[j][c] = make_float4(0.f, 0.f, 0.f, 0.f);
				}
				for (int n = 0; n <= totalBlocks; ++n) {
					if (n < totalBlocks) {
						while (ldsPeek(&sync[NB + 1]) <= n) __builtin_amdgcn_s_sleep(2);
					}
					asm volatile("" ::: "memory");
					const int qStore = (n + 1) & 1, qHold = n & 1; // ceil(lag*row/4) = n + 1 (mod 2): the block completed an odd group
#pragma unroll
					for (int j = 0; j < PASSES; ++j) {
						const int idx = 16*j + rho;
						{
							const bool listed = idx < count[qStore];
							const int row = rowClass[qStore*64 + (listed ? idx : 0)];
							const int grp = n - ((lag*row + 3) >> 2);
							const int b = 4*(grp - 1) + 2*quarter; // lanes 0-1: the bins held since the last block; lanes 2-3: this block's
							const bool ok = listed && row < nh && grp >= 1 && 4*grp < M;
#pragma unroll
							for (int c = 0; c < CH; ++c) {
								float4 val = held[j][c];
								if (quarter >= 2) {
									const float2 v0 = ring[(c*R + (b & Rm))*64 + row], v1 = ring[(c*R + ((b + 1) & Rm))*64 + row];
									val = make_float4(v0.x, v0.y, v1.x, v1.y);
								}
								if (ok) *reinterpret_cast<float4 *>(d.OUT + rowOf(d, s, row, c) + b) = val; // (M is a multiple of 8: the group lies below M)
							}
						}
						{
							const bool listed = idx < count[qHold];
							const int row = rowClass[qHold*64 + (listed ? idx : 0)];
							const int grp = n - ((lag*row + 3) >> 2);
							const int b = 4*grp + 2*(quarter & 1); // (lanes 2-3 keep copies nobody stores)
#pragma unroll
							for (int c = 0; c < CH; ++c) {
								const float2 v0 = ring[(c*R + (b & Rm))*64 + row], v1 = ring[(c*R + ((b + 1) & Rm))*64 + row];
								held[j][c] = make_float4(v0.x, v0.y, v1.x, v1.y);
							}
						}
					}
					asm volatile("" ::: "memory");
					if (k == 0) ldsPost(&sync[NB + 2], n + 1);
				}
				return;
			}
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
						if (ok) { // bins M .. M+2 of the last group land in the rows' padding.  ONE 16-byte store per lane: a lane pair fills its sector with one
							// instruction (two 8-byte stores per lane interleave the pair's halves: every sector was visited twice, and the writer's
							// stores share the CU's L1 request queue with the producers' gathers)
							float2 *dst = d.OUT + rowOf(d, s, row, c) + b0; // (rows are 16-byte aligned, b0 is even)
							*reinterpret_cast<float4 *>(dst) = make_float4(v0.x, v0.y, v1.x, v1.y);
						}
					}
				}
				asm volatile("" ::: "memory");
				if (k == 0) ldsPost(&sync[NB + 2], n + 1);
			}
			return;
		}
		// ---------------- producers: a wave-pass computes 64 records.  Two shapes: 16 rows x 4 steps (one block: the first form), or
		// 8 rows x 8 steps (BOTH blocks of the ring).  A record's operands are row segments around its bin -- 8 channels x (P, E), the previous
		// hop's energies, prevInput pairs, the maximum channel's vertical taps -- and the kernel lives on the rate at which a CU's L1 takes
		// in the lines they lie on (EXPERIMENTS.md 4.12: ~1300 half-used lines per 4-step block): with 8 consecutive bins of a row in ONE
		// load instruction a line is asked for once where two passes of 4 bins asked twice, a block apart.  Same records either way.
		const int pIndex = wave - 1 - (wave > 4);
		if (d.vocNWide) {
			constexpr int PS = 2*BS, PR = 64/PS, PASSES = 64/PR; // 8 steps x 8 rows; 8 passes per pair of blocks
			const int st8 = k & (PS - 1), r = k/PS;
			const int half = st8/BS, st = st8 & (BS - 1);
			for (int u = pIndex; u < (totalBlocks/2)*PASSES; u += NP) {
				const int pair = u/PASSES, it = u - pair*PASSES;
				const int row = PR*it + r;
				const int b = PS*pair + st8 - lag*row;
				float f[NCH*4];
#pragma unroll
				for (int j = 0; j < NCH*4; ++j) f[j] = 0.0f;
				if (row < nh && b >= 0 && b < M && !SMST_SKIP_PRODUCER_MATH(d)) computeRecord<CH, PLAIN, false, false>(d, hopsLds[row], hopsLds[row > 0 ? row - 1 : 0], s, sg, row, b, f);
#pragma unroll
				for (int h = 0; h < 2; ++h) { // block 2*pair + h lives in slot h (NB == 2)
					const int n = 2*pair + h;
					while (n - ldsPeek(&sync[NB]) >= NB) __builtin_amdgcn_s_sleep(2); // slot still being read
					asm volatile("" ::: "memory");
					if (half == h) {
#pragma unroll
						for (int j = 0; j < NCH; ++j) recs[((h*BS + st)*NCH + j)*64 + ((row + st) & 63)] = make_float4(f[4*j], f[4*j + 1], f[4*j + 2], f[4*j + 3]);
					}
					asm volatile("" ::: "memory");
					if (k == 0) ldsCount(&sync[h]); // LDS ops of a wave are in order: data first, then the count
				}
			}
			return;
		}
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
	const int UNITS = d.vocNWide ? 2*BS : BS; // producer passes that make up a block
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
				const unsigned word = unsigned(__float_as_int(f[8])); // the maximum channel; bits 8..: channels whose lock falls back to their own input
				int mc = int(word & 255u);
				mc = (mc > CH - 1) ? CH - 1 : mc;
				float2 o1 = own1[0];
				const float2 pm = make_float2(f[9], f[10]); // the maximum channel's fallback output (recordChannelFields)
				const float sm = f[11];
#pragma unroll
				for (int c = 1; c < CH; ++c) {
					if (c == mc) o1 = own1[c];
				}
				const int ringRow = mc*R;
				const float2 oL = ring[(ringRow + ((b - L) & Rm))*64 + k];
				const float2 p1 = (k == 0) ? stage[mc*128 + ((b + 1) & 127)] : ring[(ringRow + ((b + 1) & Rm))*64 + k - 1];
				const float2 pL = (k == 0) ? stage[mc*128 + ((b + L) & 127)] : ring[(ringRow + ((b + L) & Rm))*64 + k - 1];
				float2 phi = prevHopTerms(p1, make_float2(f[4], f[5]), pL, make_float2(f[6], f[7])); // previous hop's part first (what FOLD0 records pre-compute)
				phi = cfma(oL, make_float2(f[2], f[3]), phi);
				phi = cfma(o1, make_float2(f[0], f[1]), phi); // the newest operand last: two dependent instructions behind it
				const float2 om = makeOutputFb(phi, pm, sm); // :788
#pragma unroll
				for (int c = 0; c < CH; ++c) {
					float2 oc = lockedOutputN(om, f, c, word); // channel lock, :791-800: one complex multiply, its normalisation is the producer's (recordChannelFields)
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
	constexpr int NF = recordFloats(CH), NCH = (NF + 3)/4, BS = kVocOneBlock, NB = 2;
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
	// split computation: a flush() that fell between two chunks of this block's main prediction zeroed the bins that had been computed
	// (:458-463); the chunks that ran afterwards began on zeros (HopDesc.startBin; 0 otherwise)
	const int startBin = hopLds[0].startBin;
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
				const unsigned word = unsigned(__float_as_int(f[8])); // the maximum channel; 3+ channels: bits 8.. flag the channels whose lock falls back to their input
				int mc = int(word & 255u);
				mc = (mc > CH - 1) ? CH - 1 : mc;
				float2 o1 = h[(i + 7) & 7][0], oL = h[(i + 8 - L) & 7][0];
				float2 p1 = stage[(b + 1) & 127], pL = stage[(b + L) & 127];
				const float2 pm = make_float2(f[9], f[10]); // mono: the input; 2+ channels: the maximum channel's fallback output
				const float sm = f[11];
#pragma unroll
				for (int c = 1; c < CH; ++c) {
					const float2 p1c = stage[c*128 + ((b + 1) & 127)], pLc = stage[c*128 + ((b + L) & 127)];
					if (c == mc) { o1 = h[(i + 7) & 7][c]; oL = h[(i + 8 - L) & 7][c]; p1 = p1c; pL = pLc; }
				}
				float2 phi = prevHopTerms(p1, make_float2(f[4], f[5]), pL, make_float2(f[6], f[7])); // previous hop's part first (what FOLD0 records pre-compute)
				phi = cfma(oL, make_float2(f[2], f[3]), phi);
				phi = cfma(o1, make_float2(f[0], f[1]), phi); // the newest operand last: two dependent instructions behind it
				const float2 om = (CH >= 2) ? makeOutputFb(phi, pm, sm) : makeOutput(phi, pm, sm); // :788 (records of 2+ channels carry the fallback output in pm's place)
				const float2 olock = (CH == 2) ? lockedOutput(om, f) : om; // stereo: see recordChannelFields
#pragma unroll
				for (int c = 0; c < CH; ++c) {
					float2 oc;
					if constexpr (CH == 2) oc = olock;
					else if constexpr (CH == 1) oc = om;
					else oc = lockedOutputN(om, f, c, word); // channel lock, :791-800, pre-scaled by the record's producer
					if (c == mc) oc = om;
					if (b < startBin) oc = make_float2(0.f, 0.f);
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
// host-side launchers
// ------------------------------------------------------------------------------------------------------
template <int CH>
static void launchPredictT(const DevBatch &d, int sBase, int nStreams, int hopBase, int tileHops, bool plain, bool passADone, hipStream_t st) {
	const dim3 grid(divUp(d.M + d.lag*(d.T - 1), 8), nStreams);
	const size_t lds = (size_t)8*((recordFloats(CH) + 3)/4)*65*sizeof(float4);
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
template <int CH, int L>
static void launchVocoderNL(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st) {
	constexpr int NCH = (recordFloats(CH) + 3)/4;
	const size_t lds = (size_t)kVocNBlocks*kVocNBlockSteps*NCH*64*sizeof(float4) + (size_t)CH*kVocNRing*64*sizeof(float2)
	                   + (size_t)CH*128*sizeof(float2) + 64 + 64*sizeof(HopDesc) + 4*64*sizeof(int) + (size_t)6*16*CH*2*sizeof(float4);
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
	constexpr int NCH = (recordFloats(CH) + 3)/4;
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
bool singleHopSupported(const DevBatch &d) { return d.C <= kMaxFusedChannels && d.lag == d.L + 1 && d.L >= 2 && d.L <= 5; }
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
void launchPredict(const DevBatch &d, int sBase, int nStreams, int hopBase, int tileHops, bool plain, bool passADone, hipStream_t st) {
	switch (d.C) {
	case 1: launchPredictT<1>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 2: launchPredictT<2>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 3: launchPredictT<3>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 4: launchPredictT<4>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 5: launchPredictT<5>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 6: launchPredictT<6>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 7: launchPredictT<7>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 8: launchPredictT<8>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 9: launchPredictT<9>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 10: launchPredictT<10>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 11: launchPredictT<11>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 12: launchPredictT<12>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 13: launchPredictT<13>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 14: launchPredictT<14>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	case 15: launchPredictT<15>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
	default: launchPredictT<16>(d, sBase, nStreams, hopBase, tileHops, plain, passADone, st); break;
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
	case 8: launchChainT<8>(d, sBase, nStreams, hopBase, st); break;
	case 9: launchChainT<9>(d, sBase, nStreams, hopBase, st); break;
	case 10: launchChainT<10>(d, sBase, nStreams, hopBase, st); break;
	case 11: launchChainT<11>(d, sBase, nStreams, hopBase, st); break;
	case 12: launchChainT<12>(d, sBase, nStreams, hopBase, st); break;
	case 13: launchChainT<13>(d, sBase, nStreams, hopBase, st); break;
	case 14: launchChainT<14>(d, sBase, nStreams, hopBase, st); break;
	case 15: launchChainT<15>(d, sBase, nStreams, hopBase, st); break;
	default: launchChainT<16>(d, sBase, nStreams, hopBase, st); break;
	}
}
void launchVocoderMany(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st) {
	countLaunch(LK_VOC_N);
	switch (d.C) {
	case 3: launchVocoderN<3>(d, sBase, nStreams, hopBase, plain, st); return;
	case 4: launchVocoderN<4>(d, sBase, nStreams, hopBase, plain, st); return;
	case 5: launchVocoderN<5>(d, sBase, nStreams, hopBase, plain, st); return;
	case 6: launchVocoderN<6>(d, sBase, nStreams, hopBase, plain, st); return;
	case 7: launchVocoderN<7>(d, sBase, nStreams, hopBase, plain, st); return;
	default: launchVocoderN<8>(d, sBase, nStreams, hopBase, plain, st); return;
	}
}

} // namespace smst
