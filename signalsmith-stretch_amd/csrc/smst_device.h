// Device-side view of a batch (plain pointers, passed to kernels by value) and the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include "smst_types.h"

namespace smst {

struct DevBatch {
	// geometry (signalsmith-stretch.h:71-94; fftSamples/bands from the L1 contract)
	int S, C, B, I, M, N, L, T;
	int Mp;                       // row pitch (elements) of the per-tile [.][C][M] arrays: M + 32 rounded up to 16
	int recPitch;                 // float4 per wavefront step in REC: chunks*64 + 16
	int histLen, carryLen, delta; // B+I, B+I, split ? I : 0
	int carryPitch;               // B+I + 2*I: behind the window the rows hold what an empty ring position holds (sum 0, product 1e-30) -- written once, by kResetStreams
	int histPitch;                // 2*(B+I): a call's input is appended behind the window until the row is full, then the window moves back to the front
	int lag, ringSlots;           // wavefront skew (>= L+1) and LDS ring depth (power of two > lag)
	int recSteps;                 // record rows per stream: M + lag*(T-1) rounded up to 64, plus prefetch slack
	int hopStride, emitStride;    // row pitch of the per-call hop / emit tables
	int mapTableLen;
	int histCur, carryCur;        // which half of the double buffers is current
	int debugMode;                // always 0 in the product; builds with -DSMST_EXPERIMENTS read SMST_DEBUG_MODE (timing experiments)
	int noFeedFusion;             // SMST_NO_FEED_FUSION=1: pass A stays its own kernel (kPredictA) -- cross-check of the folded forms; =2: formant tiles in two passes over the spectra
	                              // (kFeedScanA + kFeedFreq + kFeedScanC with pass A folded) even where the one-pass form applies
	int feedSerial;               // SMST_FEED_SERIAL: bin-by-bin feed recurrences (kFeedSerial) instead of the scan form
	int halfState;                // carried Band.output / Prediction.energy / overlap-add sums stored in fp16 (BASELINE config 5 "fp16 internal")
	int fftLean;                  // SMST_FFT_TABLES=lean: register-blocked FFT kernels with the smaller tables (opt-in experiment, see smst_engine.cpp)
	int noFastFft;                // SMST_NO_FAST_FFT: the generic radix-4/2/3/5 ladder even where a register-blocked FFT exists (cross-check)
	int fftTeams;                 // analysis / synthesis by persistent workgroups of three free-running teams, tables in LDS (default; SMST_FFT_TEAMS=0: one frame per workgroup; =2: teams even for tiles with few frames per team -- tests)
	int teamsGrid;                // their grid: one workgroup per CU, a multiple of 8
	int synthEmit;                // synthesis + overlap-add + emission in one kernel (kSynthEmitTeams) where it applies (default 1; SMST_SYNTH_EMIT=0: kSynthTeams + kEmit; =2: also for small tiles -- tests)
	int vocNWide;                 // 3-8 channels: producer passes of 8 rows x 8 steps instead of 16 rows x 4 (SMST_VOCN_WIDE, smst_switches.h)
	int vocWide;                  // mono / stereo, gathering producers (mapped tiles): passes of 4 rows x 16 steps instead of 8 x 8 (SMST_VOC_WIDE=0: the first form)
	int vocNHalfLines;            // 3-8 channels: the writer wave stores aligned 64-byte half lines (1) or whole 128-byte lines (2) instead of 32-byte sectors (SMST_VOCN_HALF_LINES=0: the first form)
	int noStage;                  // SMST_NO_STAGE: producers of the fused kernel gather from HBM even where staging applies
	int alignAll;                 // SMST_ALIGN_ALL: the line-aligned producers for every geometry they are valid for (default: L = 4 only)
	int noAlign;                  // SMST_NO_ALIGN: staged producers with per-row windows and lag L+1 (round 3) instead of the line-aligned form
	FftPlan plan;
	// constant tables
	const float2 *twH;     // e^{-2 pi i j / H}
	const float2 *halfTw;  // e^{-i pi m / N}
	const float2 *rot;     // per-bin hop rotation (signalsmith-stretch.h:647-655)
	const float2 *twA, *twB; // fast FFT (H = 256*R3): stage twiddles laid out [n-1][p]
	const float2 *winA, *winB; // analysis window folded with e^{-i pi m/N}: u[m] = x[m+B/2]*winA[m] + x[m-H+B/2]*winB[m]
	const float4 *win4;        // (winA[m], winB[m]) interleaved: one 16-byte load per element in the fast analysis kernel
	const float4 *synTab;      // (halfTw[m], window[m+B/2] or 0, window[m-M+B/2] or 0): one 16-byte load per synthesis output
	const float4 *twA4, *twB4; // stage twiddles of the register-blocked FFT, rows (2i, 2i+1) paired: [8][16*R3], [8][R3]
	const unsigned *lcgPow;    // 16807^(j+1) mod (2^31 - 1), j < 2M: jump-ahead factors of the reference's random engine (smst_kernels_common.h: engineDraw)
	const float4 *twA6;        // lean form of twA4: (w^1, w^2), (w^3, w^4), (w^8, w^12) per thread, [3][16*R3]
	const float2 *win2, *syn2; // lean forms of win4 / synTab: the two window samples of an element only, [M]
	const float *window;   // analysis == synthesis window (Kaiser, perfect reconstruction)
	const float *wprod;    // window[i]^2 * N
	// per-stream state
	float2 *stInput, *stPrev, *stOut; // Band.input / .prevInput / .output   [S][C][M]
	float *stEnergy;                  // Prediction.energy                   [S][C][M]
	float *hist;                      // input history, a sliding window of B+I samples per row        [S][C][histPitch]
	int *histBase[2];                 // where each stream's window begins in its rows (double-buffered with histCur: kHistory reads one, writes the other)  [S]
	float *carrySum[2];               // overlap-add partial sums: a window of B+I per row    [S][C][carryPitch]
	float *carryWp[2];                // window products                                      [S][carryPitch]
	int *carryBase[2];                // where each stream's window begins in the rows of that half: a call without a hop emits from the front of the window
	                                  // and moves its beginning on (kEmitCarried) instead of copying what is left; everything else writes a window at 0.  [S]
	float *wpHead;                    // kSynthEmitTeams: the window products of a tile's first samples (those the carry reaches into), [S][wpHeadLen], written by kEmitProducts
	int wpHeadLen;                    // (ceil(B/I) + 2)*I
	float *stFreq;                    // freqEstimateWeighted / Weight       [S][2]
	const StreamParams *params;       // [S]
	// split computation spreads a block's steps over its interval (signalsmith-stretch.h:321-325) and the steps read the LIVE parameters:
	// findPeaks (:874), updateFormants step 0 (:982-983) and step 2 (:1020-1021) of the block in flight may each see other values.  In every
	// other launch the three point at `params`.
	const StreamParams *paramsPeaks, *paramsForm0, *paramsForm2;
	const float *mapTable;            // [S][kMapSlots][mapTableLen] custom frequency maps (table form of setFreqMap; StreamParams.mapSlot selects the row)
	// per-call tables
	const HopDesc *hops;   // [S][hopStride]
	const EmitDesc *emit;  // [S][emitStride]
	// per-tile workspace, [subS][T][C][M] unless noted
	float2 *Xcur, *Xprev, *OUT;
	PredEntry *PE;  // (Prediction.input, Prediction.energy) per (hop, channel, bin), 12 bytes: [subS][T][C][Mp]
	float2 *dump;   // [subS][C][64] parking lot for the recurrence kernel's out-of-range lanes
	float4 *REC;    // skewed per-step records of the bin recurrence [subS][recSteps][chunks][64 lanes]
	float2 *map;    // [subS][T][M]
	float *ratio;   // [subS][T][M]
	float *envelope; // [subS][T][M] test hook: the formant envelope (formantMetric after its eight passes, :984-1006) of every hop -- written only by the
	                 // separate envelope kernel (SMST_NO_FEED_FUSION=1), null otherwise (smst_batch_debug_get_formants)
	float *energyT, *smoothT; // [subS][M][64 hops]: feed scratch, hop index fastest
	float2 *peaksT;           // [subS][M/2 + 2][64 hops]
	float *est;     // [subS][T][2]
	float *freqEst; // [subS][T]: pitch estimate (in bins) each hop's formant envelope uses
	float *frames;  // [subS][T][C][B]
	float2 *fftScratch; // [subS][T][C][Mp]: second FFT buffer of the synthesis frames where two buffers do not fit LDS (fftNeedsScratch); null otherwise
	const int *nHops;      // [subS] hops of this tile
	const int *lastNewHop; // [subS] tile-local index of the last hop with a new spectrum, or -1
};

// The continuous wavefront of the fused mono / stereo recurrence (kVocoderCont, smst_vocoder_cont.hip): launch `tile` covers the global
// blocks [n0, n1) of a wavefront that runs through ALL tiles of a call without draining -- the rows (lanes) finish tile - 1 and begin
// `tile` in it.  `[0]` = the workspace of tile - 1, `[1]` = the workspace of `tile`.
struct ContArgs {
	float2 *Xcur[2], *Xprev[2], *OUT[2];
	const int *tileInfo; // [tile][2][subS]: hops per stream and tile (row 0 of each pair), as the tile kernels' DevBatch::nHops
	int tileStride;      // 2*subS
	int nTiles;          // tiles of the call
	int tile;            // the tile that BEGINS in this launch (its row 0 starts at block n0 = tile*period), counted from the wavefront's first tile
	int hopBase;         // index of that first tile's first hop in the rows of the call's hop table
	int n0, n1;          // global block range; both even
	int period;          // blocks per tile and row: M/8 + 2 (a zero line between two hops of a row: see the kernel)
	float2 *save;        // [S][8*C*64]: the recurrence wave's last eight outputs per lane, from one launch to the next
	int writerWave;      // which wave drains the results: 4 (the recurrence wave's SIMD, as in kVocoder) or 11 (the SIMD that holds two producers)
};

struct IoArgs {
	const float *in;
	float *out;
	long long inStreamStride, inChannelStride, outStreamStride, outChannelStride;
	const int *inSamples;  // [S] device
	const int *outSamples; // [S] device
};

// The packed input of a frame: element m = (x[base + halfB + m] w[halfB + m], x[base + halfB - M + m] w[halfB - M + m]), the real part
// present for m < B - halfB, the imaginary part for m >= M - halfB.  At 48 / 96 kHz (block = 15/16 of the FFT size) those edges are
// element-slot boundaries: slot 0 has no imaginary part, slot 15 no real part, and kAnalyseTeams fetches exactly the window.  Other
// block sizes (44.1 kHz: 5292 of 6144) end inside a slot; the team kernel then still fetches WHOLE slots 0..14 / 1..15 -- up to
// `windowPad` samples on either side of the window, which the table's zero weights cancel -- so a frame is its to take only if
// that wider span lies in this call's input too.  (First form: one compare + select per element and half: analysis 4.4 -> 6.8 ms per
// step at 44.1 kHz, slower than the per-frame kernel it was to replace.)
// The generic FFT kernels ping-pong between two buffers of `bands` complex values: in LDS while both fit (150 KiB), else one of them in memory
// (kAnalyse<true> / kSynth<true>: presetDefault / presetCheaper at 176.4 / 192 kHz).  One buffer must still fit.
__host__ __device__ inline bool fftNeedsScratch(int bands) { return (size_t)bands*16 > (size_t)150*1024; }
constexpr int kMaxBands = 150*1024/8; // 19200

struct WindowPad { int lo, hi; };
__host__ __device__ inline WindowPad windowPad(int B, int M) {
	const int halfB = B/2, MA = M/16;
	WindowPad p;
	p.lo = (M - halfB) - MA;          // samples in front of the window that slot 1's imaginary parts reach
	p.hi = 15*MA - (B - halfB);       // samples behind it that slot 14's real parts reach
	return p;
}
// The frames that kAnalyseTeams takes (the others reach into the carried history, or too close to the end of the input, and go to the
// per-frame kernel's bounds-checked path).  The host evaluates the same condition (smst_engine.cpp).
__host__ __device__ inline bool analysisWindowInCall(int B, int M, int I, int inputOffset, int which, int inSamples) {
	const WindowPad p = windowPad(B, M);
	const int end = inputOffset - (which ? I : 0);
	return p.lo >= 0 && p.hi >= 0 && end - B - p.lo >= 0 && end + p.hi <= inSamples;
}

// Which kernel variant a launcher chose, counted per process (test hook: smst_debug_launch_count).  "Bit-identical to the other
// form" tests assert through these that BOTH forms really ran.
enum LaunchKind {
	LK_VOC_ALIGNED, LK_VOC_STAGED, LK_VOC_GATHER, LK_VOC_N, LK_VOC_ONE, LK_VOC_ACROSS, LK_VOC_CONT, LK_CHAIN_UNFUSED,
	LK_ANALYSE_TEAMS, LK_ANALYSE_FAST, LK_ANALYSE_GENERIC, LK_SYNTH_TEAMS, LK_SYNTH_FAST, LK_SYNTH_GENERIC, LK_SYNTH_EMIT, LK_EMIT_CARRIED, LK_FEED_ONE_PASS, LK_COUNT
};
long long launchCount(const char *name); // -1: unknown name

void launchEnergy(const DevBatch &d, const IoArgs &io, int sBase, int nStreams, int maxSamples, float *energyOut, hipStream_t st);
void launchAnalyse(const DevBatch &d, const IoArgs &io, int sBase, int nStreams, int hopBase, int tileHops, bool anyInCall, bool anyLate, hipStream_t st);
bool launchFeed(const DevBatch &d, int sBase, int nStreams, int hopBase, int tileHops, bool anyFormants, bool anyEstimatedBase, hipStream_t st); // true: pass A done too
void launchPredict(const DevBatch &d, int sBase, int nStreams, int hopBase, int tileHops, bool plain, bool passADone, hipStream_t st);
void launchChain(const DevBatch &d, int sBase, int nStreams, int hopBase, hipStream_t st);
void launchPredictFused(const DevBatch &d, int sBase, int nStreams, int hopBase, int tileHops, bool plain, bool passADone, hipStream_t st);
void launchVocoder(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, bool bounded, hipStream_t st); // bounded: no random time factors in the tile
bool fusedSupported(const DevBatch &d);
bool continuousSupported(const DevBatch &d); // the geometries of the line-aligned producers (mono / stereo, L <= 4, M a multiple of 16, fp32 state)
int continuousPeriod(const DevBatch &d);     // blocks per tile of the continuous wavefront
void launchVocoderContinuous(const DevBatch &d, const ContArgs &a, int sBase, int nStreams, hipStream_t st);
bool singleHopSupported(const DevBatch &d);
void launchVocoderOne(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st); // tiles in which no stream has more than one hop
bool acrossSupported(const DevBatch &d);
void launchVocoderAcross(const DevBatch &d, int sBase, int nStreams, int hopBase, bool plain, hipStream_t st); // the same tiles, mono / stereo: lanes of the recurrence wave = streams
void launchSynth(const DevBatch &d, int sBase, int nStreams, int hopBase, int tileHops, hipStream_t st);
void launchEmit(const DevBatch &d, const IoArgs &io, int sBase, int nStreams, int tileIndex, int maxSpan, hipStream_t st);
void launchEmitCarried(const DevBatch &d, const IoArgs &io, hipStream_t st); // a call without hops, every stream (emit table row 0)
// synthesis + overlap-add + emission in one kernel: whether it applies to a tile; its window products (needs nothing of the tile's
// spectra: launched ahead of the recurrence's completion); the kernel itself
bool synthEmitApplies(const DevBatch &d, int nStreams, int tileHops);
void launchEmitProducts(const DevBatch &d, int sBase, int nStreams, int tileIndex, hipStream_t st);
void launchSynthEmit(const DevBatch &d, const IoArgs &io, int sBase, int nStreams, int tileIndex, hipStream_t st);
void launchCarryFeed(const DevBatch &d, int sBase, int nStreams, int hopBase, bool anyFormants, hipStream_t st);
void launchCarryOut(const DevBatch &d, int sBase, int nStreams, hipStream_t st);
void launchHistory(const DevBatch &d, const IoArgs &io, int span, hipStream_t st);
void launchPassThrough(const DevBatch &d, const IoArgs &io, const int *passFlags, int maxOut, hipStream_t st);
// flags: per-stream bit masks or null = allBits for every stream.  keep (may be null): per stream, the first samples of the overlap-add ring that
// stft.reset() does NOT reach -- split computation reads the rest of the interval from its stashed copy of the ring (signalsmith-stretch.h:407-415)
void launchResetStreams(const DevBatch &d, const int *flags, int allBits, const float *seedWp, hipStream_t st, const int *keep = nullptr);
// split computation: the spectra of the block in flight (analysed at the block's start, :293,:356-373) -> row 0 of the tile that runs the block
void launchPendingToTile(const DevBatch &d, int sBase, int nStreams, const float2 *pendIn, const float2 *pendPrev, hipStream_t st);
// ... and a flush() that fell between two synthesis steps (:397-399): channels from synthChannels[stream] on contribute no frame (negative: all do)
void launchMaskOutRows(const DevBatch &d, int sBase, int nStreams, const int *synthChannels, hipStream_t st);
void launchSeekHistory(const DevBatch &d, const IoArgs &io, const int *seekFlags, hipStream_t st);
void launchAddPreRoll(const DevBatch &d, const float *preRoll, int length, const int *offsets, hipStream_t st);
void launchComplexSelfTest(const float *in, float *out, int n, hipStream_t st); // smst_complex.h against its documented formulas (test hook)
void launchFlushTail(const DevBatch &d, const IoArgs &io, const int *tailOffset, const int *outOffset, hipStream_t st);

} // namespace smst
