// Shared host/device plain-data types for the gfx950 stretch engine.
#pragma once
#include <cstdint>

namespace smst {

constexpr int kTileHops = 64;   // hops per tile = lanes of the wave that runs the bin recurrence
constexpr int kMaxChannels = 16; // per-lane channel arrays of the recurrence kernels are sized at compile time: 1-2 channels kVocoder, 3-8 kVocoderN (records in LDS),
                                // 9-16 the un-fused pair kPredictB + kChain (records through HBM: slower, the same arithmetic)
constexpr int kMaxFusedChannels = 8;
// floats of one record of the bin recurrence (smst_recurrence.h: computeRecord / recordChannelFields): four twists, the maximum channel, then per-channel fields
constexpr int recordFloats(int channels) { return channels <= 2 ? 9 + 3*channels : 12 + 2*channels; }
constexpr int kMaxFftPasses = 12;
constexpr int kTileHasStride = 12; // per-tile summary bytes of the host scheduler: any hop / mapped / formants / new spectrum / random time factor / analysis window in the call / reaching into the history / a start bin / a pre-analysed hop / a hop without a new spectrum / a formant hop that estimates its base frequency
constexpr int kEnergyParts = 16; // partial sums per stream in the silence-gate reduction

// Hop flags (reference: signalsmith-stretch.h:299-313)
enum : unsigned {
	HOP_ACTIVE = 1u,
	HOP_NEW_SPECTRUM = 2u,   // :299
	HOP_REANALYSE_PREV = 4u, // :303
	HOP_MAPPED = 8u,         // :300
	HOP_FORMANTS = 16u,      // :310
	HOP_RANDOM_TF = 32u,     // :639
	HOP_PREANALYSED = 64u,   // split computation: the block began in an earlier call and was analysed then (:293, :332-373); its spectra come from the pending buffers
};

// Source codes for the per-hop spectra (which row holds Band.input / Band.prevInput for this hop)
constexpr int SRC_STATE = -1;      // carried state row (st_input resp. st_prev)
constexpr int SRC_REANALYSED = -2; // this hop's own re-analysed previous window (Xprev row k)

// One hop of one stream inside a tile.  Filled by the host scheduler (K0), read by every kernel.
struct HopDesc {
	int inputOffset;  // window end of the current analysis, relative to the call's first input sample (:288)
	int inSrc;        // >=0: Xcur row of that tile-local hop; SRC_STATE
	int prevSrc;      // >=0: Xcur row; SRC_STATE; SRC_REANALYSED
	unsigned flags;
	float timeFactor; // already clamped to >= 1/maxCleanStretch (:638)
	int outPos;       // output index (within the call) at which this hop fires (:280-285)
	unsigned seed;    // counter-based RNG stream for timeFactor > 2 (:639-640)
	int startBin;     // 0, except for a split-computation block that a flush() interrupted between two chunks of the main prediction
	                  // (:722-803 ran for the bins below, then :458-463 zeroed them): outputs below this bin stay zero (kVocoderOne only)
};

// Per-stream emission window of a tile (K4 overlap-add gather)
struct EmitDesc {
	int nLo, nHi;     // output samples [nLo, nHi) of the call are final after this tile
	int firstHopPos;  // outPos of the tile's first hop
	int hopCount;     // hops of this stream in this tile
};

// Per-stream parameters (reference: signalsmith-stretch.h:107-135, 513-517)
struct StreamParams {
	float freqMultiplier;     // :108
	float freqTonalityLimit;  // :110-113
	float formantMultiplier;  // :125
	float invFormantMultiplier;
	float formantBaseFreq;    // :134
	int formantCompensation;  // :127
	int hasCustomMap;         // :120 (table form)
	int mapLen;               // points of this stream's table (the batch's array has one row pitch: the longest table's length)
	int mapSlot;              // which of the stream's kMapSlots table rows holds them: a new table goes to a row that no step of a split-computation
	                          // block in flight has latched (Batch::setFreqMapTable), as the reference's std::function is replaced as a whole (:120-122)
};
constexpr int kMapSlots = 3; // findPeaks' table (:874) + updateFormants step 2's (:1020) of the block in flight + the live one

// fp16 storage of the carried state (opt-in per batch): fp32 <-> half conversions only, no half arithmetic
typedef _Float16 half_t;
struct Half2 { half_t x, y; };

// Prediction.input and Prediction.energy of one (hop, channel, bin), packed to 12 bytes: written once by pass A, read by the
// record producers with one 12-byte load (the 16-byte float4 it replaces carried a quarter of dead bytes through HBM)
struct PredEntry { float x, y, e; };

struct FftPlan {
	int H;      // complex FFT length = fftSamples/2 = bands
	int N;      // fftSamples
	int npass;
	int radix[kMaxFftPasses];
};

} // namespace smst
