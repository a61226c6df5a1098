// Host-side engine: owns a batch of independent streams on one GPU, mirrors the reference's block scheduler
// (signalsmith-stretch.h:280-319) per stream, and drives the gfx950 kernels tile by tile.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#include <hip/hip_runtime.h>
#include "smst_device.h"

namespace smst {

struct StreamSched { // reference members: signalsmith-stretch.h:494-529
	size_t samplesSinceLast = SIZE_MAX; // blockProcess.samplesSinceLast, :496
	int prevInputOffset = -1;           // :527
	bool didSeek = false;               // :528
	float seekTimeFactor = 1;           // :529
	size_t silenceCounter = 0;          // :510
	bool silenceFirst = true;           // :511
	unsigned seed = 0;                  // randomEngine, :616 -- the state of libstdc++'s minstd_rand0 (smst_kernels_common.h: engineDraw)
};

// Split computation (signalsmith-stretch.h:292-296,321-325): a block's steps are spread over its interval, so between two interval
// boundaries one block is IN FLIGHT -- analysed from the input as it stood at the block's start (:293), everything else still to come,
// and the steps that come read the live parameters and the live Band state.  The engine analyses such a block when it starts, keeps its
// spectra, and runs the rest when the interval is complete (or when a flush() needs part of it); what happened in between is recorded here.
struct PendingBlock {
	bool valid = false;
	unsigned flags = 0;      // HopDesc.flags as latched at the block's start (:299-310)
	float timeFactor = 1;    // :312
	unsigned seed = 0;       // the random engine's state before this block's draws
	int startBin = 0;        // a flush() between two chunks of the main prediction (:722-803): the bins below were computed, then zeroed (:458-463)
	bool zeroPrevAfter = false; // a flush() after `.input -> .prevInput` (:806-811): Band.prevInput stays zero
	// the parameters as findPeaks (:874), updateFormants step 0 (:982) and step 2 (:1020) saw them, once each has run (frozen*)
	bool frozenPeaks = false, frozenForm0 = false, frozenForm2 = false;
	StreamParams peaks{}, form0{}, form2{};
};
struct StepLayout { // index of each step of a block in the reference's order (:304-318, :620-632); -1: the block has no such step
	int reanalyse0 = -1, analyse0 = -1, copyIn = -1, peaks = -1, form0 = -1, form2 = -1, main0 = 0, prevCopy = -1, spectrum = 0, synth0 = 0, steps = 0;
};

struct BatchTimings { // filled when profiling is enabled (hipEvent pairs around each kernel class)
	double analyseMs = 0, feedMs = 0, predictMs = 0, chainMs = 0, synthMs = 0, emitMs = 0, otherMs = 0;
	long analyseLaunches = 0, synthLaunches = 0, chainLaunches = 0, predictLaunches = 0, emitLaunches = 0;
	double chainLiveMs = 0;   // mode 2: the recurrence kernel timed in place (events on its own stream, nothing serialised)
	long chainLiveLaunches = 0;
};

class Batch {
public:
	Batch(int streams, int channels, int block, int interval, bool split, int device, long seed, bool halfState = false);
	~Batch();
	Batch(const Batch &) = delete;
	Batch &operator=(const Batch &) = delete;

	// geometry queries (signalsmith-stretch.h:42-47,96-104,166-168,205-207)
	int streams() const { return S; }
	int channels() const { return C; }
	int blockSamples() const { return B; }
	int intervalSamples() const { return I; }
	int fftSamples() const { return N; }
	int bands() const { return M; }
	bool splitComputation() const { return split; }
	bool halfPrecisionState() const { return halfState; }
	int inputLatency() const { return B - B/2; }
	int outputLatency() const { return B/2 + (split ? I : 0); }
	int seekLength() const { return B + I; }
	int outputSeekLength(float playbackRate) const { return int(inputLatency() + playbackRate*outputLatency()); }

	void reset(); // :49-60, all streams
	// parameter setters, stream = -1 for all (signalsmith-stretch.h:107-135)
	void setTransposeFactor(int stream, float multiplier, float tonalityLimit);
	void setTransposeSemitones(int stream, float semitones, float tonalityLimit);
	void setFormantFactor(int stream, float multiplier, bool compensatePitch);
	void setFormantSemitones(int stream, float semitones, bool compensatePitch);
	void setFormantBase(int stream, float baseFreq);
	void setFreqMapTable(int stream, const float *table, int n); // table form of setFreqMap (:120); n == 0 clears

	// All sample pointers are DEVICE pointers here (planar: stream stride, channel stride, contiguous samples).
	// `active` (host, may be null) masks streams out of the call entirely.
	void process(const float *in, long long inStreamStride, long long inChannelStride, const int *inSamples,
	             float *out, long long outStreamStride, long long outChannelStride, const int *outSamples,
	             const unsigned char *active = nullptr); // :210-423
	void seek(const float *in, long long inStreamStride, long long inChannelStride, const int *inSamples,
	          const double *playbackRates, const unsigned char *active = nullptr); // :140-165
	void flush(float *out, long long outStreamStride, long long outChannelStride, const int *outSamples,
	           const float *playbackRates, const unsigned char *active = nullptr); // :427-464
	void outputSeek(const float *in, long long inStreamStride, long long inChannelStride, const int *inputLengths); // :173-204
	void synchronize();

	hipStream_t stream() const { return st; }
	int device() const { return dev; }
	void growLivePool(size_t pairs);
	void enableProfiling(int mode); // 1: every kernel class, serialised; 2: recurrence kernel in place (timing events come from a pool created here, not inside process())
	BatchTimings takeTimings();
	// Host time of process() since the last take: wall time inside the calls, the part of it spent WAITING for the device (the call before the
	// previous one to finish -- there are two sets of per-call tables -- and the silence gate's readback), and the number of calls.  What is
	// left is the host's own work: the per-stream block scheduler, the table fills and the enqueues.
	struct HostTimes { double callMs = 0, waitTablesMs = 0, waitGateMs = 0; long calls = 0; };
	HostTimes takeHostTimes() { HostTimes t = hostTimes; hostTimes = HostTimes(); return t; }
	size_t workspaceBytes() const { return wsBytes; }
	long allocationEvents() const { return allocEvents; } // test hook: must not move across steady-state process() calls
	// order the batch's work after everything already enqueued on `other` / make `other` wait for the batch's work so far
	void waitForStream(hipStream_t other);
	void signalStream(hipStream_t other);
	int subBatchStreams() const { return subS; }
	int lastBlockSteps(int stream) const { return (stream >= 0 && stream < S) ? lastSteps[stream] : 0; } // blockProcess.steps of the stream's newest block (:284-318)
	int lastCallBlocks(int stream) const { return (stream >= 0 && stream < S) ? lastStarts[stream] : 0; } // blocks that began in the stream's most recent process() (:281)
	// the object is copyable in the reference (a plain struct, signalsmith-stretch.h:34-35): same geometry required; every piece of
	// carried state, the parameters and the scheduler state of `other` replace this batch's
	void copyStateFrom(Batch &other);
	// What the reference's configure() (:71-94) leaves ALONE when an instance is configured again: the random engine (seeded in the
	// constructor only, :38-39), prevInputOffset, didSeek / seekTimeFactor, the silence counter and the pitch-estimate averages
	// (all reset by reset(), :49-60, not by configure).  `other` is the stream's previous batch (same stream count).
	void inheritAcrossConfigure(Batch &other);

	// test hooks: copy state rows to the host (which: 0 input, 1 prevInput, 2 output -> 2*C*M floats; 3 energy -> C*M)
	void debugGetState(int stream, int which, float *dst);
	void debugGetCarry(int stream, float *sums, float *products); // [C][B+I], [B+I]
	// teacher-forcing hooks (tests only): overwrite the carried per-bin state / overlap-add carry of one stream
	void debugSetState(int stream, int which, const float *src);
	void debugSetCarry(int stream, const float *sums, const float *products);
	// output map (inputBin, freqGrad per bin, signalsmith-stretch.h:587-590) of the stream's last hop of the last process()
	// call; false if that hop had no frequency map
	bool debugGetMap(int stream, float *dst);
	// formant stage of the stream's newest hop of the last process() call (separate feed kernels only): energy ratio per bin, envelope per bin,
	// the pitch estimate in bins; false if that hop had no formant processing or the batch runs the fused feed kernels
	bool debugGetFormants(int stream, float *ratio, float *envelope, float *freqEstimate);

private:
	int S, C, B, I, N, M, L;
	bool split;
	bool halfState = false; // carried Band.output / Prediction.energy / overlap-add sums in fp16 (fp32 arithmetic throughout)
	int dev;
	hipStream_t st = nullptr;       // feed-forward kernels + everything the caller synchronises on
	hipStream_t stChain = nullptr;  // the bin recurrence (few waves, latency-bound): overlaps with the bulk kernels
	hipStream_t stSynth = nullptr;  // synthesis + emission of the previous tile
	hipStream_t stGate = nullptr;   // silence-gate reduction + table uploads of the NEXT call, while the previous call still runs
	// Per-call device tables exist twice: a call fills one set on `stGate` while kernels of the previous call read the other
	// ... and so does their PINNED host staging (h*): nothing pageable is handed to an async copy, so the only host
	// synchronisation of a call is the silence-gate readback
	struct CallSet {
		int *inSamples, *outSamples, *flags, *tileInfo, *resetBits; HopDesc *hops; EmitDesc *emit;
		int *hInSamples, *hOutSamples, *hFlags, *hTileInfo, *hResetBits; HopDesc *hHops; EmitDesc *hEmit; float *hEnergy;
		HopDesc *pendHops = nullptr, *hPendHops = nullptr; // split computation: the block each stream leaves in flight at the end of the call ([S], analysis only)
		size_t hopsCap, emitCap, tileInfoCap;
		hipEvent_t done, tables; bool used;
		std::vector<void *> retiredDevice, retiredPinned; // outgrown tables that the previous call may still read
	} callSets[2]{};
	int callCur = 0;
	hipEvent_t evStart = nullptr, evFeed[3] = {nullptr, nullptr, nullptr}, evChain[3] = {nullptr, nullptr, nullptr}, evOut[3] = {nullptr, nullptr, nullptr}, evSynth[3] = {nullptr, nullptr, nullptr}; // (the third set: the continuous wavefront's tile pipeline is one stage deeper)
	struct TileBuffers { float2 *Xcur, *Xprev, *OUT, *dump, *map, *peaksT; float4 *REC; PredEntry *PE; float *ratio, *envelope, *energyT, *smoothT, *est, *freqEst, *frames; float2 *fftScratch; } slots[3]{}; // [2]: only what a plain tile touches, only where the continuous wavefront applies (allocateWorkspace)
	bool overlap = true, noFuse = false, noSingleHop = false, noAcross = false, carriedEmit = true, continuous = false; // (smst_switches.h)
	int contWriterWave = 4;
	float2 *dContSave = nullptr; // kVocoderCont: the recurrence wave's history between two launches, [S][8*C*64]
	double workspaceGiB = 0;
	int subStreamsAsked = 0;
	int subS = 0;
	// per-call host scratch, kept between calls (no heap traffic in steady state); growth events are counted
	std::vector<int> hopFirst, hopCount, maxSpanV;
	std::vector<unsigned char> tileHasV, passV, leavesPendingV, ridesV;
	long allocEvents = 0; // device allocations + pinned allocations + host table growth since construction
	size_t wsBytes = 0;
	DevBatch d{};
	std::vector<StreamSched> sched;
	unsigned lcgHopJump = 1; // 16807^(2M - 2) mod (2^31 - 1): what one randomised hop advances a stream's engine by
	struct LastHop { int slot = -1, local = -1, subLocal = 0; bool mapped = false, formants = false; };
	std::vector<LastHop> lastHop; // where each stream's newest hop sits in the tile workspaces (debugGetMap)
	std::vector<StreamParams> params;
	bool paramsDirty = true;
	// ---- split computation: the block in flight (see PendingBlock) ----
	std::vector<int> lastSteps, lastStarts;
	std::vector<int> carryBase; // the host's copy of DevBatch::carryBase[carryCur]
	EmitDesc *dZeroEmit = nullptr;
	std::vector<int> histBase;  // the host's copy of DevBatch::histBase (where each stream's input-history window begins)
	std::vector<PendingBlock> pend;
	float2 *dPendIn = nullptr, *dPendPrev = nullptr; // its spectra, [S][C][Mp]: Band.input and (re-analysed) Band.prevInput
	struct PendSet { // tables of one run of blocks in flight (double-buffered like CallSet: pinned staging, asynchronous uploads)
		HopDesc *hops, *hHops; EmitDesc *emit, *hEmit; int *tileInfo, *hTileInfo, *bits, *hBits, *synthChannels, *hSynthChannels;
		StreamParams *prm[3], *hPrm[3];
		hipEvent_t done; bool used;
	} pendSets[2]{};
	int pendCur = 0;
	int *dZeroCounts = nullptr; // [S] zeros: the sample counts of a run that consumes and emits nothing
	std::vector<int> pendList;  // streams of the run being prepared
	std::vector<unsigned char> pendTileHas;
	std::vector<int> pendMaxSpan, keepV;
	StepLayout stepLayout(unsigned flags) const;
	size_t stepsExecuted(size_t steps, size_t samplesIntoInterval) const;
	void freezePendingParams(int s);   // before a setter changes params[s]
	unsigned seedAfterDroppedBlock(int s) const; // the random engine once the block in flight is dropped (reset / silence / configure)
	void runPendingBlocks(const int *synthChannels); // runs the blocks of `pendList` (a tile of one hop per stream; no input, no output samples)
	struct TileRun { const IoArgs *io; int nTiles, maxHops; const unsigned char *tileHas; const int *maxSpan; const int *dTileInfo; bool pendingRun; const int *dSynthChannels; bool carriedOnly; };
	void settleCarry();
	void runTiles(const TileRun &run);
	void runTilesRange(const TileRun &run, int tile0, int tile1, int carryFirst); // tile by tile
	bool continuousApplies(const TileRun &run, int tile) const;
	void runTilesContinuous(const TileRun &run, int tile0, int tile1, int carryFirst); // the recurrence as ONE wavefront through the tiles [tile0, tile1) (kVocoderCont)
	bool profiling = false, liveTiming = false;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> liveEvents; // pairs recorded since the last takeTimings()
	std::vector<std::pair<hipEvent_t, hipEvent_t>> livePool;   // every pair ever created; [0, liveEvents.size()) are in use
	BatchTimings timings;
	HostTimes hostTimes;

	std::vector<void *> allocations;
	float *dEnergy = nullptr;
	int *dInSamples = nullptr, *dOutSamples = nullptr, *dFlags = nullptr, *dAux0 = nullptr, *dAux1 = nullptr;
	HopDesc *dHops = nullptr;
	EmitDesc *dEmit = nullptr;
	int *dTileInfo = nullptr;
	int *hSeek = nullptr;          // pinned staging of seek()
	float *hSeekEnergy = nullptr;
	float *dSeedWp = nullptr; // reset(0.1) window-product seed, device copy
	hipEvent_t evOrder = nullptr;
	std::vector<void *> pinned;
	template <typename T> T *pinnedAlloc(size_t count);
	void pinnedFree(void *p);
	void releaseAll();
	void resetStreams(const int *bitsHost, int allBits, const int *keepHost = nullptr); // per-stream bit masks (kResetStreams) or null = allBits for all; keepHost: see launchResetStreams; synchronous upload (reset / flush)
	bool checkLaunches = false; // SMST_CHECK_LAUNCHES=1: hipGetLastError() after every launch group of process(), not only at its end
	void checkLaunch(const char *what);
	int *dResetBits = nullptr, *dKeep = nullptr;
	std::vector<int> resetBitsV;
	float *dZeros = nullptr;
	size_t zerosCapacity = 0;
	float *dScratchOut = nullptr;
	size_t scratchOutCapacity = 0;
	StreamParams *dParams = nullptr;
	float *dMapTable = nullptr;
	std::vector<float> hostMapTable;
	std::vector<float> seedCarryWp;

	template <typename T> T *devAlloc(size_t count);
	void devFree(void *p);
	void construct(const FftPlan &plan, long seed);
	void uploadParams();
	void allocateWorkspace();
	template <typename F> void timed(double &acc, F &&f);
};

// error plumbing for the C ABI
struct Error : std::exception {
	std::string msg;
	bool device; // true: a HIP runtime call failed (SMST_ERR_DEVICE); false: a bad argument / state (SMST_ERR_INVALID)
	explicit Error(std::string m, bool deviceError = false) : msg(std::move(m)), device(deviceError) {}
	const char *what() const noexcept override { return msg.c_str(); }
};

} // namespace smst
