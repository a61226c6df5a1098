// C ABI (include/smst.h) over smst::Batch.  Host-memory calls are staged through device buffers owned by the
// handle; device-memory calls go straight to the engine.
#include "../../include/smst.h"
#include "smst_engine.h"
#include <cstdio>
#include <mutex>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

using smst::Batch;

static thread_local std::string g_lastError;

struct smst_batch {
	std::unique_ptr<Batch> engine;
	// host staging (SMST_MEM_HOST)
	float *dIn = nullptr, *dOut = nullptr;
	size_t inCap = 0, outCap = 0;
	long long stagingAllocs = 0;
	~smst_batch() {
		if (engine) hipSetDevice(engine->device());
		if (dIn) hipFree(dIn);
		if (dOut) hipFree(dOut);
	}
};

struct smst_stretch {
	long seed = 0;
	int device = 0;
	std::unique_ptr<smst_batch> batch; // S = 1, created by configure/preset
	// parameters set before configure() survive it, as members of the reference object do
	float transposeFactor = 1, tonalityLimit = 0;
	bool transposeSet = false;
	float formantFactor = 1;
	bool formantComp = false;
	float formantBase = 0;
	std::vector<float> mapTable;
};

#define SMST_TRY try {
#define SMST_CATCH \
	} catch (const smst::Error &e) { g_lastError = e.what(); return e.device ? SMST_ERR_DEVICE : SMST_ERR_INVALID; } \
	catch (const std::exception &e) { g_lastError = e.what(); return SMST_ERR_INVALID; }

static int fail(const char *msg) {
	g_lastError = msg;
	return SMST_ERR_INVALID;
}

static void ensureStage(float *&ptr, size_t &cap, size_t need, int device, long long &allocs) {
	if (need <= cap) return;
	++allocs;
	hipSetDevice(device);
	if (ptr) hipFree(ptr);
	ptr = nullptr;
	cap = 0;
	size_t want = need + need/8 + 1024;
	if (hipMalloc(reinterpret_cast<void **>(&ptr), want*sizeof(float)) != hipSuccess) throw smst::Error("hipMalloc (staging) failed", true);
	cap = want;
}

extern "C" {

const char *smst_last_error(void) { return g_lastError.c_str(); }
void smst_reference_version(int out[3]) { out[0] = 1; out[1] = 3; out[2] = 2; }
int smst_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// ---------------------------------------------------------------------------------------------------------
// batch API
// ---------------------------------------------------------------------------------------------------------
int smst_batch_create_ex(smst_batch **out, int streams, int channels, int block, int interval, int split, int device, long seed, unsigned flags) {
	if (!out) return fail("null output pointer");
	SMST_TRY
	if (flags & ~unsigned(SMST_FLAG_HALF_STATE)) throw smst::Error("unknown creation flag");
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw smst::Error("hipGetDeviceCount: no HIP device available (the gfx950 path has no CPU fallback)", true);
	if (device < 0 || device >= n) throw smst::Error("device ordinal out of range");
	std::unique_ptr<smst_batch> b(new smst_batch());
	b->engine.reset(new Batch(streams, channels, block, interval, split != 0, device, seed, (flags & SMST_FLAG_HALF_STATE) != 0));
	*out = b.release();
	return SMST_OK;
	SMST_CATCH
}
int smst_batch_create(smst_batch **out, int streams, int channels, int block, int interval, int split, int device, long seed) {
	return smst_batch_create_ex(out, streams, channels, block, interval, split, device, seed, 0u);
}
int smst_batch_create_preset_ex(smst_batch **out, int streams, int channels, int preset, float sampleRate, int split, int device, long seed, unsigned flags) {
	// signalsmith-stretch.h:63-68 (float products truncated to int by configure's int parameters)
	if (preset == 0) return smst_batch_create_ex(out, streams, channels, int(sampleRate*0.12), int(sampleRate*0.03), split < 0 ? 0 : split, device, seed, flags);
	if (preset == 1) return smst_batch_create_ex(out, streams, channels, int(sampleRate*0.1), int(sampleRate*0.04), split < 0 ? 1 : split, device, seed, flags);
	return fail("unknown preset");
}
int smst_batch_create_preset(smst_batch **out, int streams, int channels, int preset, float sampleRate, int split, int device, long seed) {
	// signalsmith-stretch.h:63-68 (float products truncated to int by configure's int parameters)
	if (preset == 0) return smst_batch_create(out, streams, channels, int(sampleRate*0.12), int(sampleRate*0.03), split < 0 ? 0 : split, device, seed);
	if (preset == 1) return smst_batch_create(out, streams, channels, int(sampleRate*0.1), int(sampleRate*0.04), split < 0 ? 1 : split, device, seed);
	return fail("unknown preset");
}
void smst_batch_destroy(smst_batch *b) { delete b; }

#define BATCH_Q(name, expr) int name(const smst_batch *b) { if (!b || !b->engine) return fail("null batch"); return (expr); }
BATCH_Q(smst_batch_streams, b->engine->streams())
BATCH_Q(smst_batch_channels, b->engine->channels())
BATCH_Q(smst_batch_block_samples, b->engine->blockSamples())
BATCH_Q(smst_batch_interval_samples, b->engine->intervalSamples())
BATCH_Q(smst_batch_fft_samples, b->engine->fftSamples())
BATCH_Q(smst_batch_bands, b->engine->bands())
BATCH_Q(smst_batch_input_latency, b->engine->inputLatency())
BATCH_Q(smst_batch_output_latency, b->engine->outputLatency())
BATCH_Q(smst_batch_seek_length, b->engine->seekLength())
BATCH_Q(smst_batch_half_state, b->engine->halfPrecisionState() ? 1 : 0)
int smst_batch_output_seek_length(const smst_batch *b, float rate) { if (!b || !b->engine) return fail("null batch"); return b->engine->outputSeekLength(rate); }
long long smst_batch_workspace_bytes(const smst_batch *b) { if (!b || !b->engine) return fail("null batch"); return (long long)b->engine->workspaceBytes(); }

#define BATCH_CALL(body) if (!b || !b->engine) return fail("null batch"); SMST_TRY body; return SMST_OK; SMST_CATCH

int smst_batch_reset(smst_batch *b) { BATCH_CALL(b->engine->reset()) }
int smst_batch_set_transpose_factor(smst_batch *b, int s, float m, float t) { BATCH_CALL(b->engine->setTransposeFactor(s, m, t)) }
int smst_batch_set_transpose_semitones(smst_batch *b, int s, float st, float t) { BATCH_CALL(b->engine->setTransposeSemitones(s, st, t)) }
int smst_batch_set_formant_factor(smst_batch *b, int s, float m, int c) { BATCH_CALL(b->engine->setFormantFactor(s, m, c != 0)) }
int smst_batch_set_formant_semitones(smst_batch *b, int s, float st, int c) { BATCH_CALL(b->engine->setFormantSemitones(s, st, c != 0)) }
int smst_batch_set_formant_base(smst_batch *b, int s, float f) { BATCH_CALL(b->engine->setFormantBase(s, f)) }
int smst_batch_set_freq_map_table(smst_batch *b, int s, const float *table, int n) { BATCH_CALL(b->engine->setFreqMapTable(s, table, n)) }
int smst_batch_synchronize(smst_batch *b) { BATCH_CALL(b->engine->synchronize()) }
void *smst_batch_hip_stream(smst_batch *b) { return (b && b->engine) ? (void *)b->engine->stream() : nullptr; }
int smst_batch_enable_profiling(smst_batch *b, int mode) { BATCH_CALL(b->engine->enableProfiling(mode)) }
int smst_batch_take_timings(smst_batch *b, double ms[8], long long launches[6]) {
	BATCH_CALL({
		smst::BatchTimings t = b->engine->takeTimings();
		ms[0] = t.analyseMs; ms[1] = t.feedMs; ms[2] = t.predictMs; ms[3] = t.chainMs; ms[4] = t.synthMs; ms[5] = t.emitMs; ms[6] = t.otherMs;
		launches[0] = t.analyseLaunches; launches[1] = t.predictLaunches; launches[2] = t.chainLaunches; launches[3] = t.synthLaunches; launches[4] = t.emitLaunches;
		ms[7] = t.chainLiveMs; launches[5] = t.chainLiveLaunches;
	})
}
int smst_batch_take_host_times(smst_batch *b, double ms[3], long long *calls) {
	BATCH_CALL({
		const smst::Batch::HostTimes t = b->engine->takeHostTimes();
		ms[0] = t.callMs; ms[1] = t.waitTablesMs; ms[2] = t.waitGateMs;
		if (calls) *calls = t.calls;
	})
}
int smst_batch_debug_get_state(smst_batch *b, int stream, int which, float *dst) { BATCH_CALL(b->engine->debugGetState(stream, which, dst)) }
int smst_batch_debug_get_carry(smst_batch *b, int stream, float *sums, float *products) { BATCH_CALL(b->engine->debugGetCarry(stream, sums, products)) }
int smst_batch_debug_set_state(smst_batch *b, int stream, int which, const float *src) { BATCH_CALL(b->engine->debugSetState(stream, which, src)) }
int smst_batch_debug_set_carry(smst_batch *b, int stream, const float *sums, const float *products) { BATCH_CALL(b->engine->debugSetCarry(stream, sums, products)) }
long long smst_batch_debug_allocation_events(const smst_batch *b) { if (!b || !b->engine) return fail("null batch"); return (long long)b->engine->allocationEvents() + b->stagingAllocs; }
int smst_batch_wait_for_stream(smst_batch *b, void *hipStream) { BATCH_CALL(b->engine->waitForStream(static_cast<hipStream_t>(hipStream))) }
int smst_batch_signal_stream(smst_batch *b, void *hipStream) { BATCH_CALL(b->engine->signalStream(static_cast<hipStream_t>(hipStream))) }
int smst_debug_complex_selftest(int device, const float *in, float *out, int n) {
	SMST_TRY
	if (n < 1 || !in || !out) throw smst::Error("complex self-test: bad arguments");
	if (hipSetDevice(device) != hipSuccess) throw smst::Error("hipSetDevice failed", true);
	float *dIn = nullptr, *dOut = nullptr;
	if (hipMalloc(reinterpret_cast<void **>(&dIn), (size_t)n*7*sizeof(float)) != hipSuccess) throw smst::Error("hipMalloc failed", true);
	if (hipMalloc(reinterpret_cast<void **>(&dOut), (size_t)n*8*sizeof(float)) != hipSuccess) { hipFree(dIn); throw smst::Error("hipMalloc failed", true); }
	hipError_t e = hipMemcpy(dIn, in, (size_t)n*7*sizeof(float), hipMemcpyHostToDevice);
	if (e == hipSuccess) { smst::launchComplexSelfTest(dIn, dOut, n, nullptr); e = hipGetLastError(); }
	if (e == hipSuccess) e = hipMemcpy(out, dOut, (size_t)n*8*sizeof(float), hipMemcpyDeviceToHost);
	hipFree(dIn);
	hipFree(dOut);
	if (e != hipSuccess) throw smst::Error(std::string("complex self-test: ") + hipGetErrorString(e), true);
	return 0;
	SMST_CATCH
}
long long smst_debug_launch_count(const char *name) { return smst::launchCount(name); }
int smst_batch_debug_get_formants(smst_batch *b, int stream, float *ratio, float *envelope, float *freqEstimate) {
	if (!b || !b->engine) return fail("null batch");
	if (!ratio || !envelope || !freqEstimate) return fail("null destination");
	SMST_TRY
	return b->engine->debugGetFormants(stream, ratio, envelope, freqEstimate) ? 1 : 0;
	SMST_CATCH
}
int smst_batch_debug_get_map(smst_batch *b, int stream, float *dst) {
	if (!b || !b->engine) return fail("null batch");
	SMST_TRY
	return b->engine->debugGetMap(stream, dst) ? 1 : 0;
	SMST_CATCH
}

// host staging: copy the strided host planes into a dense device image [S][C][maxLen]
static const float *stageIn(smst_batch *b, const float *in, long long ss, long long cs, const int *n, int &maxLen) {
	Batch &e = *b->engine;
	const int S = e.streams(), C = e.channels();
	maxLen = 0;
	for (int s = 0; s < S; ++s) maxLen = std::max(maxLen, n[s]);
	if (maxLen == 0) maxLen = 1;
	ensureStage(b->dIn, b->inCap, (size_t)S*C*maxLen, e.device(), b->stagingAllocs);
	hipSetDevice(e.device());
	for (int s = 0; s < S; ++s) {
		for (int c = 0; c < C; ++c) {
			if (n[s] <= 0) continue;
			if (hipMemcpyAsync(b->dIn + ((size_t)s*C + c)*maxLen, in + s*ss + c*cs, (size_t)n[s]*sizeof(float), hipMemcpyHostToDevice, e.stream()) != hipSuccess)
				throw smst::Error("hipMemcpyAsync (H2D) failed", true);
		}
	}
	if (hipStreamSynchronize(e.stream()) != hipSuccess) throw smst::Error("hipStreamSynchronize failed", true);
	return b->dIn;
}
static void unstageOut(smst_batch *b, float *out, long long ss, long long cs, const int *n, int maxLen) {
	Batch &e = *b->engine;
	const int S = e.streams(), C = e.channels();
	hipSetDevice(e.device());
	for (int s = 0; s < S; ++s) {
		for (int c = 0; c < C; ++c) {
			if (n[s] <= 0) continue;
			if (hipMemcpyAsync(out + s*ss + c*cs, b->dOut + ((size_t)s*C + c)*maxLen, (size_t)n[s]*sizeof(float), hipMemcpyDeviceToHost, e.stream()) != hipSuccess)
				throw smst::Error("hipMemcpyAsync (D2H) failed", true);
		}
	}
	if (hipStreamSynchronize(e.stream()) != hipSuccess) throw smst::Error("hipStreamSynchronize failed", true);
}
static int maxOf(const int *n, int S) {
	int m = 0;
	for (int s = 0; s < S; ++s) m = std::max(m, n[s]);
	return std::max(m, 1);
}

int smst_batch_seek(smst_batch *b, const float *in, long long ss, long long cs, const int *inSamples, const double *rates, int memory) {
	BATCH_CALL({
		if (!inSamples) throw smst::Error("null sample counts");
		if (memory == SMST_MEM_DEVICE) {
			b->engine->seek(in, ss, cs, inSamples, rates);
		} else {
			int maxLen;
			const float *dIn = stageIn(b, in, ss, cs, inSamples, maxLen);
			b->engine->seek(dIn, (long long)b->engine->channels()*maxLen, maxLen, inSamples, rates);
			b->engine->synchronize();
		}
	})
}
int smst_batch_process(smst_batch *b, const float *in, long long iss, long long ics, const int *inSamples,
                       float *out, long long oss, long long ocs, const int *outSamples, int memory) {
	BATCH_CALL({
		if (!inSamples || !outSamples) throw smst::Error("null sample counts");
		if (memory == SMST_MEM_DEVICE) {
			b->engine->process(in, iss, ics, inSamples, out, oss, ocs, outSamples);
		} else {
			Batch &e = *b->engine;
			int maxIn;
			const float *dIn = stageIn(b, in, iss, ics, inSamples, maxIn);
			const int maxOut = maxOf(outSamples, e.streams());
			ensureStage(b->dOut, b->outCap, (size_t)e.streams()*e.channels()*maxOut, e.device(), b->stagingAllocs);
			e.process(dIn, (long long)e.channels()*maxIn, maxIn, inSamples, b->dOut, (long long)e.channels()*maxOut, maxOut, outSamples);
			unstageOut(b, out, oss, ocs, outSamples, maxOut);
		}
	})
}
int smst_batch_flush(smst_batch *b, float *out, long long oss, long long ocs, const int *outSamples, const float *rates, int memory) {
	BATCH_CALL({
		if (!outSamples) throw smst::Error("null sample counts");
		Batch &e = *b->engine;
		// a NEGATIVE count leaves that stream out of the flush (flush() of an instance also resets it, :456-463: a count of 0 would do that)
		std::vector<unsigned char> active(e.streams(), 1);
		std::vector<int> counts(outSamples, outSamples + e.streams());
		bool all = true;
		for (int s = 0; s < e.streams(); ++s) if (counts[s] < 0) { active[s] = 0; counts[s] = 0; all = false; }
		const unsigned char *mask = all ? nullptr : active.data();
		if (memory == SMST_MEM_DEVICE) {
			e.flush(out, oss, ocs, counts.data(), rates, mask);
		} else {
			const int maxOut = maxOf(counts.data(), e.streams());
			ensureStage(b->dOut, b->outCap, (size_t)e.streams()*e.channels()*maxOut, e.device(), b->stagingAllocs);
			e.flush(b->dOut, (long long)e.channels()*maxOut, maxOut, counts.data(), rates, mask);
			unstageOut(b, out, oss, ocs, counts.data(), maxOut);
		}
	})
}
int smst_batch_output_seek(smst_batch *b, const float *in, long long ss, long long cs, const int *inputLengths, int memory) {
	BATCH_CALL({
		if (!inputLengths) throw smst::Error("null lengths");
		if (memory == SMST_MEM_DEVICE) {
			b->engine->outputSeek(in, ss, cs, inputLengths);
		} else {
			int maxLen;
			const float *dIn = stageIn(b, in, ss, cs, inputLengths, maxLen);
			b->engine->outputSeek(dIn, (long long)b->engine->channels()*maxLen, maxLen, inputLengths);
			b->engine->synchronize();
		}
	})
}

// ---------------------------------------------------------------------------------------------------------
// single-stream handle API (web/emscripten/main.cpp:15-77 with a handle instead of the global singleton)
// ---------------------------------------------------------------------------------------------------------
static std::string g_defaultDeviceError; // set once by smst_default_device() when SMST_DEVICE names no device of this process
int smst_create(smst_stretch **out, long seed, int device) {
	if (!out) return fail("null output pointer");
	SMST_TRY
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw smst::Error("hipGetDeviceCount: no HIP device available (the gfx950 path has no CPU fallback)", true);
	// (the SMST_DEVICE message only for the sentinel smst_default_device() returns when the variable named no device: an explicit ordinal
	// that is out of range has nothing to do with the default device)
	if (device == -1 && !g_defaultDeviceError.empty()) throw smst::Error(g_defaultDeviceError);
	if (device < 0 || device >= n) throw smst::Error("device ordinal " + std::to_string(device) + " out of range: this process sees " + std::to_string(n) + " device(s)");
	smst_stretch *h = new smst_stretch();
	h->seed = seed;
	h->device = device;
	*out = h;
	return SMST_OK;
	SMST_CATCH
}
void smst_destroy(smst_stretch *h) { delete h; }

// the device new single-stream handles are created on: SMST_DEVICE (validated against the device count, once) or the setter
static std::atomic<int> g_defaultDevice{-1};
// SMST_DEVICE that is no ordinal this process can see is an ERROR (one rank per GPU: a wrong ordinal or a mismatch with HIP_VISIBLE_DEVICES
// would put every rank on GPU 0 without a word): smst_default_device() reports it on stderr once and returns -1, and smst_create() on the
// default device fails with SMST_ERR_INVALID naming the value and the device count.
int smst_default_device(void) {
	int v = g_defaultDevice.load(std::memory_order_acquire);
	if (v == -1) {
		v = 0;
		if (const char *env = std::getenv("SMST_DEVICE")) {
			char *end = nullptr;
			const long asked = std::strtol(env, &end, 10);
			const int n = smst_device_count();
			if (end != env && *end == '\0' && asked >= 0 && (n <= 0 || asked < n)) {
				v = int(asked);
			} else {
				static std::once_flag once;
				std::call_once(once, [&] {
					g_defaultDeviceError = std::string("SMST_DEVICE=\"") + env + "\" is not a device ordinal of this process (" + std::to_string(n) + " device(s) visible)";
					std::fprintf(stderr, "libsmst_hip: %s\n", g_defaultDeviceError.c_str());
				});
				v = -2; // latched: invalid
			}
		}
		int expected = -1;
		if (!g_defaultDevice.compare_exchange_strong(expected, v, std::memory_order_acq_rel)) v = expected; // another thread (or the setter) was first
	}
	if (v == -2) { fail(g_defaultDeviceError.c_str()); return -1; }
	return v;
}
int smst_set_default_device(int device) {
	if (device < 0 || device >= smst_device_count()) return fail("device ordinal out of range");
	g_defaultDevice.store(device, std::memory_order_release);
	return SMST_OK;
}
int smst_clone(smst_stretch **out, const smst_stretch *src) {
	if (!out || !src) return fail("null pointer");
	SMST_TRY
	std::unique_ptr<smst_stretch> h(new smst_stretch());
	h->seed = src->seed; h->device = src->device;
	h->transposeFactor = src->transposeFactor; h->tonalityLimit = src->tonalityLimit; h->transposeSet = src->transposeSet;
	h->formantFactor = src->formantFactor; h->formantComp = src->formantComp; h->formantBase = src->formantBase;
	h->mapTable = src->mapTable;
	if (src->batch) {
		Batch &e = *src->batch->engine;
		std::unique_ptr<smst_batch> b(new smst_batch());
		b->engine.reset(new Batch(1, e.channels(), e.blockSamples(), e.intervalSamples(), e.splitComputation(), src->device, src->seed, e.halfPrecisionState()));
		b->engine->copyStateFrom(e);
		h->batch = std::move(b);
	}
	*out = h.release();
	return SMST_OK;
	SMST_CATCH
}

static void applyParams(smst_stretch *h) {
	Batch &e = *h->batch->engine;
	if (h->transposeSet) e.setTransposeFactor(0, h->transposeFactor, h->tonalityLimit);
	e.setFormantFactor(0, h->formantFactor, h->formantComp);
	e.setFormantBase(0, h->formantBase);
	if (!h->mapTable.empty()) e.setFreqMapTable(0, h->mapTable.data(), int(h->mapTable.size()));
}
int smst_configure(smst_stretch *h, int channels, int block, int interval, int split) {
	if (!h) return fail("null handle");
	SMST_TRY
	std::unique_ptr<smst_batch> b(new smst_batch());
	b->engine.reset(new Batch(1, channels, block, interval, split != 0, h->device, h->seed));
	if (h->batch) b->engine->inheritAcrossConfigure(*h->batch->engine); // configure() does not reseed the engine (:38-39, :71-94)
	h->batch = std::move(b);
	applyParams(h);
	return SMST_OK;
	SMST_CATCH
}
int smst_preset_default(smst_stretch *h, int channels, float sampleRate, int split) {
	return smst_configure(h, channels, int(sampleRate*0.12), int(sampleRate*0.03), split < 0 ? 0 : split);
}
int smst_preset_cheaper(smst_stretch *h, int channels, float sampleRate, int split) {
	return smst_configure(h, channels, int(sampleRate*0.1), int(sampleRate*0.04), split < 0 ? 1 : split);
}

#define STRETCH_Q(name, expr) int name(const smst_stretch *h) { if (!h || !h->batch) return fail("unconfigured handle"); const Batch &e = *h->batch->engine; return (expr); }
STRETCH_Q(smst_block_samples, e.blockSamples())
STRETCH_Q(smst_interval_samples, e.intervalSamples())
STRETCH_Q(smst_input_latency, e.inputLatency())
STRETCH_Q(smst_output_latency, e.outputLatency())
STRETCH_Q(smst_split_computation, e.splitComputation() ? 1 : 0)
STRETCH_Q(smst_block_steps, e.lastBlockSteps(0))
STRETCH_Q(smst_blocks_started, e.lastCallBlocks(0))
STRETCH_Q(smst_seek_length, e.seekLength())
int smst_output_seek_length(const smst_stretch *h, float rate) { if (!h || !h->batch) return fail("unconfigured handle"); return h->batch->engine->outputSeekLength(rate); }

#define STRETCH_CALL(body) if (!h) return fail("null handle"); SMST_TRY body; return SMST_OK; SMST_CATCH

int smst_reset(smst_stretch *h) { STRETCH_CALL(if (h->batch) h->batch->engine->reset()) }
int smst_set_transpose_factor(smst_stretch *h, float m, float t) {
	STRETCH_CALL({
		h->transposeFactor = m; h->tonalityLimit = t; h->transposeSet = true; h->mapTable.clear();
		if (h->batch) h->batch->engine->setTransposeFactor(0, m, t);
	})
}
int smst_set_transpose_semitones(smst_stretch *h, float st, float t) { return smst_set_transpose_factor(h, float(std::pow(2, st/12)), t); }
int smst_set_formant_factor(smst_stretch *h, float m, int comp) {
	STRETCH_CALL({
		h->formantFactor = m; h->formantComp = comp != 0;
		if (h->batch) h->batch->engine->setFormantFactor(0, m, comp != 0);
	})
}
int smst_set_formant_semitones(smst_stretch *h, float st, int comp) { return smst_set_formant_factor(h, float(std::pow(2, st/12)), comp); }
int smst_set_formant_base(smst_stretch *h, float f) {
	STRETCH_CALL({
		h->formantBase = f;
		if (h->batch) h->batch->engine->setFormantBase(0, f);
	})
}
int smst_set_freq_map_table(smst_stretch *h, const float *table, int n) {
	STRETCH_CALL({
		if (table && n > 0) h->mapTable.assign(table, table + n); else h->mapTable.clear();
		if (h->batch) h->batch->engine->setFreqMapTable(0, table, n);
	})
}

// planar pointer arrays -> dense host image [C][n] (the reference indexes buffers[c][i], README.md:46)
static void gatherPlanes(const float *const *planes, int C, int n, std::vector<float> &dense) {
	dense.resize((size_t)C*std::max(n, 1));
	for (int c = 0; c < C; ++c) if (n > 0) std::copy(planes[c], planes[c] + n, dense.begin() + (size_t)c*n);
}
static void scatterPlanes(const std::vector<float> &dense, float *const *planes, int C, int n) {
	for (int c = 0; c < C; ++c) if (n > 0) std::copy(dense.begin() + (size_t)c*n, dense.begin() + (size_t)(c + 1)*n, planes[c]);
}

int smst_seek(smst_stretch *h, const float *const *inputs, int inputSamples, double playbackRate) {
	if (!h || !h->batch) return fail("unconfigured handle");
	std::vector<float> in;
	gatherPlanes(inputs, h->batch->engine->channels(), inputSamples, in);
	return smst_batch_seek(h->batch.get(), in.data(), 0, std::max(inputSamples, 1), &inputSamples, &playbackRate, SMST_MEM_HOST);
}
int smst_process(smst_stretch *h, const float *const *inputs, int inputSamples, float *const *outputs, int outputSamples) {
	if (!h || !h->batch) return fail("unconfigured handle");
	const int C = h->batch->engine->channels();
	std::vector<float> in, out((size_t)C*std::max(outputSamples, 1));
	gatherPlanes(inputs, C, inputSamples, in);
	int rc = smst_batch_process(h->batch.get(), in.data(), 0, std::max(inputSamples, 1), &inputSamples, out.data(), 0, std::max(outputSamples, 1), &outputSamples, SMST_MEM_HOST);
	if (rc == SMST_OK) scatterPlanes(out, outputs, C, outputSamples);
	return rc;
}
int smst_flush(smst_stretch *h, float *const *outputs, int outputSamples, float playbackRate) {
	if (!h || !h->batch) return fail("unconfigured handle");
	const int C = h->batch->engine->channels();
	std::vector<float> out((size_t)C*std::max(outputSamples, 1));
	int rc = smst_batch_flush(h->batch.get(), out.data(), 0, std::max(outputSamples, 1), &outputSamples, &playbackRate, SMST_MEM_HOST);
	if (rc == SMST_OK) scatterPlanes(out, outputs, C, outputSamples);
	return rc;
}
int smst_output_seek(smst_stretch *h, const float *const *inputs, int inputLength) {
	if (!h || !h->batch) return fail("unconfigured handle");
	std::vector<float> in;
	gatherPlanes(inputs, h->batch->engine->channels(), inputLength, in);
	return smst_batch_output_seek(h->batch.get(), in.data(), 0, std::max(inputLength, 1), &inputLength, SMST_MEM_HOST);
}
int smst_exact(smst_stretch *h, const float *const *inputs, int inputSamples, float *const *outputs, int outputSamples) {
	// signalsmith-stretch.h:468-491
	if (!h || !h->batch) return fail("unconfigured handle");
	Batch &e = *h->batch->engine;
	const int C = e.channels();
	float playbackRate = inputSamples/float(outputSamples);
	int seekLength = e.outputSeekLength(playbackRate);
	if (inputSamples < seekLength) {
		for (int c = 0; c < C; ++c) std::fill(outputs[c], outputs[c] + outputSamples, 0.0f);
		g_lastError = "exact(): input shorter than outputSeekLength";
		return SMST_ERR_SHORT;
	}
	int rc = smst_output_seek(h, inputs, seekLength);
	if (rc != SMST_OK) return rc;
	int outputIndex = int(outputSamples - seekLength/playbackRate);
	std::vector<const float *> inOff(C);
	std::vector<float *> outOff(C);
	for (int c = 0; c < C; ++c) inOff[c] = inputs[c] + seekLength;
	rc = smst_process(h, inOff.data(), inputSamples - seekLength, outputs, outputIndex);
	if (rc != SMST_OK) return rc;
	for (int c = 0; c < C; ++c) outOff[c] = outputs[c] + outputIndex;
	return smst_flush(h, outOff.data(), outputSamples - outputIndex, playbackRate);
}

} // extern "C"
