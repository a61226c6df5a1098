/* smst.h -- C ABI of the MI355X (gfx950) implementation of the Signalsmith Stretch spectral hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Library: libsmst_hip.so.
 *
 * Two groups of entry points:
 *
 *  (1) single-stream handle API -- one-to-one with the flat ABI the reference itself ships for its WASM
 *      build (reference: web/emscripten/main.cpp:15-77, 17 functions over a global singleton; here the
 *      singleton becomes a handle) plus the three members that ABI lacks (outputSeek/exact: reference
 *      signalsmith-stretch.h:173-207,468-491; setFreqMap in table form: :120-122).  Sample buffers are HOST
 *      pointers, planar (`buffers[channel][index]`, reference README.md:46).
 *
 *  (2) batch API -- S independent streams that share one configuration, processed together on one GPU.
 *      This is the data-parallel axis the reference does not have (one SignalsmithStretch instance per
 *      stream, signalsmith-stretch.h:34-35); each stream behaves exactly like one reference instance.
 *      Buffers are planar with explicit strides: sample (s, c, i) at base[s*streamStride + c*channelStride + i].
 *      `memory` selects SMST_MEM_HOST (staged through the library's own device buffers) or SMST_MEM_DEVICE
 *      (pointers are device pointers on the batch's GPU; the call is asynchronous on the batch's stream
 *      except for one 64-byte-per-stream readback inside process -- use smst_batch_synchronize()).
 *      Device inputs must be COMPLETE when the call is made, or the producing stream must be handed to
 *      smst_batch_wait_for_stream() first; inputs / outputs must stay valid until smst_batch_synchronize() returns (or
 *      until a stream passed to smst_batch_signal_stream() has caught up); the
 *      host-side part of a call (silence gate, block scheduler) overlaps the kernels of the previous call.
 *
 * Limits the reference does not have (signalsmith-stretch.h:71-94 accepts any channel count and block size); configure / create
 * return SMST_ERR_INVALID with the limit in smst_last_error() beyond them:
 *   - 1 ... 16 channels per stream (the recurrence kernels size their per-lane channel arrays at compile time: 1-2 channels and 3-8 channels
 *     keep the per-bin records in LDS, 9-16 channels take the un-fused kernel pair with records through HBM -- slower, the same arithmetic);
 *   - fftSamples/2 = 2^k * {1, 3, 5} bands (the reference's own fast sizes), at most 19200: one FFT buffer of bands*8 bytes has to fit a
 *     CU's LDS.  Up to 9600 bands (every preset up to 96 kHz: presetDefault there has 6144) both ping-pong buffers do; beyond that --
 *     the presets at 176.4 / 192 kHz: 10240 / 12288 bands -- the second buffer lives in memory (slower per frame, the same arithmetic);
 *   - interval >= fftSamples/62 (the vertical step of the phase prediction, round(fftSamples/interval), has to fit the wavefront's
 *     skew); interval <= block.
 *   - Sample = float arithmetic only (the C++ drop-in accepts double buffers and converts at the boundary).
 *
 * Hardware queues: a call is pipelined over three HIP streams; in a process with streams of its own (an RCCL communicator is
 * enough) set GPU_MAX_HW_QUEUES=8 before the HIP runtime starts, or two of them may share a queue and run in submission order
 * (measured: +12 % per step).
 *
 * Every function returns 0 on success and a negative code on failure (the reference has no error channel:
 * signalsmith-stretch.h is UB when unconfigured; only exact() reports, :471-480).  smst_last_error()
 * returns the message of the last failure on the calling thread.
 */
#ifndef SMST_H
#define SMST_H

#ifdef __cplusplus
extern "C" {
#endif

#define SMST_OK 0
#define SMST_ERR_INVALID (-1)  /* bad argument / unconfigured handle */
#define SMST_ERR_DEVICE (-2)   /* HIP runtime error (no GPU, out of memory, launch failure) */
#define SMST_ERR_SHORT (-3)    /* exact(): input shorter than outputSeekLength (signalsmith-stretch.h:471-480) */

/* creation flags (smst_batch_create_ex / smst_batch_create_preset_ex) */
#define SMST_FLAG_HALF_STATE 1u /* BASELINE config 5 "fp16 internal": the state that outlives a tile -- Band.output (as half2),
                                 * Prediction.energy (as the half of its square root) and the overlap-add partial sums (as half) -- is
                                 * STORED in fp16; every computation stays fp32.  The reference has no such mode (Sample is float or
                                 * double, signalsmith-stretch.h:34): results agree with it in the magnitude domain to ~1e-3 and are
                                 * no longer bit-identical across different chunkings of the same audio. */

#define SMST_MEM_HOST 0
#define SMST_MEM_DEVICE 1

const char *smst_last_error(void);
/* version of the reference API this library mirrors: {1,3,2} (signalsmith-stretch.h:36) */
void smst_reference_version(int out[3]);
int smst_device_count(void);

/* ---------------------------------------------------------------------------------------------------------
 * (1) single-stream handle API
 * ------------------------------------------------------------------------------------------------------- */
typedef struct smst_stretch smst_stretch;

/* SignalsmithStretch() / SignalsmithStretch(long seed): signalsmith-stretch.h:38-39.  device = HIP ordinal.
 * seed: the reference seeds its std::default_random_engine with it (:39, :616) and draws the per-bin time factors of stretches beyond
 * 2x from that engine (:639-640).  This library carries the same engine -- libstdc++'s (minstd_rand0 through
 * uniform_real_distribution<float>), i.e. the one a g++ build of the reference has -- so an instance created with seed S makes the
 * draws a reference instance constructed with S makes.  Stream s of a batch is the instance of seed + s. */
int smst_create(smst_stretch **out, long seed, int device);
void smst_destroy(smst_stretch *h);
/* The reference object is a plain struct and therefore COPYABLE (signalsmith-stretch.h:34-35; e.g. a std::vector of them):
 * a new handle with the configuration, every parameter and the complete processing state of `src` (input history,
 * Band.input/.prevInput/.output, Prediction.energy, overlap-add ring, scheduler state) -- both continue identically from
 * here, independently of each other.  An unconfigured `src` gives an unconfigured copy. */
int smst_clone(smst_stretch **out, const smst_stretch *src);
/* The device new single-stream objects of the C++ drop-in header are created on (the reference has no such notion): the
 * environment variable SMST_DEVICE at first use, 0 if unset, or whatever smst_set_default_device() was given last.
 * An SMST_DEVICE that is not an ordinal this process can see (not a number, or >= smst_device_count()) is an ERROR, not device 0: the
 * function reports it on stderr once and returns -1 from then on, and smst_create(..., -1) fails with SMST_ERR_INVALID and a message
 * that names the value and the device count (one rank per GPU: a silent fall-back would put every rank on GPU 0). */
int smst_default_device(void);
int smst_set_default_device(int device);

/* presetDefault / presetCheaper / configure: signalsmith-stretch.h:63-94; web/emscripten/main.cpp:43-51.
 * split: 0/1, or -1 for the preset's own default (false for default, true for cheaper). */
int smst_preset_default(smst_stretch *h, int channels, float sampleRate, int split);
int smst_preset_cheaper(smst_stretch *h, int channels, float sampleRate, int split);
int smst_configure(smst_stretch *h, int channels, int blockSamples, int intervalSamples, int split);

/* queries: signalsmith-stretch.h:42-47,96-104,166-168,205-207; main.cpp:28-39 */
int smst_block_samples(const smst_stretch *h);
int smst_interval_samples(const smst_stretch *h);
int smst_input_latency(const smst_stretch *h);
int smst_output_latency(const smst_stretch *h);
int smst_split_computation(const smst_stretch *h);
/* Number of processing steps of the newest block (blockProcess.steps, signalsmith-stretch.h:284-318: analysis, spectral processing and
 * synthesis steps as the reference counts them for this block's flags and channel count); 0 before the first block.  The C++ drop-in
 * header reports it through the reference's SIGNALSMITH_STRETCH_PROFILE_PROCESS_STEP(step, steps) hook (:329-331). */
int smst_block_steps(const smst_stretch *h);
/* Number of blocks that BEGAN in the most recent smst_process() call (the `blockProcess.samplesSinceLast >= interval` branch,
 * signalsmith-stretch.h:281, taken that many times); 0 for a call that only emitted samples of a block already under way.  The drop-in
 * header announces steps through the reference's profiling hooks only for calls in which a block began. */
int smst_blocks_started(const smst_stretch *h);
int smst_seek_length(const smst_stretch *h);
int smst_output_seek_length(const smst_stretch *h, float playbackRate);

/* reset: signalsmith-stretch.h:49-60; main.cpp:40-42 */
int smst_reset(smst_stretch *h);

/* parameters: signalsmith-stretch.h:107-135; main.cpp:52-66.  Frequencies are relative to the sample rate. */
int smst_set_transpose_factor(smst_stretch *h, float multiplier, float tonalityLimit);
int smst_set_transpose_semitones(smst_stretch *h, float semitones, float tonalityLimit);
int smst_set_formant_factor(smst_stretch *h, float multiplier, int compensatePitch);
int smst_set_formant_semitones(smst_stretch *h, float semitones, int compensatePitch);
int smst_set_formant_base(smst_stretch *h, float baseFreq);
/* setFreqMap (signalsmith-stretch.h:120-122) in table form: table[i] = map((i + 0.5)/(2n)), linear in between
 * and beyond; n = 0 removes the map.  In a batch every stream keeps its own table, knot for knot, whatever the lengths of
 * the other streams' tables (no call on one stream changes another stream's map). */
int smst_set_freq_map_table(smst_stretch *h, const float *table, int n);

/* seek / process / flush: signalsmith-stretch.h:140-165, 210-423, 427-464; main.cpp:68-76 */
int smst_seek(smst_stretch *h, const float *const *inputs, int inputSamples, double playbackRate);
int smst_process(smst_stretch *h, const float *const *inputs, int inputSamples, float *const *outputs, int outputSamples);
int smst_flush(smst_stretch *h, float *const *outputs, int outputSamples, float playbackRate);
/* outputSeek / exact: signalsmith-stretch.h:173-204, 468-491 */
int smst_output_seek(smst_stretch *h, const float *const *inputs, int inputLength);
int smst_exact(smst_stretch *h, const float *const *inputs, int inputSamples, float *const *outputs, int outputSamples);

/* ---------------------------------------------------------------------------------------------------------
 * (2) batch API
 * ------------------------------------------------------------------------------------------------------- */
typedef struct smst_batch smst_batch;

int smst_batch_create(smst_batch **out, int streams, int channels, int blockSamples, int intervalSamples,
                      int split, int device, long seed);
/* preset: 0 = presetDefault (block = 0.12 sr, interval = 0.03 sr), 1 = presetCheaper (0.1 sr, 0.04 sr) */
int smst_batch_create_preset(smst_batch **out, int streams, int channels, int preset, float sampleRate,
                             int split, int device, long seed);
int smst_batch_create_ex(smst_batch **out, int streams, int channels, int blockSamples, int intervalSamples,
                         int split, int device, long seed, unsigned flags);
int smst_batch_create_preset_ex(smst_batch **out, int streams, int channels, int preset, float sampleRate,
                                int split, int device, long seed, unsigned flags);
void smst_batch_destroy(smst_batch *b);

int smst_batch_streams(const smst_batch *b);
int smst_batch_channels(const smst_batch *b);
int smst_batch_block_samples(const smst_batch *b);
int smst_batch_interval_samples(const smst_batch *b);
int smst_batch_fft_samples(const smst_batch *b);
int smst_batch_bands(const smst_batch *b);
int smst_batch_input_latency(const smst_batch *b);
int smst_batch_output_latency(const smst_batch *b);
int smst_batch_seek_length(const smst_batch *b);
int smst_batch_half_state(const smst_batch *b); /* 1 if created with SMST_FLAG_HALF_STATE */
int smst_batch_output_seek_length(const smst_batch *b, float playbackRate);
long long smst_batch_workspace_bytes(const smst_batch *b);

int smst_batch_reset(smst_batch *b);
/* stream = -1 applies to every stream */
int smst_batch_set_transpose_factor(smst_batch *b, int stream, float multiplier, float tonalityLimit);
int smst_batch_set_transpose_semitones(smst_batch *b, int stream, float semitones, float tonalityLimit);
int smst_batch_set_formant_factor(smst_batch *b, int stream, float multiplier, int compensatePitch);
int smst_batch_set_formant_semitones(smst_batch *b, int stream, float semitones, int compensatePitch);
int smst_batch_set_formant_base(smst_batch *b, int stream, float baseFreq);
int smst_batch_set_freq_map_table(smst_batch *b, int stream, const float *table, int n);

/* inSamples / outSamples / rates / lengths: HOST arrays with one entry per stream. */
int smst_batch_seek(smst_batch *b, const float *in, long long inStreamStride, long long inChannelStride,
                    const int *inSamples, const double *playbackRates, int memory);
int smst_batch_process(smst_batch *b, const float *in, long long inStreamStride, long long inChannelStride,
                       const int *inSamples, float *out, long long outStreamStride, long long outChannelStride,
                       const int *outSamples, int memory);
/* flush(): per stream as signalsmith-stretch.h:427-464 (the stream's output ring is read out and the stream starts afresh).  A NEGATIVE
 * outSamples[s] leaves stream s out of the call altogether -- a count of 0 still resets it, as flush(outputs, 0) of an instance does. */
int smst_batch_flush(smst_batch *b, float *out, long long outStreamStride, long long outChannelStride,
                     const int *outSamples, const float *playbackRates, int memory);
int smst_batch_output_seek(smst_batch *b, const float *in, long long inStreamStride, long long inChannelStride,
                           const int *inputLengths, int memory);
int smst_batch_synchronize(smst_batch *b);
/* raw hipStream_t the batch enqueues on (so callers can order their own device work against it) */
void *smst_batch_hip_stream(smst_batch *b);
/* Stream ordering for device-memory callers, without a host synchronisation:
 * wait_for_stream: everything the batch enqueues from now on runs after the work ALREADY enqueued on `hipStream` (the
 *                  caller's producer of the input tensors); call it before smst_batch_process / _seek / _output_seek.
 * signal_stream:   work enqueued on `hipStream` from now on runs after everything the batch has enqueued so far (so a
 *                  consumer of the outputs need not call smst_batch_synchronize). */
int smst_batch_wait_for_stream(smst_batch *b, void *hipStream);
int smst_batch_signal_stream(smst_batch *b, void *hipStream);

/* measurement hooks: per-kernel-class device time (hipEvent pairs on the batch's stream) accumulated since the
 * last call.  ms[0..6] = analyse, feed, predict, chain, synth, emit, other; launches[0..4] = analyse, predict,
 * chain, synth, emit. */
/* mode 1: HIP-event pairs around every kernel class, tiles serialised (ms[0..6]: analyse, feed, predict, recurrence, synth,
 * emit, other; launches[0..4]: analyse, predict, recurrence, synth, emit).  mode 2: only the recurrence kernel, timed in
 * place on its own stream while the other streams keep overlapping it (ms[7], launches[5]).  0: off. */
int smst_batch_enable_profiling(smst_batch *b, int mode);
int smst_batch_take_timings(smst_batch *b, double ms[8], long long launches[6]);
/* Host time of smst_batch_process since the last take (always counted, no mode): ms[0] wall time inside the calls, of which ms[1] was spent
 * waiting for the call before the previous one to finish (two sets of per-call tables: the host runs at most two calls ahead of the device --
 * back-pressure, not work) and ms[2] waiting for the silence gate's readback; ms[0] - ms[1] - ms[2] is the host's own work (block scheduler,
 * table fills, enqueues).  *calls (may be null): the number of calls.  One host thread per batch: its work must stay below the device's step. */
int smst_batch_take_host_times(smst_batch *b, double ms[3], long long *calls);

/* test hooks (tests/ only): per-stream state rows.  which: 0 Band.input, 1 Band.prevInput, 2 Band.output
 * (interleaved re,im: 2*channels*bands floats), 3 Prediction.energy (channels*bands floats). */
int smst_batch_debug_get_state(smst_batch *b, int stream, int which, float *dst);
int smst_batch_debug_get_carry(smst_batch *b, int stream, float *sums, float *products);
/* teacher forcing (SURVEY.md App. D.2 i): overwrite one stream's carried per-bin state (same selectors / layouts as
 * the getters) or its overlap-add carry ([channels][block+interval] sums, [block+interval] window products, index 0 =
 * the next output sample).  The batch is synchronised first. */
int smst_batch_debug_set_state(smst_batch *b, int stream, int which, const float *src);
int smst_batch_debug_set_carry(smst_batch *b, int stream, const float *sums, const float *products);
/* number of device / pinned allocations and host-table growth events since the batch was created.  process() is
 * allocation-free in steady state (the reference asserts the same of itself: cmd/main-dev.cpp:158-163, "allocated during
 * process()"): tests/test_abi.py::test_process_does_not_allocate_in_steady_state checks that this number stands still. */
long long smst_batch_debug_allocation_events(const smst_batch *b);
/* output map of the stream's newest hop (2*bands floats: inputBin, freqGrad per bin; signalsmith-stretch.h:587-590,
 * :882-917).  Returns 1 if that hop had a frequency map, 0 if not (dst untouched), negative on error. */
int smst_batch_debug_get_map(smst_batch *b, int stream, float *dst);
/* the formant stage of the stream's newest hop (signalsmith-stretch.h:972-1036): ratio[bands] = the per-bin energy ratio applied to every channel's
 * inputEnergy (:1026-1033), envelope[bands] = formantMetric after its eight max-decay / min-grow passes (:984-1006), *freqEstimate = the pitch
 * estimate in bins the envelope was built with (:980-981, :929-966).  Available only in a batch created with SMST_NO_FEED_FUSION=1 (the separate
 * envelope kernel writes them; the default form keeps them in LDS and is bit-identical to it: test_feed_fusion_equals_separate).  Returns 1, or 0
 * if that hop had no formant processing / the batch runs the fused kernels (destinations untouched), negative on error. */
int smst_batch_debug_get_formants(smst_batch *b, int stream, float *ratio, float *envelope, float *freqEstimate);
/* the kernels' packed complex helpers (csrc/smst_complex.h, gfx950 inline assembly) evaluated on the device: in = n x
 * (a.re, a.im, b.re, b.im, c.re, c.im, fraction), out = n x (a*b, a*conj(b), a*b + c, a + (b - a)*fraction) as 8 floats. */
int smst_debug_complex_selftest(int device, const float *in, float *out, int n);
/* launches, since the library was loaded, of one kernel variant: "vocoder_aligned", "vocoder_staged", "vocoder_gather",
 * "vocoder_n", "vocoder_one", "vocoder_across", "chain_unfused", "analyse_teams", "analyse_fast", "analyse_generic",
 * "synth_teams", "synth_fast", "synth_generic" (-1: unknown name).  The "this form is bit-identical to that form" tests
 * assert through it that both forms really ran. */
long long smst_debug_launch_count(const char *name);

#ifdef __cplusplus
}
#endif
#endif /* SMST_H */
