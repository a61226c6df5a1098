// Drop-in C++ front-end for the MI355X implementation: same class name, namespace and member signatures as the
// reference header (reference: signalsmith-stretch.h:34-491; include path as include/signalsmith-stretch/
// signalsmith-stretch.h:1), forwarding every call to the C ABI in include/smst.h (libsmst_hip.so).
// Existing callers (e.g. the reference's cmd/main.cpp:44-82) compile unchanged against this header.
//
// Differences a caller can observe are listed in DESIGN.md ("deviations"): the arithmetic is fp32 whatever Sample is; setFreqMap
// samples the std::function into a 4096-point table; a RandomEngine template argument is accepted and ignored:
// the >2x-stretch randomisation always uses the DEFAULT engine of a g++ build of the reference (libstdc++'s minstd_rand0), seeded as
// the reference seeds it -- same seed, same draws.
#ifndef SIGNALSMITH_STRETCH_H
#define SIGNALSMITH_STRETCH_H

#include <cmath>
#include <cstddef>
#include <functional>
#include <random>
#include <stdexcept>
#include <type_traits>
#include <utility>
#include <vector>

#include "../smst.h"

namespace signalsmith { namespace stretch {

template <typename Sample = float, class RandomEngine = void>
struct SignalsmithStretch {
	// Sample: the type of the caller's buffers and parameters, as in the reference (:34).  The device computes in fp32 whatever it is:
	// SignalsmithStretch<double> compiles and runs -- samples and parameters are converted at this boundary -- and gives the
	// float instantiation's results; it does NOT give the reference's double-precision ones.
	static_assert(std::is_floating_point<Sample>::value, "Sample is float or double (the gfx950 implementation computes in fp32)");
	static constexpr size_t version[3] = {1, 3, 2};

	SignalsmithStretch() : SignalsmithStretch(long(std::random_device{}())) {}
	SignalsmithStretch(long seed) : seed(seed) {
		if (smst_create(&handle, seed, defaultDevice()) != SMST_OK) throw std::runtime_error(smst_last_error());
	}
	~SignalsmithStretch() { smst_destroy(handle); }
	// The reference is a plain struct: copyable (the copy carries the whole processing state and continues independently) and
	// movable.  Here a copy is smst_clone -- a second set of device buffers -- and a move hands the handle over.  A MOVED-FROM
	// object stays a valid one, as the reference's does: its next use creates a fresh, unconfigured instance (h()), and copying
	// from it copies that.
	SignalsmithStretch(const SignalsmithStretch &other) : channels(other.channels), seed(other.seed) {
		if (smst_clone(&handle, other.h()) != SMST_OK) throw std::runtime_error(smst_last_error());
	}
	SignalsmithStretch &operator=(const SignalsmithStretch &other) {
		if (this != &other) {
			smst_stretch *copy = nullptr;
			if (smst_clone(&copy, other.h()) != SMST_OK) throw std::runtime_error(smst_last_error());
			smst_destroy(handle);
			handle = copy;
			channels = other.channels;
			seed = other.seed;
		}
		return *this;
	}
	SignalsmithStretch(SignalsmithStretch &&other) noexcept : handle(other.handle), channels(other.channels), seed(other.seed),
		inPlanar(std::move(other.inPlanar)), outPlanar(std::move(other.outPlanar)), inPtrs(std::move(other.inPtrs)), outPtrs(std::move(other.outPtrs)) {
		other.handle = nullptr;
		other.channels = 0;
	}
	SignalsmithStretch &operator=(SignalsmithStretch &&other) noexcept {
		if (this != &other) {
			smst_destroy(handle);
			handle = other.handle;
			channels = other.channels;
			seed = other.seed;
			other.handle = nullptr;
			other.channels = 0;
		}
		return *this;
	}
	// the GPU new objects are created on (no counterpart in the reference): SMST_DEVICE in the environment, or this setter
	static void setDefaultDevice(int device) {
		if (smst_set_default_device(device) != SMST_OK) throw std::runtime_error(smst_last_error());
	}

	int inputLatency() const { return smst_input_latency(h()); }
	int outputLatency() const { return smst_output_latency(h()); }
	void reset() { check(smst_reset(h())); }

	void presetDefault(int nChannels, Sample sampleRate, bool splitComputation = false) {
		channels = nChannels;
		check(smst_preset_default(h(), nChannels, sampleRate, splitComputation));
	}
	void presetCheaper(int nChannels, Sample sampleRate, bool splitComputation = true) {
		channels = nChannels;
		check(smst_preset_cheaper(h(), nChannels, sampleRate, splitComputation));
	}
	void configure(int nChannels, int blockSamples, int intervalSamples, bool splitComputation = false) {
		channels = nChannels;
		check(smst_configure(h(), nChannels, blockSamples, intervalSamples, splitComputation));
	}
	int blockSamples() const { return smst_block_samples(h()); }
	int intervalSamples() const { return smst_interval_samples(h()); }
	bool splitComputation() const { return smst_split_computation(h()) != 0; }

	void setTransposeFactor(Sample multiplier, Sample tonalityLimit = 0) { check(smst_set_transpose_factor(h(), multiplier, tonalityLimit)); }
	void setTransposeSemitones(Sample semitones, Sample tonalityLimit = 0) { check(smst_set_transpose_semitones(h(), semitones, tonalityLimit)); }
	void setFreqMap(std::function<Sample(Sample)> inputToOutput) {
		if (!inputToOutput) { check(smst_set_freq_map_table(h(), nullptr, 0)); return; }
		const int n = 4096;
		std::vector<float> table(n);
		for (int i = 0; i < n; ++i) table[i] = inputToOutput((i + 0.5f)/(2*n));
		check(smst_set_freq_map_table(h(), table.data(), n));
	}
	void setFormantFactor(Sample multiplier, bool compensatePitch = false) { check(smst_set_formant_factor(h(), multiplier, compensatePitch)); }
	void setFormantSemitones(Sample semitones, bool compensatePitch = false) { check(smst_set_formant_semitones(h(), semitones, compensatePitch)); }
	void setFormantBase(Sample baseFreq = 0) { check(smst_set_formant_base(h(), baseFreq)); }

	template <class Inputs>
	void seek(Inputs &&inputs, int inputSamples, double playbackRate) {
		gather(inputs, inputSamples, 0);
		check(smst_seek(h(), inPtrs.data(), inputSamples, playbackRate));
	}
	int seekLength() const { return smst_seek_length(h()); }
	template <class Inputs>
	void outputSeek(Inputs &&inputs, int inputLength) {
		gather(inputs, inputLength, 0);
		check(smst_output_seek(h(), inPtrs.data(), inputLength));
	}
	int outputSeekLength(Sample playbackRate) const { return smst_output_seek_length(h(), playbackRate); }

	// The reference's profiling hooks (:211-213, :329-331, :402-404, :420-422): START and END bracket the call as they do there.  The
	// per-step hooks: the spectral steps of a call run as kernels over the whole call, not interleaved with the output samples on the
	// host -- see process() for what STEP / ENDSTEP report.
	template <class Inputs, class Outputs>
	void process(Inputs &&inputs, int inputSamples, Outputs &&outputs, int outputSamples) {
#ifdef SIGNALSMITH_STRETCH_PROFILE_PROCESS_START
		SIGNALSMITH_STRETCH_PROFILE_PROCESS_START(inputSamples, outputSamples);
#endif
		gather(inputs, inputSamples, 0);
		prepareOut(outputSamples);
		// The reference announces every step of a block (STEP(step, steps)) and closes it (ENDSTEP) as the steps run, interleaved with the
		// output samples (:327-404).  Here a call's device work is ONE asynchronous submission, so the per-step attribution is SYNTHETIC:
		// after the call, and only if a block began in it (smst_blocks_started), every step of the newest block is announced and closed at
		// once with ONE count -- the reference's own for that block's flags and channel count (smst_block_steps) -- so that a harness
		// which sizes its per-step tables from `steps` (cmd/main-dev.cpp:44-52) sees consistent values.  The call's time lies between
		// START and the first STEP; a call that began several blocks still reports one (the newest); a call that began none reports none.
		check(smst_process(h(), inPtrs.data(), inputSamples, outPtrs.data(), outputSamples));
#if defined(SIGNALSMITH_STRETCH_PROFILE_PROCESS_STEP) && defined(SIGNALSMITH_STRETCH_PROFILE_PROCESS_ENDSTEP)
		if (smst_blocks_started(h()) > 0) {
			for (int step = 0, steps = smst_block_steps(h()); step < steps; ++step) {
				SIGNALSMITH_STRETCH_PROFILE_PROCESS_STEP(size_t(step), size_t(steps));
				SIGNALSMITH_STRETCH_PROFILE_PROCESS_ENDSTEP();
			}
		}
#endif
		scatter(outputs, outputSamples);
#ifdef SIGNALSMITH_STRETCH_PROFILE_PROCESS_END
		SIGNALSMITH_STRETCH_PROFILE_PROCESS_END();
#endif
	}
	template <class Outputs>
	void flush(Outputs &&outputs, int outputSamples, Sample playbackRate = 0) {
		prepareOut(outputSamples);
		check(smst_flush(h(), outPtrs.data(), outputSamples, playbackRate));
		scatter(outputs, outputSamples);
	}
	template <class Inputs, class Outputs>
	bool exact(Inputs &&inputs, int inputSamples, Outputs &&outputs, int outputSamples) {
		gather(inputs, inputSamples, 0);
		prepareOut(outputSamples);
		int rc = smst_exact(h(), inPtrs.data(), inputSamples, outPtrs.data(), outputSamples);
		if (rc != SMST_OK && rc != SMST_ERR_SHORT) check(rc);
		scatter(outputs, outputSamples);
		return rc == SMST_OK;
	}

private:
	mutable smst_stretch *handle = nullptr; // null only in a moved-from object, until its next use
	int channels = 0;
	long seed = 0;
	smst_stretch *h() const {
		if (!handle && smst_create(&handle, seed, defaultDevice()) != SMST_OK) throw std::runtime_error(smst_last_error());
		return handle;
	}
	std::vector<float> inPlanar, outPlanar;
	std::vector<const float *> inPtrs;
	std::vector<float *> outPtrs;

	static int defaultDevice() { return smst_default_device(); }
	static void check(int rc) {
		if (rc != SMST_OK) throw std::runtime_error(smst_last_error());
	}
	// the reference accepts anything indexable as buffer[channel][index] (README.md:46): copy to planar floats
	template <class Inputs>
	void gather(Inputs &&inputs, int n, int offset) {
		inPlanar.resize(size_t(channels)*size_t(n > 0 ? n : 1));
		inPtrs.resize(channels);
		for (int c = 0; c < channels; ++c) {
			auto &&channel = inputs[c];
			float *dst = inPlanar.data() + size_t(c)*size_t(n > 0 ? n : 1);
			for (int i = 0; i < n; ++i) dst[i] = channel[i + offset];
			inPtrs[c] = dst;
		}
	}
	void prepareOut(int n) {
		outPlanar.assign(size_t(channels)*size_t(n > 0 ? n : 1), 0.0f);
		outPtrs.resize(channels);
		for (int c = 0; c < channels; ++c) outPtrs[c] = outPlanar.data() + size_t(c)*size_t(n > 0 ? n : 1);
	}
	template <class Outputs>
	void scatter(Outputs &&outputs, int n) {
		for (int c = 0; c < channels; ++c) {
			auto &&channel = outputs[c];
			for (int i = 0; i < n; ++i) channel[i] = outPtrs[c][i];
		}
	}
};
template <typename Sample, class RandomEngine>
constexpr size_t SignalsmithStretch<Sample, RandomEngine>::version[3];

}} // namespace
#endif
