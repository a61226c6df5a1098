// TEST INFRASTRUCTURE ONLY (oracle) -- not on the product path.
//
// Flat C ABI around the UNMODIFIED reference header (/root/reference/signalsmith-stretch.h), compiled
// where it lies against the L1 restatement in oracle/linear_shim/.  Built by oracle/Makefile into
// oracle/_ref/libsmst_ref.so.  Used by tests/ (the checker), __graft_entry__.smoke() and bench.py's
// cpu_baseline leg only.
//
// `private` is opened up so that tests can inspect / inject the per-bin state (teacher-forced parity,
// SURVEY.md App. D.2); the standard headers are included first so the macro only touches the reference.
#include <algorithm>
#include <array>
#include <cmath>
#include <complex>
#include <cstring>
#include <functional>
#include <limits>
#include <random>
#include <type_traits>
#include <vector>
#include "signalsmith-linear/stft.h"

#define private public
#include "signalsmith-stretch.h"
#undef private

using Stretch = signalsmith::stretch::SignalsmithStretch<float>;

namespace {
struct Planar {
	const float *base;
	long stride;
	const float *operator[](int c) const { return base + c*stride; }
};
struct PlanarOut {
	float *base;
	long stride;
	float *operator[](int c) { return base + c*stride; }
};
struct RefHandle {
	Stretch stretch;
	std::vector<float> mapTable;
	RefHandle(long seed) : stretch(seed) {}
};
}

extern "C" {

void *smst_ref_create(long seed) { return new RefHandle(seed); }
void smst_ref_destroy(void *h) { delete static_cast<RefHandle *>(h); }
#define S (static_cast<RefHandle *>(h)->stretch)

void smst_ref_preset_default(void *h, int channels, float sampleRate, int split) { S.presetDefault(channels, sampleRate, split != 0); }
void smst_ref_preset_cheaper(void *h, int channels, float sampleRate, int split) { S.presetCheaper(channels, sampleRate, split != 0); }
void smst_ref_configure(void *h, int channels, int block, int interval, int split) { S.configure(channels, block, interval, split != 0); }
void smst_ref_reset(void *h) { S.reset(); }
int smst_ref_block_samples(void *h) { return S.blockSamples(); }
int smst_ref_interval_samples(void *h) { return S.intervalSamples(); }
int smst_ref_input_latency(void *h) { return S.inputLatency(); }
int smst_ref_output_latency(void *h) { return S.outputLatency(); }
int smst_ref_fft_samples(void *h) { return int(S.stft.fftSamples()); }
int smst_ref_bands(void *h) { return S.bands; }
int smst_ref_seek_length(void *h) { return S.seekLength(); }
int smst_ref_output_seek_length(void *h, float rate) { return S.outputSeekLength(rate); }
void smst_ref_set_transpose_factor(void *h, float mult, float tonality) { S.setTransposeFactor(mult, tonality); }
void smst_ref_set_transpose_semitones(void *h, float semis, float tonality) { S.setTransposeSemitones(semis, tonality); }
void smst_ref_set_formant_factor(void *h, float mult, int comp) { S.setFormantFactor(mult, comp != 0); }
void smst_ref_set_formant_semitones(void *h, float semis, int comp) { S.setFormantSemitones(semis, comp != 0); }
void smst_ref_set_formant_base(void *h, float f) { S.setFormantBase(f); }
// frequency map as a table sampled at (i + 0.5)/(2 n) i.e. bin centres of an n-bin grid, lerp in between
void smst_ref_set_freq_map_table(void *h, const float *table, int n) {
	auto *handle = static_cast<RefHandle *>(h);
	if (!table || n <= 0) {
		handle->stretch.setFreqMap(nullptr);
		return;
	}
	handle->mapTable.assign(table, table + n);
	auto *t = &handle->mapTable;
	handle->stretch.setFreqMap([t](float f) {
		float pos = f*2*float(t->size()) - 0.5f;
		int n = int(t->size());
		if (pos <= 0) return (*t)[0] + ((*t)[1] - (*t)[0])*pos;
		if (pos >= n - 1) return (*t)[n - 1] + ((*t)[n - 1] - (*t)[n - 2])*(pos - (n - 1));
		int lo = int(std::floor(pos));
		float frac = pos - lo;
		return (*t)[lo] + ((*t)[lo + 1] - (*t)[lo])*frac;
	});
}
void smst_ref_seek(void *h, const float *in, long stride, int inputSamples, double rate) {
	Planar p{in, stride};
	S.seek(p, inputSamples, rate);
}
void smst_ref_output_seek(void *h, const float *in, long stride, int inputLength) {
	Planar p{in, stride};
	S.outputSeek(p, inputLength);
}
void smst_ref_process(void *h, const float *in, long inStride, int inputSamples, float *out, long outStride, int outputSamples) {
	Planar p{in, inStride};
	PlanarOut o{out, outStride};
	S.process(p, inputSamples, o, outputSamples);
}
void smst_ref_flush(void *h, float *out, long outStride, int outputSamples, float playbackRate) {
	PlanarOut o{out, outStride};
	S.flush(o, outputSamples, playbackRate);
}
int smst_ref_exact(void *h, const float *in, long inStride, int inputSamples, float *out, long outStride, int outputSamples) {
	Planar p{in, inStride};
	PlanarOut o{out, outStride};
	return S.exact(p, inputSamples, o, outputSamples) ? 1 : 0;
}

// ---- state inspection (tests only). which: 0=input 1=prevInput 2=output (complex, 2 floats/bin), 3=inputEnergy, 4=prediction energy
void smst_ref_get_bands(void *h, int which, float *dst) {
	int n = S.channels*S.bands;
	for (int i = 0; i < n; ++i) {
		const auto &b = S.channelBands[i];
		switch (which) {
		case 0: dst[2*i] = b.input.real(); dst[2*i + 1] = b.input.imag(); break;
		case 1: dst[2*i] = b.prevInput.real(); dst[2*i + 1] = b.prevInput.imag(); break;
		case 2: dst[2*i] = b.output.real(); dst[2*i + 1] = b.output.imag(); break;
		case 3: dst[i] = b.inputEnergy; break;
		case 4: dst[i] = S.channelPredictions[i].energy; break;
		}
	}
}
// teacher forcing between two checker instances (one-hop conditioning measurements): overwrite the per-bin state
void smst_ref_set_bands(void *h, int which, const float *src) {
	int n = S.channels*S.bands;
	for (int i = 0; i < n; ++i) {
		auto &b = S.channelBands[i];
		switch (which) {
		case 0: b.input = {src[2*i], src[2*i + 1]}; break;
		case 1: b.prevInput = {src[2*i], src[2*i + 1]}; break;
		case 2: b.output = {src[2*i], src[2*i + 1]}; break;
		case 3: b.inputEnergy = src[i]; break;
		case 4: S.channelPredictions[i].energy = src[i]; break;
		}
	}
}
void smst_ref_set_output_ring(void *h, const float *sums, const float *products) {
	int B = S.blockSamples();
	for (int c = 0; c < S.channels; ++c) {
		for (int i = 0; i < B; ++i) {
			size_t i2 = (S.stft.output.pos + i)%B;
			S.stft.output.buffer[i2 + size_t(c)*B] = sums[c*B + i];
			if (c == 0) S.stft.output.windowProducts[i2] = products[i];
		}
	}
}
// formant envelope (bands + 2 entries) and the pitch estimate it was built with (signalsmith-stretch.h:968-1006)
float smst_ref_get_formant_metric(void *h, float *dst) {
	std::copy(S.formantMetric.begin(), S.formantMetric.end(), dst);
	return S.freqEstimate;
}
// channel-summed energy and its smoothed version as findPeaks saw them (signalsmith-stretch.h:818-848)
void smst_ref_get_energy(void *h, float *energy, float *smoothed) {
	std::copy(S.energy.begin(), S.energy.end(), energy);
	std::copy(S.smoothedEnergy.begin(), S.smoothedEnergy.end(), smoothed);
}
void smst_ref_get_output_map(void *h, float *dst) {
	for (int b = 0; b < S.bands; ++b) {
		dst[2*b] = S.outputMap[b].inputBin;
		dst[2*b + 1] = S.outputMap[b].freqGrad;
	}
}
int smst_ref_get_peaks(void *h, float *dst, int maxPeaks) {
	int n = int(S.peaks.size());
	for (int i = 0; i < n && i < maxPeaks; ++i) {
		dst[2*i] = S.peaks[i].input;
		dst[2*i + 1] = S.peaks[i].output;
	}
	return n;
}
void smst_ref_get_window(void *h, float *dst) {
	const auto &w = S.stft.analysisWindow();
	std::copy(w.begin(), w.end(), dst);
}
// output ring as the reader sees it: accumulators and window products starting at the read position
void smst_ref_get_output_ring(void *h, float *sums, float *products) {
	int B = S.blockSamples();
	for (int c = 0; c < S.channels; ++c) {
		for (int i = 0; i < B; ++i) {
			size_t i2 = (S.stft.output.pos + i)%B;
			sums[c*B + i] = S.stft.output.buffer[i2 + size_t(c)*B];
			if (c == 0) products[i] = S.stft.output.windowProducts[i2];
		}
	}
}
// spectrum of one windowed block through the L1 restatement (for checking the GPU FFT kernels in isolation)
void smst_ref_analyse_block(void *h, const float *block, float *spectrumOut) {
	auto &stft = S.stft;
	auto saved = stft.input;
	size_t B = stft.blockSamples();
	stft.writeInput(0, B, block);
	stft.moveInput(B);
	stft.analyseStep(0);
	for (size_t b = 0; b < stft.bands(); ++b) {
		spectrumOut[2*b] = stft.spectrum(0)[b].real();
		spectrumOut[2*b + 1] = stft.spectrum(0)[b].imag();
	}
	stft.input = saved;
}
#undef S
}
