// TEST INFRASTRUCTURE ONLY (oracle) -- not on the product path; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this.
//
// Plain C++ (fp32) restatement of the reference's algorithm for the hot path
//   SignalsmithStretch<float>::process()   (/root/reference/signalsmith-stretch.h:210-423, :618-1036)
// written hop-by-hop in struct-of-arrays form: block schedule -> analysis -> energy/peaks/map/formants ->
// per-bin prediction coefficients -> bin recurrence -> synthesis -> overlap-add emission.  Every function cites
// the reference lines it restates.  The STFT layer (signalsmith-linear 0.2.6, absent from the reference tree) is
// the restatement in oracle/linear_shim/.
//
// PINNED: tests/test_oracle_golden.py checks this port (and oracle/_ref, the unmodified reference header) against
// golden vectors produced by the reference's own shipped WASM build, and test_port_matches_ref checks it
// sample-for-sample against oracle/_ref.  Coverage: process() from any state, seek(), reset(), short flush.
// Not restated: the std::default_random_engine path for stretch > 2x (:639-640, implementation-defined).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "signalsmith-linear/stft.h"

namespace {

using Complex = std::complex<float>;
using STFT = signalsmith::linear::DynamicSTFT<float, false, true>;

inline Complex mul(Complex a, Complex b) { // _impl::mul<false>, :17-26
	return Complex(a.real()*b.real() - a.imag()*b.imag(), a.real()*b.imag() + a.imag()*b.real());
}
inline Complex mulConj(Complex a, Complex b) { // _impl::mul<true>: a * conj(b)
	return Complex(b.real()*a.real() + b.imag()*a.imag(), b.real()*a.imag() - b.imag()*a.real());
}
inline float norm(Complex a) { return a.real()*a.real() + a.imag()*a.imag(); } // :27-31

struct Port {
	static constexpr float noiseFloor = 1e-15f;  // :508
	static constexpr float maxCleanStretch = 2;  // :509

	int channels = 0, bands = 0, block = 0, interval = 0;
	bool split = false;
	STFT stft;
	STFT::Input stashedInput;
	STFT::Output stashedOutput;

	// scheduler state (:494-505, :510-511, :527-529)
	size_t samplesSinceLast = SIZE_MAX;
	int prevInputOffset = -1;
	bool didSeek = false;
	float seekTimeFactor = 1;
	size_t silenceCounter = 0;
	bool silenceFirst = true;

	// parameters (:513-517, :971)
	float freqMultiplier = 1, freqTonalityLimit = 0.5f;
	bool formantCompensation = false;
	float formantMultiplier = 1, invFormantMultiplier = 1, formantBaseFreq = 0;
	float freqEstimateWeighted = 0, freqEstimateWeight = 0;

	// per-bin state, struct-of-arrays [channel][bin] (reference: Band{input, prevInput, output, inputEnergy} :538-542,
	// Prediction{energy, input} :592-595)
	std::vector<Complex> input, prevInput, output, predInput;
	std::vector<float> inputEnergy, predEnergy;
	std::vector<float> energy, smoothed, formantMetric, mapBin, mapGrad;
	std::vector<float> peakIn, peakOut;
	std::vector<float> tmp;

	void configure(int nChannels, int blockSamples, int intervalSamples, bool splitComputation) { // :71-94
		split = splitComputation;
		channels = nChannels;
		block = blockSamples;
		interval = intervalSamples;
		stft.configure(channels, channels, blockSamples, intervalSamples + 1);
		stft.setInterval(intervalSamples, stft.kaiser);
		stft.reset(0.1f);
		stashedInput = stft.input;
		stashedOutput = stft.output;
		bands = int(stft.bands());
		size_t n = size_t(bands)*channels;
		input.assign(n, 0); prevInput.assign(n, 0); output.assign(n, 0); predInput.assign(n, 0);
		inputEnergy.assign(n, 0); predEnergy.assign(n, 0);
		energy.assign(bands, 0); smoothed.assign(bands, 0); formantMetric.assign(bands + 2, 0);
		mapBin.assign(bands, 0); mapGrad.assign(bands, 1);
		samplesSinceLast = SIZE_MAX;
		tmp.resize(blockSamples + intervalSamples);
	}
	void reset() { // :49-60 (Prediction.energy is NOT cleared by the reference either)
		stft.reset(0.1f);
		stashedInput = stft.input;
		stashedOutput = stft.output;
		prevInputOffset = -1;
		std::fill(input.begin(), input.end(), Complex(0));
		std::fill(prevInput.begin(), prevInput.end(), Complex(0));
		std::fill(output.begin(), output.end(), Complex(0));
		silenceCounter = 0;
		didSeek = false;
		samplesSinceLast = SIZE_MAX;
		freqEstimateWeighted = freqEstimateWeight = 0;
	}
	int inputLatency() const { return int(stft.analysisLatency()); }                       // :42-44
	int outputLatency() const { return int(stft.synthesisLatency() + (split ? interval : 0)); } // :45-47

	void setTransposeFactor(float multiplier, float tonalityLimit) { // :107-115
		freqMultiplier = multiplier;
		freqTonalityLimit = (tonalityLimit > 0) ? tonalityLimit/std::sqrt(multiplier) : 1.0f;
	}
	void setFormantFactor(float multiplier, bool compensate) { // :124-128
		formantMultiplier = multiplier;
		invFormantMultiplier = 1/multiplier;
		formantCompensation = compensate;
	}

	float bandToFreq(float b) const { return stft.binToFreq(b); } // :531-536
	float freqToBand(float f) const { return stft.freqToBin(f); }
	float mapFreq(float freq) const { // :850-856
		if (freq > freqTonalityLimit) return freq + (freqMultiplier - 1)*freqTonalityLimit;
		return freq*freqMultiplier;
	}
	float invMapFormant(float freq) const { // :920-925
		if (freq*invFormantMultiplier > freqTonalityLimit) return freq + (1 - formantMultiplier)*freqTonalityLimit;
		return freq*invFormantMultiplier;
	}

	// fractional reads with zeros outside [0, bands)  (:547-580)
	template <typename T> T at(const std::vector<T> &a, int c, int i) const {
		if (i < 0 || i >= bands) return T(0);
		return a[size_t(c)*bands + i];
	}
	template <typename T> T lerp(const std::vector<T> &a, int c, int lo, float frac) const {
		T low = at(a, c, lo), high = at(a, c, lo + 1);
		return low + (high - low)*frac;
	}
	template <typename T> T lerp(const std::vector<T> &a, int c, float x) const {
		int lo = int(std::floor(x));
		return lerp(a, c, lo, x - lo);
	}

	void copyInput(const float *in, long stride, int toIndex, int &prevCopied) { // :215-229
		int length = std::min<int>(block + interval, toIndex - prevCopied);
		int offset = toIndex - length;
		for (int c = 0; c < channels; ++c) stft.writeInput(c, length, in + c*stride + offset);
		stft.moveInput(length);
		prevCopied = toIndex;
	}

	void seek(const float *in, long stride, int inputSamples, double playbackRate) { // :140-165
		int len = block + interval;
		int startIndex = std::max(0, inputSamples - len);
		int padStart = len + startIndex - inputSamples;
		float totalEnergy = 0;
		for (int c = 0; c < channels; ++c) {
			tmp.assign(len, 0.0f);
			for (int i = startIndex; i < inputSamples; ++i) {
				float s = in[c*stride + i];
				totalEnergy += s*s;
				tmp[i - startIndex + padStart] = s;
			}
			stft.writeInput(c, len, tmp.data());
		}
		stft.moveInput(len);
		if (totalEnergy >= noiseFloor) {
			silenceCounter = 0;
			silenceFirst = true;
		}
		didSeek = true;
		seekTimeFactor = (playbackRate*interval > 1) ? float(1/playbackRate) : float(interval);
	}

	// ---- one hop of spectral processing (:633-813), all steps at once ------------------------------------------
	void smoothEnergyAndPeaks() { // :818-848, :859-880
		const float smoothingBins = float(stft.fftSamples())/interval;
		const float slew = 1/(1 + smoothingBins*0.5f);
		std::fill(energy.begin(), energy.end(), 0.0f);
		for (int c = 0; c < channels; ++c) {
			for (int b = 0; b < bands; ++b) {
				float e = norm(input[size_t(c)*bands + b]);
				inputEnergy[size_t(c)*bands + b] = e;
				energy[b] += e;
			}
		}
		smoothed = energy;
		float e = 0;
		for (int rep = 0; rep < 2; ++rep) {
			for (int b = bands - 1; b >= 0; --b) { e += (smoothed[b] - e)*slew; smoothed[b] = e; }
			for (int b = 0; b < bands; ++b) { e += (smoothed[b] - e)*slew; smoothed[b] = e; }
		}
		peakIn.clear();
		peakOut.clear();
		int start = 0;
		while (start < bands) {
			if (energy[start] > smoothed[start]) {
				int end = start;
				float bandSum = 0, energySum = 0;
				while (end < bands && energy[end] > smoothed[end]) {
					bandSum += end*energy[end];
					energySum += energy[end];
					++end;
				}
				float avgBand = bandSum/energySum;
				peakIn.push_back(avgBand);
				peakOut.push_back(freqToBand(mapFreq(bandToFreq(avgBand))));
				start = end;
			}
			++start;
		}
	}
	void buildOutputMap() { // :882-917
		for (int b = 0; b < bands; ++b) { mapBin[b] = float(b); mapGrad[b] = 1; }
		if (peakIn.empty()) return;
		float bottomOffset = peakIn[0] - peakOut[0];
		for (int b = 0; b < std::min<int>(bands, int(std::ceil(peakOut[0]))); ++b) mapBin[b] = b + bottomOffset;
		for (size_t p = 1; p < peakIn.size(); ++p) {
			float prevI = peakIn[p - 1], prevO = peakOut[p - 1], nextI = peakIn[p], nextO = peakOut[p];
			float rangeScale = 1/(nextO - prevO);
			float outOffset = prevI - prevO;
			float outScale = nextI - nextO - prevI + prevO;
			float gradScale = outScale*rangeScale;
			int startBin = std::max<int>(0, int(std::ceil(prevO)));
			int endBin = std::min<int>(bands, int(std::ceil(nextO)));
			for (int b = startBin; b < endBin; ++b) {
				float r = (b - prevO)*rangeScale;
				float h = r*r*(3 - 2*r);
				mapBin[b] = b + outOffset + h*outScale;
				mapGrad[b] = 1 + 6*r*(1 - r)*gradScale;
			}
		}
		float topOffset = peakIn.back() - peakOut.back();
		for (int b = std::max<int>(0, int(peakOut.back())); b < bands; ++b) { mapBin[b] = b + topOffset; mapGrad[b] = 1; }
	}
	float estimateFrequency() { // :929-966
		int p0 = 0, p1 = 0, p2 = 0;
		for (int b = 1; b < bands - 1; ++b) {
			float e = formantMetric[b];
			if (e < formantMetric[b - 1] || e <= formantMetric[b + 1]) continue;
			if (e > formantMetric[p0]) {
				if (e > formantMetric[p1]) {
					if (e > formantMetric[p2]) { p0 = p1; p1 = p2; p2 = b; }
					else { p0 = p1; p1 = b; }
				} else {
					p0 = b;
				}
			}
		}
		int peakEstimate = p2;
		if (formantMetric[p1] > formantMetric[p2]*0.1f) {
			int diff = std::abs(peakEstimate - p1);
			if (diff > peakEstimate/8 && diff < peakEstimate*7/8) peakEstimate = peakEstimate%diff;
			if (formantMetric[p0] > formantMetric[p2]*0.01f) {
				int diff2 = std::abs(peakEstimate - p0);
				if (diff2 > peakEstimate/8 && diff2 < peakEstimate*7/8) peakEstimate = peakEstimate%diff2;
			}
		}
		float weight = formantMetric[p2];
		freqEstimateWeighted += (peakEstimate*weight - freqEstimateWeighted)*0.25f;
		freqEstimateWeight += (weight - freqEstimateWeight)*0.25f;
		return freqEstimateWeighted/(freqEstimateWeight + 1e-30f);
	}
	void applyFormants() { // :972-1036
		std::fill(formantMetric.begin(), formantMetric.end(), 0.0f);
		for (int c = 0; c < channels; ++c) for (int b = 0; b < bands; ++b) formantMetric[b] += inputEnergy[size_t(c)*bands + b];
		float freqEstimate = freqToBand(formantBaseFreq);
		if (formantBaseFreq <= 0) freqEstimate = estimateFrequency();
		float decay = 1 - 1/(freqEstimate*0.5f + 1);
		float e = 0;
		for (int rep = 0; rep < 2; ++rep) {
			for (int b = bands - 1; b >= 0; --b) { e = std::max(formantMetric[b], e*decay); formantMetric[b] = e; }
			for (int b = 0; b < bands; ++b) { e = std::max(formantMetric[b], e*decay); formantMetric[b] = e; }
		}
		decay = 1/decay;
		for (int rep = 0; rep < 2; ++rep) {
			for (int b = bands - 1; b >= 0; --b) { e = std::min(formantMetric[b], e*decay); formantMetric[b] = e; }
			for (int b = 0; b < bands; ++b) { e = std::min(formantMetric[b], e*decay); formantMetric[b] = e; }
		}
		for (int b = 0; b < bands; ++b) {
			float inputF = bandToFreq(float(b));
			float outputF = formantCompensation ? mapFreq(inputF) : inputF; // :1020
			outputF = invMapFormant(outputF);
			float band = freqToBand(outputF);
			float targetE = 0;
			if (!(band < 0)) {
				band = std::min<float>(band, float(bands));
				int fl = int(std::floor(band));
				float fr = band - fl;
				targetE = formantMetric[fl] + (formantMetric[fl + 1] - formantMetric[fl])*fr;
			}
			float ratio = targetE/(formantMetric[b] + 1e-30f);
			for (int c = 0; c < channels; ++c) inputEnergy[size_t(c)*bands + b] *= ratio;
		}
	}
	static Complex makeOutput(Complex phase, Complex in, float energyTarget) { // :596-603
		float phaseNorm = norm(phase);
		if (phaseNorm <= noiseFloor) {
			phase = in;
			phaseNorm = norm(in) + noiseFloor;
		}
		return phase*std::sqrt(energyTarget/phaseNorm);
	}
	void processSpectrum(bool newSpectrum, bool mapped, bool formants, float timeFactor) {
		const int L = int(std::round(float(stft.fftSamples())/interval)); // :636-637
		timeFactor = std::max<float>(timeFactor, 1/maxCleanStretch);      // :638
		if (newSpectrum) { // :642-660, the reference's own fp32 rotation recurrence
			for (int c = 0; c < channels; ++c) {
				Complex rot = std::polar(1.0f, bandToFreq(0)*interval*float(2*M_PI));
				float freqStep = bandToFreq(1) - bandToFreq(0);
				Complex rotStep = std::polar(1.0f, freqStep*interval*float(2*M_PI));
				for (int b = 0; b < bands; ++b) {
					size_t i = size_t(c)*bands + b;
					output[i] = mul(output[i], rot);
					prevInput[i] = mul(prevInput[i], rot);
					rot = mul(rot, rotStep);
				}
			}
		}
		if (mapped) { // :661-674
			smoothEnergyAndPeaks();
			buildOutputMap();
		} else { // :675-686
			for (size_t i = 0; i < input.size(); ++i) inputEnergy[i] = norm(input[i]);
			for (int b = 0; b < bands; ++b) { mapBin[b] = float(b); mapGrad[b] = 1; }
		}
		if (formants) applyFormants(); // :689-695
		for (int c = 0; c < channels; ++c) { // preliminary prediction, :697-719
			for (int b = 0; b < bands; ++b) {
				size_t i = size_t(c)*bands + b;
				int lo = int(std::floor(mapBin[b]));
				float frac = mapBin[b] - lo;
				float prevEnergy = predEnergy[i];
				predEnergy[i] = lerp(inputEnergy, c, lo, frac)*std::max<float>(0, mapGrad[b]);
				predInput[i] = lerp(input, c, lo, frac);
				Complex prevIn = lerp(prevInput, c, lo, frac);
				Complex twist = mulConj(predInput[i], prevIn);
				output[i] = mul(output[i], twist)/(std::max(prevEnergy, predEnergy[i]) + noiseFloor);
			}
		}
		for (int b = 0; b < bands; ++b) { // main prediction + channel locking, :727-801
			int mc = 0;
			float maxEnergy = predEnergy[b];
			for (int c = 1; c < channels; ++c) {
				float e = predEnergy[size_t(c)*bands + b];
				if (e > maxEnergy) { mc = c; maxEnergy = e; }
			}
			size_t row = size_t(mc)*bands;
			Complex phase = 0;
			if (b > 0) {
				Complex down = lerp(input, mc, mapBin[b] - timeFactor);
				phase += mul(output[row + b - 1], mulConj(predInput[row + b], down));
				if (b >= L) {
					Complex longDown = lerp(input, mc, mapBin[b] - L*timeFactor);
					phase += mul(output[row + b - L], mulConj(predInput[row + b], longDown));
				}
			}
			if (b < bands - 1) {
				Complex down = lerp(input, mc, mapBin[b + 1] - timeFactor);
				phase += mulConj(output[row + b + 1], mulConj(predInput[row + b + 1], down));
				if (b < bands - L) {
					Complex longDown = lerp(input, mc, mapBin[b + L] - L*timeFactor);
					phase += mulConj(output[row + b + L], mulConj(predInput[row + b + L], longDown));
				}
			}
			output[row + b] = makeOutput(phase, predInput[row + b], predEnergy[row + b]);
			for (int c = 0; c < channels; ++c) {
				if (c == mc) continue;
				size_t i = size_t(c)*bands + b;
				Complex channelTwist = mulConj(predInput[i], predInput[row + b]);
				output[i] = makeOutput(mul(output[row + b], channelTwist), predInput[i], predEnergy[i]);
			}
		}
		if (newSpectrum) prevInput = input; // :806-811
	}

	// ---- process(), hop-structured (:210-423) -------------------------------------------------------------------
	void process(const float *in, long inStride, int inputSamples, float *out, long outStride, int outputSamples) {
		int prevCopied = 0;
		float totalEnergy = 0; // :231-238
		for (int c = 0; c < channels; ++c) for (int i = 0; i < inputSamples; ++i) totalEnergy += in[c*inStride + i]*in[c*inStride + i];
		if (totalEnergy < noiseFloor) { // :240-278
			if (silenceCounter >= size_t(2*block)) {
				if (silenceFirst) {
					silenceFirst = false;
					samplesSinceLast = SIZE_MAX;
					std::fill(input.begin(), input.end(), Complex(0));
					std::fill(prevInput.begin(), prevInput.end(), Complex(0));
					std::fill(output.begin(), output.end(), Complex(0));
					std::fill(inputEnergy.begin(), inputEnergy.end(), 0.0f);
				}
				for (int c = 0; c < channels; ++c)
					for (int i = 0; i < outputSamples; ++i) out[c*outStride + i] = inputSamples > 0 ? in[c*inStride + i%inputSamples] : 0.0f;
				copyInput(in, inStride, inputSamples, prevCopied);
				return;
			}
			silenceCounter += inputSamples;
		} else {
			silenceCounter = 0;
			silenceFirst = true;
		}
		for (int outputIndex = 0; outputIndex < outputSamples; ++outputIndex) {
			if (samplesSinceLast >= size_t(interval)) { // new block, :281-319 (all steps executed at once)
				samplesSinceLast = 0;
				int inputOffset = int(std::round(outputIndex*float(inputSamples)/outputSamples)); // :288
				int inputInterval = inputOffset - prevInputOffset;
				prevInputOffset = inputOffset;
				copyInput(in, inStride, inputOffset, prevCopied);
				if (split) { // :294-297
					stashedOutput = stft.output;
					stft.moveOutput(interval);
				}
				bool newSpectrum = didSeek || inputInterval > 0;                         // :299
				bool mapped = freqMultiplier != 1;                                        // :300
				bool reanalysePrev = newSpectrum && (didSeek || std::abs(inputInterval - interval) > 1); // :303
				bool formants = formantMultiplier != 1 || (formantCompensation && mapped); // :310
				float timeFactor = didSeek ? seekTimeFactor : interval/std::max<float>(1, float(inputInterval)); // :312
				didSeek = false;
				if (newSpectrum) {
					if (reanalysePrev) { // :333-354
						for (int c = 0; c < channels; ++c) {
							stft.analyseStep(c, interval);
							std::copy(stft.spectrum(c), stft.spectrum(c) + bands, prevInput.begin() + size_t(c)*bands);
						}
					}
					for (int c = 0; c < channels; ++c) { // :357-376
						stft.analyseStep(c);
						std::copy(stft.spectrum(c), stft.spectrum(c) + bands, input.begin() + size_t(c)*bands);
					}
				}
				processSpectrum(newSpectrum, mapped, formants, timeFactor);
				for (int c = 0; c < channels; ++c) { // :384-399
					std::copy(output.begin() + size_t(c)*bands, output.begin() + size_t(c + 1)*bands, stft.spectrum(c));
					stft.synthesiseStep(c);
				}
			}
			++samplesSinceLast; // :406-415
			if (split) stashedOutput.swap(stft.output);
			for (int c = 0; c < channels; ++c) {
				float v = 0;
				stft.readOutput(c, 1, &v);
				out[c*outStride + outputIndex] = v;
			}
			stft.moveOutput(1);
			if (split) stashedOutput.swap(stft.output);
		}
		copyInput(in, inStride, inputSamples, prevCopied); // :418-419
		prevInputOffset -= inputSamples;
	}

	void flushShort(float *out, long outStride, int outputSamples) { // :442-463 for outputSamples <= interval
		int tail = std::min(outputSamples, interval);
		tmp.resize(tail);
		stft.finishOutput(1);
		for (int c = 0; c < channels; ++c) {
			stft.readOutput(c, tail, tmp.data());
			for (int i = 0; i < tail; ++i) out[c*outStride + i] = tmp[i];
			stft.readOutput(c, tail, tail, tmp.data());
			for (int i = 0; i < tail; ++i) out[c*outStride + tail - 1 - i] -= tmp[i];
		}
		stft.reset(0.1f);
		std::fill(prevInput.begin(), prevInput.end(), Complex(0));
		std::fill(output.begin(), output.end(), Complex(0));
	}
};

} // namespace

extern "C" {
void *smst_port_create() { return new Port(); }
void smst_port_destroy(void *h) { delete static_cast<Port *>(h); }
void smst_port_configure(void *h, int channels, int block, int interval, int split) { static_cast<Port *>(h)->configure(channels, block, interval, split != 0); }
void smst_port_preset_default(void *h, int channels, float sr, int split) { static_cast<Port *>(h)->configure(channels, int(sr*0.12), int(sr*0.03), split != 0); } // :63-65
void smst_port_preset_cheaper(void *h, int channels, float sr, int split) { static_cast<Port *>(h)->configure(channels, int(sr*0.1), int(sr*0.04), split != 0); }  // :66-68
void smst_port_reset(void *h) { static_cast<Port *>(h)->reset(); }
int smst_port_block_samples(void *h) { return static_cast<Port *>(h)->block; }
int smst_port_interval_samples(void *h) { return static_cast<Port *>(h)->interval; }
int smst_port_input_latency(void *h) { return static_cast<Port *>(h)->inputLatency(); }
int smst_port_output_latency(void *h) { return static_cast<Port *>(h)->outputLatency(); }
void smst_port_set_transpose_factor(void *h, float m, float t) { static_cast<Port *>(h)->setTransposeFactor(m, t); }
void smst_port_set_transpose_semitones(void *h, float s, float t) { static_cast<Port *>(h)->setTransposeFactor(float(std::pow(2, s/12)), t); } // :116-118
void smst_port_set_formant_factor(void *h, float m, int comp) { static_cast<Port *>(h)->setFormantFactor(m, comp != 0); }
void smst_port_set_formant_semitones(void *h, float s, int comp) { static_cast<Port *>(h)->setFormantFactor(float(std::pow(2, s/12)), comp != 0); }
void smst_port_set_formant_base(void *h, float f) { static_cast<Port *>(h)->formantBaseFreq = f; }
void smst_port_seek(void *h, const float *in, long stride, int n, double rate) { static_cast<Port *>(h)->seek(in, stride, n, rate); }
void smst_port_process(void *h, const float *in, long inStride, int nIn, float *out, long outStride, int nOut) {
	static_cast<Port *>(h)->process(in, inStride, nIn, out, outStride, nOut);
}
void smst_port_flush_short(void *h, float *out, long outStride, int nOut) { static_cast<Port *>(h)->flushShort(out, outStride, nOut); }
}
