// TEST INFRASTRUCTURE ONLY (oracle) -- not on the product path.
//
// CPU restatement of the un-vendored dependency `Signalsmith-Audio/linear` tag 0.2.6
// (`signalsmith-linear/stft.h`, fetched by the reference's CMakeLists.txt:6-13 and absent from
// /root/reference).  It provides exactly the members of
// `signalsmith::linear::DynamicSTFT<Sample,false,true>` that the reference header uses
// (signalsmith-stretch.h:43,46,50-52,74-80,97,100,142,156,158,164,167,202,217,225,227,241,281,
// 293-296,303-307,312,318,335-338,346,357-360,368,388,397-398,407,411,414-415,439,444-456,519-522,
// 532,535,636) so that the UNMODIFIED reference header compiles into oracle/_ref/.
//
// The arithmetic follows the published behaviour of that library as established in SURVEY.md
// Appendix A (probe-verified against the reference's shipped WASM build): half-bin-shifted
// ("modified") real spectrum with the time origin at the window centre, unnormalised forward
// transform, inverse gain = fftSamples, Kaiser window with heuristic bandwidth forced to perfect
// reconstruction, window-product normalisation seeded by reset(weight).  This file is pinned by
// tests/test_oracle_golden.py (oracle/_ref vs. the WASM golden vectors in tests/golden/).
#ifndef SMST_ORACLE_LINEAR_STFT_SHIM_H
#define SMST_ORACLE_LINEAR_STFT_SHIM_H

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstddef>
#include <vector>

namespace signalsmith { namespace linear {

namespace shim_detail {

// Mixed-radix (2,3,4,5) out-of-place Stockham FFT, decimation in frequency, natural order in/out.
template <typename Sample>
struct ComplexFFT {
	using Complex = std::complex<Sample>;
	size_t size = 0;
	std::vector<int> radices;
	std::vector<std::vector<Complex>> stageTwiddles; // per stage: [p*(r-1) + (q-1)] = exp(-2 pi i p q / nCur)
	std::vector<Complex> scratch;

	static size_t fastSizeAbove(size_t size) {
		size_t power2 = 1;
		while (power2*8 < size) power2 *= 2;
		size_t multiple = (size + power2 - 1)/power2;
		if (multiple == 7) ++multiple;
		return multiple*power2;
	}

	void resize(size_t n) {
		size = n;
		radices.clear();
		stageTwiddles.clear();
		scratch.assign(n, Complex(0));
		size_t rem = n;
		// odd factors first (they see the longest contiguous inner loops last), then radix-4, then a 2
		std::vector<int> r;
		while (rem%5 == 0) { r.push_back(5); rem /= 5; }
		while (rem%3 == 0) { r.push_back(3); rem /= 3; }
		while (rem%4 == 0) { r.push_back(4); rem /= 4; }
		while (rem%2 == 0) { r.push_back(2); rem /= 2; }
		if (rem != 1) { // unsupported prime factor: fall back to a (slow) generic radix
			r.push_back(int(rem));
		}
		radices = r;
		size_t nCur = n;
		for (int radix : radices) {
			size_t m = nCur/radix;
			std::vector<Complex> tw(m*(radix - 1));
			for (size_t p = 0; p < m; ++p) {
				for (int q = 1; q < radix; ++q) {
					double ang = -2*M_PI*double(p*q)/double(nCur);
					tw[p*(radix - 1) + (q - 1)] = Complex(Sample(std::cos(ang)), Sample(std::sin(ang)));
				}
			}
			stageTwiddles.push_back(std::move(tw));
			nCur = m;
		}
	}

	static Complex mulI(Complex v) { return Complex(-v.imag(), v.real()); }   // v * i
	static Complex mulNegI(Complex v) { return Complex(v.imag(), -v.real()); } // v * -i
	static Complex cmul(Complex a, Complex b) {
		return Complex(a.real()*b.real() - a.imag()*b.imag(), a.real()*b.imag() + a.imag()*b.real());
	}

	template <bool inverse>
	void pass(int radix, size_t nCur, size_t s, const Complex *tw, const Complex *x, Complex *y) const {
		size_t m = nCur/radix;
		for (size_t p = 0; p < m; ++p) {
			const Complex *w = tw + p*(radix - 1);
			Complex wq[8];
			for (int q = 1; q < radix && q < 8; ++q) wq[q] = inverse ? std::conj(w[q - 1]) : w[q - 1];
			for (size_t q0 = 0; q0 < s; ++q0) {
				const Complex *in = x + q0 + s*p;
				Complex *out = y + q0 + s*radix*p;
				size_t inStride = s*m;
				if (radix == 4) {
					Complex a = in[0], b = in[inStride], c = in[2*inStride], d = in[3*inStride];
					Complex apc = a + c, amc = a - c, bpd = b + d, bmd = b - d;
					Complex jbmd = inverse ? mulI(bmd) : mulNegI(bmd);
					out[0] = apc + bpd;
					out[s] = cmul(amc + jbmd, wq[1]);
					out[2*s] = cmul(apc - bpd, wq[2]);
					out[3*s] = cmul(amc - jbmd, wq[3]);
				} else if (radix == 2) {
					Complex a = in[0], b = in[inStride];
					out[0] = a + b;
					out[s] = cmul(a - b, wq[1]);
				} else if (radix == 3) {
					const Sample c3 = Sample(-0.5), s3 = Sample(0.86602540378443864676);
					Complex a = in[0], b = in[inStride], c = in[2*inStride];
					Complex bpc = b + c, bmc = b - c;
					Complex t = a + bpc*c3;
					Complex u = (inverse ? mulI(bmc) : mulNegI(bmc))*s3;
					out[0] = a + bpc;
					out[s] = cmul(t + u, wq[1]);
					out[2*s] = cmul(t - u, wq[2]);
				} else if (radix == 5) {
					const Sample c1 = Sample(0.30901699437494742410), c2 = Sample(-0.80901699437494742410);
					const Sample s1 = Sample(0.95105651629515357212), s2 = Sample(0.58778525229247312917);
					Complex a = in[0], b = in[inStride], c = in[2*inStride], d = in[3*inStride], e = in[4*inStride];
					Complex bpe = b + e, bme = b - e, cpd = c + d, cmd = c - d;
					Complex t1 = a + bpe*c1 + cpd*c2;
					Complex t2 = a + bpe*c2 + cpd*c1;
					Complex u1 = bme*s1 + cmd*s2;
					Complex u2 = bme*s2 - cmd*s1;
					Complex ju1 = inverse ? mulI(u1) : mulNegI(u1);
					Complex ju2 = inverse ? mulI(u2) : mulNegI(u2);
					out[0] = a + bpe + cpd;
					out[s] = cmul(t1 + ju1, wq[1]);
					out[2*s] = cmul(t2 + ju2, wq[2]);
					out[3*s] = cmul(t2 - ju2, wq[3]);
					out[4*s] = cmul(t1 - ju1, wq[4]);
				} else { // generic O(r^2) butterfly
					for (int j = 0; j < radix; ++j) {
						Complex sum = 0;
						for (int k = 0; k < radix; ++k) {
							double ang = (inverse ? 2 : -2)*M_PI*double((j*k)%radix)/radix;
							sum += cmul(in[k*inStride], Complex(Sample(std::cos(ang)), Sample(std::sin(ang))));
						}
						if (j > 0) {
							Complex wj = w[j - 1];
							sum = cmul(sum, inverse ? std::conj(wj) : wj);
						}
						out[j*s] = sum;
					}
				}
			}
		}
	}

	// in and out must not alias
	template <bool inverse>
	void run(const Complex *in, Complex *out) {
		size_t stages = radices.size();
		if (stages == 0) {
			if (size) out[0] = in[0];
			return;
		}
		if (scratchB.size() != size) scratchB.assign(size, Complex(0));
		const Complex *src = in;
		size_t nCur = size, s = 1;
		for (size_t st = 0; st < stages; ++st) {
			Complex *dst = (st + 1 == stages) ? out : (st%2 == 0 ? scratch.data() : scratchB.data());
			pass<inverse>(radices[st], nCur, s, stageTwiddles[st].data(), src, dst);
			nCur /= radices[st];
			s *= radices[st];
			src = dst;
		}
	}
	std::vector<Complex> scratchB;
};

} // namespace shim_detail

template <typename Sample, bool splitComputation = false, bool modified = false>
struct DynamicSTFT {
	static_assert(!splitComputation, "oracle shim: only the non-split L1 mode (the one the reference header uses) is restated");
	static_assert(modified, "oracle shim: only the modified (half-bin-shifted) spectrum (the one the reference header uses) is restated");
	using Complex = std::complex<Sample>;

	enum WindowShape { ignore, acg, kaiser };
	static constexpr Sample almostZero = Sample(1e-30);

	struct Input {
		size_t pos = 0;
		std::vector<Sample> buffer;
		void swap(Input &other) {
			std::swap(pos, other.pos);
			std::swap(buffer, other.buffer);
		}
	};
	struct Output {
		size_t pos = 0;
		std::vector<Sample> buffer;
		std::vector<Sample> windowProducts;
		void swap(Output &other) {
			std::swap(pos, other.pos);
			std::swap(buffer, other.buffer);
			std::swap(windowProducts, other.windowProducts);
		}
	};
	Input input;
	Output output;

	void configure(size_t inChannels, size_t outChannels, size_t blockSamples, size_t extraInputHistory = 0, size_t intervalSamples = 0) {
		_inChannels = inChannels;
		_outChannels = outChannels;
		_blockSamples = blockSamples;
		_fftSamples = 2*shim_detail::ComplexFFT<Sample>::fastSizeAbove((blockSamples + 1)/2);
		_bands = _fftSamples/2;
		_inputLength = blockSamples + extraInputHistory;
		fft.resize(_bands);
		halfTwiddle.resize(_bands);
		for (size_t m = 0; m < _bands; ++m) {
			double ang = -M_PI*double(m)/double(_fftSamples);
			halfTwiddle[m] = Complex(Sample(std::cos(ang)), Sample(std::sin(ang)));
		}
		input.buffer.assign(_inputLength*_inChannels, 0);
		output.buffer.assign(_blockSamples*_outChannels, 0);
		output.windowProducts.assign(_blockSamples, 0);
		spectrumBuffer.assign(_bands*std::max(_inChannels, _outChannels), Complex(0));
		timeBuffer.assign(_fftSamples, 0);
		packed.assign(_bands, Complex(0));
		packedOut.assign(_bands, Complex(0));
		_analysisWindow.assign(_blockSamples, 0);
		_synthesisWindow.assign(_blockSamples, 0);
		setInterval(intervalSamples ? intervalSamples : blockSamples/4, kaiser);
		reset();
	}

	size_t blockSamples() const { return _blockSamples; }
	size_t fftSamples() const { return _fftSamples; }
	size_t defaultInterval() const { return _defaultInterval; }
	size_t bands() const { return _bands; }
	size_t analysisLatency() const { return _blockSamples - _analysisOffset; }
	size_t synthesisLatency() const { return _synthesisOffset; }

	Sample binToFreq(Sample b) const { return (b + Sample(0.5))/Sample(_fftSamples); }
	Sample freqToBin(Sample f) const { return f*Sample(_fftSamples) - Sample(0.5); }

	// Kaiser window, heuristic-optimal bandwidth, forced to perfect reconstruction over the interval
	// (probe-verified to 1 ulp).
	void setInterval(size_t defaultInterval, WindowShape shape = ignore) {
		_defaultInterval = defaultInterval;
		if (shape == ignore) return;
		double bandwidth = double(_blockSamples)/double(defaultInterval);
		if (shape == kaiser || shape == acg) {
			double bw = bandwidth + 8/((bandwidth + 3)*(bandwidth + 3)) + 0.25*std::max(3 - bandwidth, 0.0);
			bw = std::max(bw, 2.0);
			double beta = M_PI*std::sqrt(bw*bw*0.25 - 1);
			double invB0 = 1/bessel0(beta);
			for (size_t i = 0; i < _blockSamples; ++i) {
				double r = (2*double(i) + 1)/double(_blockSamples) - 1;
				double arg = std::sqrt(std::max(0.0, 1 - r*r));
				_synthesisWindow[i] = Sample(bessel0(beta*arg)*invB0);
			}
		}
		for (size_t j = 0; j < defaultInterval && j < _blockSamples; ++j) {
			Sample sum2 = 0;
			for (size_t i = j; i < _blockSamples; i += defaultInterval) sum2 += _synthesisWindow[i]*_synthesisWindow[i];
			Sample gain = 1/std::sqrt(sum2);
			for (size_t i = j; i < _blockSamples; i += defaultInterval) _synthesisWindow[i] *= gain;
		}
		_analysisWindow = _synthesisWindow;
		_analysisOffset = _synthesisOffset = _blockSamples/2;
	}

	// window products as if previous blocks had been added at `productWeight`
	void reset(Sample productWeight = 1) {
		input.pos = _blockSamples;
		output.pos = 0;
		std::fill(input.buffer.begin(), input.buffer.end(), Sample(0));
		std::fill(output.buffer.begin(), output.buffer.end(), Sample(0));
		std::fill(spectrumBuffer.begin(), spectrumBuffer.end(), Complex(0));
		std::fill(output.windowProducts.begin(), output.windowProducts.end(), Sample(0));
		addWindowProduct();
		for (int i = int(_blockSamples) - int(_defaultInterval) - 1; i >= 0; --i) {
			output.windowProducts[i] += output.windowProducts[i + _defaultInterval];
		}
		for (auto &v : output.windowProducts) v = v*productWeight + almostZero;
		moveOutput(_defaultInterval);
	}

	void writeInput(size_t channel, size_t offset, size_t length, const Sample *data) {
		Sample *buffer = input.buffer.data() + channel*_inputLength;
		size_t start = (input.pos + offset)%_inputLength;
		for (size_t i = 0; i < length; ++i) {
			size_t i2 = start + i;
			if (i2 >= _inputLength) i2 -= _inputLength;
			buffer[i2] = data[i];
		}
	}
	void writeInput(size_t channel, size_t length, const Sample *data) {
		writeInput(channel, 0, length, data);
	}
	void moveInput(size_t samples) {
		input.pos = (input.pos + samples)%_inputLength;
	}

	size_t analyseSteps() const { return _inChannels; }
	void analyse(size_t samplesInPast = 0) {
		for (size_t s = 0; s < analyseSteps(); ++s) analyseStep(s, samplesInPast);
	}
	// Window the block ending `samplesInPast` before the input head; forward modified real FFT.
	void analyseStep(size_t step, size_t samplesInPast = 0) {
		size_t channel = step;
		const Sample *buffer = input.buffer.data() + channel*_inputLength;
		size_t N = _fftSamples;
		std::fill(timeBuffer.begin(), timeBuffer.end(), Sample(0));
		// block sample i sits at time (i - analysisOffset) relative to the window centre
		size_t start = (input.pos + 2*_inputLength - (_blockSamples + samplesInPast)%_inputLength)%_inputLength;
		for (size_t i = 0; i < _blockSamples; ++i) {
			size_t i2 = start + i;
			if (i2 >= _inputLength) i2 -= _inputLength;
			Sample v = buffer[i2]*_analysisWindow[i];
			if (i < _analysisOffset) {
				timeBuffer[N + i - _analysisOffset] = -v; // negative time: basis has period 2N, antiperiodic in N
			} else {
				timeBuffer[i - _analysisOffset] = v;
			}
		}
		forwardModifiedReal(timeBuffer.data(), spectrum(channel));
	}

	Complex *spectrum(size_t channel) { return spectrumBuffer.data() + channel*_bands; }
	const Complex *spectrum(size_t channel) const { return spectrumBuffer.data() + channel*_bands; }

	size_t synthesiseSteps() const { return _outChannels; }
	void synthesise() {
		for (size_t s = 0; s < synthesiseSteps(); ++s) synthesiseStep(s);
	}
	void synthesiseStep(size_t step) {
		if (step == 0) addWindowProduct();
		size_t channel = step;
		inverseModifiedReal(spectrum(channel), timeBuffer.data());
		size_t N = _fftSamples;
		Sample *buffer = output.buffer.data() + channel*_blockSamples;
		for (size_t i = 0; i < _blockSamples; ++i) {
			Sample v;
			if (i < _synthesisOffset) v = -timeBuffer[N + i - _synthesisOffset];
			else v = timeBuffer[i - _synthesisOffset];
			size_t i2 = output.pos + i;
			if (i2 >= _blockSamples) i2 -= _blockSamples;
			buffer[i2] += v*_synthesisWindow[i];
		}
	}

	void readOutput(size_t channel, size_t offset, size_t length, Sample *data) const {
		const Sample *buffer = output.buffer.data() + channel*_blockSamples;
		size_t start = (output.pos + offset)%_blockSamples;
		for (size_t i = 0; i < length; ++i) {
			size_t i2 = start + i;
			if (i2 >= _blockSamples) i2 -= _blockSamples;
			data[i] = buffer[i2]/output.windowProducts[i2];
		}
	}
	void readOutput(size_t channel, size_t length, Sample *data) const {
		readOutput(channel, 0, length, data);
	}
	// adds (already-normalised) samples into the output, so that readOutput() returns them added
	void addOutput(size_t channel, size_t offset, size_t length, const Sample *data) {
		Sample *buffer = output.buffer.data() + channel*_blockSamples;
		size_t start = (output.pos + offset)%_blockSamples;
		for (size_t i = 0; i < length; ++i) {
			size_t i2 = start + i;
			if (i2 >= _blockSamples) i2 -= _blockSamples;
			buffer[i2] += data[i]*output.windowProducts[i2];
		}
	}
	void addOutput(size_t channel, size_t length, const Sample *data) {
		addOutput(channel, 0, length, data);
	}
	void moveOutput(size_t samples) {
		if (samples == 1) { // hot: the reference calls this once per output sample (signalsmith-stretch.h:414)
			for (size_t c = 0; c < _outChannels; ++c) output.buffer[output.pos + c*_blockSamples] = 0;
			output.windowProducts[output.pos] = almostZero;
			if (++output.pos >= _blockSamples) output.pos = 0;
			return;
		}
		for (size_t i = 0; i < samples; ++i) {
			for (size_t c = 0; c < _outChannels; ++c) output.buffer[output.pos + c*_blockSamples] = 0;
			output.windowProducts[output.pos] = almostZero;
			if (++output.pos >= _blockSamples) output.pos = 0;
		}
	}
	// Running maximum of the window products from the read position onwards, so the un-overlapped tail
	// fades instead of being amplified ([upstream-recollection] only -- see SURVEY.md A.2; pinned by the
	// flush golden vector).
	void finishOutput(Sample strength = 1, size_t offset = 0) {
		Sample maxWindowProduct = 0;
		for (size_t i = offset; i < _blockSamples; ++i) {
			size_t i2 = output.pos + i;
			if (i2 >= _blockSamples) i2 -= _blockSamples;
			Sample &wp = output.windowProducts[i2];
			maxWindowProduct = std::max(wp, maxWindowProduct);
			wp += (maxWindowProduct - wp)*strength;
		}
	}

	const std::vector<Sample> &analysisWindow() const { return _analysisWindow; }
	const std::vector<Sample> &synthesisWindow() const { return _synthesisWindow; }

private:
	size_t _inChannels = 0, _outChannels = 0, _blockSamples = 0, _fftSamples = 0, _bands = 0;
	size_t _inputLength = 0, _defaultInterval = 0;
	size_t _analysisOffset = 0, _synthesisOffset = 0;
	std::vector<Sample> _analysisWindow, _synthesisWindow;
	std::vector<Complex> spectrumBuffer;
	std::vector<Sample> timeBuffer;
	std::vector<Complex> packed, packedOut, halfTwiddle;
	shim_detail::ComplexFFT<Sample> fft;

	static double bessel0(double x) {
		const double significanceLimit = 1e-4;
		double result = 0, term = 1, m = 0;
		while (term > significanceLimit) {
			result += term;
			++m;
			term *= (x*x)/(4*m*m);
		}
		return result;
	}

	void addWindowProduct() {
		for (size_t i = 0; i < _blockSamples; ++i) {
			size_t i2 = output.pos + i;
			if (i2 >= _blockSamples) i2 -= _blockSamples;
			output.windowProducts[i2] += _analysisWindow[i]*_synthesisWindow[i]*Sample(_fftSamples);
		}
	}

	// X[k] = sum_n x[n] exp(-2 pi i (k+1/2) n / N), k < N/2, via one N/2-point complex FFT
	void forwardModifiedReal(const Sample *time, Complex *bins) {
		size_t H = _bands, N = _fftSamples;
		for (size_t m = 0; m < H; ++m) {
			Complex u(time[m], -time[m + H]);
			packed[m] = shim_detail::ComplexFFT<Sample>::cmul(u, halfTwiddle[m]);
		}
		fft.template run<false>(packed.data(), packedOut.data());
		for (size_t k = 0; k < H; ++k) {
			if (k%2 == 0) bins[k] = packedOut[k/2];
			else bins[k] = std::conj(packedOut[(N - 1 - k)/2]);
		}
	}
	// inverse with gain N: time = N * x
	void inverseModifiedReal(const Complex *bins, Sample *time) {
		size_t H = _bands, N = _fftSamples;
		for (size_t j = 0; j < H; ++j) {
			size_t k = 2*j;
			if (k < H) packed[j] = bins[k];
			else packed[j] = std::conj(bins[N - 1 - k]);
		}
		fft.template run<true>(packed.data(), packedOut.data());
		for (size_t m = 0; m < H; ++m) {
			Complex v = shim_detail::ComplexFFT<Sample>::cmul(packedOut[m], std::conj(halfTwiddle[m]));
			time[m] = 2*v.real();
			time[m + H] = -2*v.imag();
		}
	}
};

}} // namespace
#endif
