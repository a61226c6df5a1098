// Host-side engine implementation.  See smst_engine.h.
#include "smst_engine.h"
#include "smst_switches.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>

namespace smst {

#define SMST_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw Error(std::string(#expr) + ": " + hipGetErrorString(e_), true); } while (0)

static constexpr float kNoiseFloor = 1e-15f;     // signalsmith-stretch.h:508
static constexpr float kMaxCleanStretch = 2.0f;  // :509

// fftSamples chosen by the L1 layer: 2*fastSizeAbove(ceil(block/2)) with fast sizes {4,5,6,8}*2^k
// (probe-verified for 5760/5292/4800/9600/11520).
static int fastSizeAbove(int size) {
	int power2 = 1;
	while (power2*8 < size) power2 *= 2;
	int multiple = (size + power2 - 1)/power2;
	if (multiple == 7) ++multiple;
	return multiple*power2;
}

static FftPlan makePlan(int H, int N) {
	FftPlan plan{};
	plan.H = H;
	plan.N = N;
	int rem = H, odd = 1;
	while (rem%3 == 0) { rem /= 3; odd *= 3; }
	while (rem%5 == 0) { rem /= 5; odd *= 5; }
	int log2 = 0;
	while (rem%2 == 0) { rem /= 2; ++log2; }
	if (rem != 1 || (odd != 1 && odd != 3 && odd != 5)) throw Error("unsupported FFT size (need 2^k * {1,3,5})");
	int n = 0;
	for (int i = 0; i < log2/2; ++i) plan.radix[n++] = 4;
	if (log2%2) plan.radix[n++] = 2;
	if (odd > 1) plan.radix[n++] = odd;
	if (n > kMaxFftPasses) throw Error("FFT too long");
	plan.npass = n;
	return plan;
}

// Kaiser window, heuristic-optimal bandwidth, forced to perfect reconstruction
static std::vector<float> makeWindow(int B, int I) {
	auto bessel0 = [](double x) {
		double result = 0, term = 1, m = 0;
		while (term > 1e-4) {
			result += term;
			++m;
			term *= (x*x)/(4*m*m);
		}
		return result;
	};
	std::vector<float> w(B);
	double bandwidth = double(B)/double(I);
	double bw = bandwidth + 8/((bandwidth + 3)*(bandwidth + 3)) + 0.25*std::max(3 - bandwidth, 0.0);
	bw = std::max(bw, 2.0);
	double beta = M_PI*std::sqrt(bw*bw*0.25 - 1);
	double invB0 = 1/bessel0(beta);
	for (int i = 0; i < B; ++i) {
		double r = (2*double(i) + 1)/double(B) - 1;
		double arg = std::sqrt(std::max(0.0, 1 - r*r));
		w[i] = float(bessel0(beta*arg)*invB0);
	}
	for (int j = 0; j < I && j < B; ++j) {
		float sum2 = 0;
		for (int i = j; i < B; i += I) sum2 += w[i]*w[i];
		float gain = 1/std::sqrt(sum2);
		for (int i = j; i < B; i += I) w[i] *= gain;
	}
	return w;
}

template <typename T> T *Batch::devAlloc(size_t count) {
	void *p = nullptr;
	if (count == 0) count = 1;
	SMST_HIP(hipMalloc(&p, count*sizeof(T)));
	allocations.push_back(p);
	++allocEvents;
	return static_cast<T *>(p);
}
template <typename T> T *Batch::pinnedAlloc(size_t count) {
	void *p = nullptr;
	if (count == 0) count = 1;
	SMST_HIP(hipHostMalloc(&p, count*sizeof(T), hipHostMallocDefault));
	pinned.push_back(p);
	++allocEvents;
	return static_cast<T *>(p);
}
void Batch::pinnedFree(void *p) {
	if (!p) return;
	auto it = std::find(pinned.begin(), pinned.end(), p);
	if (it != pinned.end()) pinned.erase(it);
	hipHostFree(p);
}
void Batch::devFree(void *p) {
	if (!p) return;
	auto it = std::find(allocations.begin(), allocations.end(), p);
	if (it != allocations.end()) allocations.erase(it);
	hipFree(p);
}

Batch::Batch(int streams, int channels, int block, int interval, bool splitComputation, int device, long seed, bool halfPrecisionState)
	: S(streams), C(channels), B(block), I(interval), split(splitComputation), halfState(halfPrecisionState), dev(device) {
	// geometry first: nothing below may throw before the HIP objects exist, and everything after is covered by the
	// clean-up in the catch block (a throwing constructor does not run the destructor)
	if (S < 1 || C < 1 || C > kMaxChannels || B < 4 || I < 1 || I > B) throw Error("invalid configuration (need 1.." + std::to_string(kMaxChannels) + " channels, interval <= block)");
	N = 2*fastSizeAbove((B + 1)/2);
	M = N/2;
	if (M > kMaxBands) throw Error("block too long: " + std::to_string(M) + " bands, at most " + std::to_string(kMaxBands) + " (one FFT buffer of bands*8 bytes must fit a CU's LDS; up to 9600 bands both buffers do, beyond that the second one lives in memory)");
	L = int(std::round(float(N)/float(I))); // longVerticalStep, signalsmith-stretch.h:636-637
	if (L < 1) L = 1;
	{
		int ring = 4;
		while (ring < L + 2) ring *= 2;
		if (ring > 64) throw Error("interval too small relative to the FFT size (vertical step too long)");
		// (9-16 channels, or a vertical step the fused kernels do not take: kChain keeps a ring of `ring` bins per channel and lane in LDS)
		if (C > kMaxFusedChannels && ((size_t)C*ring*64 + (size_t)C*128)*sizeof(float2) > (size_t)160*1024)
			throw Error("more than 8 channels need interval >= fftSamples/13 (the un-fused recurrence keeps a history ring per channel in LDS)");
	}
	const FftPlan plan = makePlan(M, N);
	try {
		construct(plan, seed);
	} catch (...) {
		releaseAll();
		throw;
	}
}

void Batch::construct(const FftPlan &plan, long seed) {
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	{ // the recurrence is latency-bound with few waves: give its stream the highest priority so its workgroups are
	  // dispatched ahead of the bulk kernels' when they share the machine (raising the analysis stream as well makes
	  // the recurrence wait for CUs: measured 1.9 ms instead of 1.2 ms per launch)
		int lo = 0, hi = 0;
		SMST_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
		SMST_HIP(hipStreamCreateWithPriority(&stChain, hipStreamNonBlocking, hi));
	}
	SMST_HIP(hipStreamCreateWithFlags(&stSynth, hipStreamNonBlocking));
	SMST_HIP(hipStreamCreateWithFlags(&stGate, hipStreamNonBlocking));
	for (int i = 0; i < 2; ++i) {
		SMST_HIP(hipEventCreateWithFlags(&callSets[i].done, hipEventDisableTiming));
		SMST_HIP(hipEventCreateWithFlags(&callSets[i].tables, hipEventDisableTiming));
	}
	SMST_HIP(hipEventCreateWithFlags(&evStart, hipEventDisableTiming));
	SMST_HIP(hipEventCreateWithFlags(&evOrder, hipEventDisableTiming));
	for (int i = 0; i < 3; ++i) {
		SMST_HIP(hipEventCreateWithFlags(&evFeed[i], hipEventDisableTiming));
		SMST_HIP(hipEventCreateWithFlags(&evChain[i], hipEventDisableTiming));
		SMST_HIP(hipEventCreateWithFlags(&evOut[i], hipEventDisableTiming));
		SMST_HIP(hipEventCreateWithFlags(&evSynth[i], hipEventDisableTiming));
	}
	// the cross-check switches: one struct, one table, read once (smst_switches.h)
	const Switches sw = Switches::fromEnvironment();
	overlap = sw.overlap; noFuse = sw.noFuse; noSingleHop = sw.noSingleHop; noAcross = sw.noAcross; carriedEmit = sw.carriedEmit != 0; checkLaunches = sw.checkLaunches; continuous = sw.continuous; contWriterWave = sw.contWriterWave;
	workspaceGiB = sw.workspaceGiB;
	subStreamsAsked = sw.subStreams;

	d.S = S; d.C = C; d.B = B; d.I = I; d.M = M; d.N = N; d.L = L; d.T = kTileHops;
	d.histLen = B + I;
	d.carryLen = B + I;
	d.delta = split ? I : 0;
	d.lag = L + 1;
	int ring = 4;
	while (ring < d.lag + 1) ring *= 2;
	d.ringSlots = ring;
	d.plan = plan;
	d.mapTableLen = 0;
	d.halfState = halfState ? 1 : 0;
	d.debugMode = sw.debugMode;
	d.noFeedFusion = sw.noFeedFusion;
	d.noStage = sw.noStage;
	d.vocNWide = sw.vocNWide;
	d.vocWide = sw.vocWide;
	d.vocNHalfLines = sw.vocNHalfLines;
	d.noAlign = sw.noAlign;
	d.alignAll = sw.alignAll;
	d.noFastFft = sw.noFastFft;
	// lean FFT tables (8-byte window entries + generated modulation, six stage twiddles instead of fifteen) are OPT-IN: they take 0.2 ms
	// off a 16.3-ms step, and their one extra rounding per element (spectra 1.2e-7 away from the full tables') flipped a peak decision
	// of a noise stream in tests/test_parity_gpu.py::test_batch_ragged -- parity first
	d.fftLean = sw.fftLean ? 1 : 0;
	d.feedSerial = sw.feedSerial;
	d.fftTeams = sw.fftTeams;
	d.synthEmit = sw.synthEmit;
	{
		int cus = 0;
		SMST_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
		d.teamsGrid = std::max(8, cus/8*8);
	}

	// constant tables
	std::vector<float2> tw(M), half(M), rot(M);
	for (int j = 0; j < M; ++j) {
		double a = -2*M_PI*double(j)/double(M);
		tw[j] = make_float2(float(std::cos(a)), float(std::sin(a)));
		double h = -M_PI*double(j)/double(N);
		half[j] = make_float2(float(std::cos(h)), float(std::sin(h)));
	}
	{ // hop rotation, evaluated with the reference's own fp32 recurrence (signalsmith-stretch.h:647-655)
		const float f0 = (0.0f + 0.5f)/float(N), f1 = (1.0f + 0.5f)/float(N);
		std::complex<float> r = std::polar(1.0f, f0*float(I)*float(2*M_PI));
		const float freqStep = f1 - f0;
		const std::complex<float> step = std::polar(1.0f, freqStep*float(I)*float(2*M_PI));
		for (int b = 0; b < M; ++b) {
			rot[b] = make_float2(r.real(), r.imag());
			r = std::complex<float>(r.real()*step.real() - r.imag()*step.imag(), r.real()*step.imag() + r.imag()*step.real());
		}
	}
	std::vector<float> win = makeWindow(B, I), wprod(B);
	for (int i = 0; i < B; ++i) wprod[i] = win[i]*win[i]*float(N);
	// reset(0.1) window-product seed: three phantom blocks at weight 0.1, read position one interval in
	{
		std::vector<float> wp(B, 0.0f);
		for (int i = 0; i < B; ++i) wp[i] += wprod[i];
		for (int i = B - I - 1; i >= 0; --i) wp[i] += wp[i + I];
		for (auto &v : wp) v = v*0.1f + 1e-30f;
		seedCarryWp.assign(d.carryLen, 1e-30f);
		for (int r = 0; r + I < B; ++r) seedCarryWp[r] = wp[r + I];
	}

	auto upload = [&](const void *src, size_t bytes) {
		void *p = devAlloc<unsigned char>(bytes);
		SMST_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
		return p;
	};
	d.twH = static_cast<float2 *>(upload(tw.data(), M*sizeof(float2)));
	d.halfTw = static_cast<float2 *>(upload(half.data(), M*sizeof(float2)));
	d.rot = static_cast<float2 *>(upload(rot.data(), M*sizeof(float2)));
	{
		std::vector<unsigned> pw(2*(size_t)M);
		unsigned long long x = 1;
		for (size_t j = 0; j < pw.size(); ++j) {
			x = x*16807ull % 2147483647ull;
			pw[j] = unsigned(x);
		}
		lcgHopJump = pw[2*(size_t)M - 3]; // 16807^(2M - 2): the draws of one randomised hop
		d.lcgPow = static_cast<unsigned *>(upload(pw.data(), pw.size()*sizeof(unsigned)));
	}
	d.window = static_cast<float *>(upload(win.data(), B*sizeof(float)));
	{ // folded analysis tables and, for H = 256*R3, the stage twiddles of the register-blocked FFT
		const int halfB = B/2;
		std::vector<float2> wa(M), wb(M);
		for (int m = 0; m < M; ++m) {
			float a = (m < B - halfB) ? win[m + halfB] : 0.0f;
			float b = (m >= M - halfB) ? win[m - M + halfB] : 0.0f;
			wa[m] = make_float2(a*half[m].x, a*half[m].y);        // w * e^{-i pi m/N}
			wb[m] = make_float2(-b*half[m].y, b*half[m].x);       // i * w * e^{-i pi m/N}
		}
		d.winA = static_cast<float2 *>(upload(wa.data(), M*sizeof(float2)));
		d.winB = static_cast<float2 *>(upload(wb.data(), M*sizeof(float2)));
		std::vector<float4> w4(M), st4(M);
		for (int m = 0; m < M; ++m) {
			w4[m] = make_float4(wa[m].x, wa[m].y, wb[m].x, wb[m].y);
			st4[m] = make_float4(half[m].x, half[m].y, (m < B - halfB) ? win[m + halfB] : 0.0f, (m >= M - halfB) ? win[m - M + halfB] : 0.0f);
		}
		d.win4 = static_cast<float4 *>(upload(w4.data(), M*sizeof(float4)));
		d.synTab = static_cast<float4 *>(upload(st4.data(), M*sizeof(float4)));
		{ // lean tables: the window samples alone (the modulation e^{-i pi m/N} is generated from halfTw[t] and constants)
			std::vector<float2> w2(M), s2(M);
			for (int m = 0; m < M; ++m) {
				const float a = (m < B - halfB) ? win[m + halfB] : 0.0f, b = (m >= M - halfB) ? win[m - M + halfB] : 0.0f;
				w2[m] = make_float2(a, b);
				s2[m] = make_float2(a, b);
			}
			d.win2 = static_cast<float2 *>(upload(w2.data(), M*sizeof(float2)));
			d.syn2 = static_cast<float2 *>(upload(s2.data(), M*sizeof(float2)));
		}
		d.twA = d.twB = nullptr;
		d.twA4 = d.twB4 = d.twA6 = nullptr;
		if (M%256 == 0 && (M/256 == 10 || M/256 == 12 || M/256 == 20 || M/256 == 24)) { // the register-blocked FFT's sizes: 16 x 16 x R3
			const int R3 = M/256, MA = 16*R3;
			std::vector<float2> ta((size_t)15*MA), tb((size_t)15*R3);
			for (int n = 1; n < 16; ++n) {
				for (int p = 0; p < MA; ++p) {
					double a = -2*M_PI*double(p*n)/double(M);
					ta[(size_t)(n - 1)*MA + p] = make_float2(float(std::cos(a)), float(std::sin(a)));
				}
				for (int p = 0; p < R3; ++p) {
					double a = -2*M_PI*double(p*n)/double(MA);
					tb[(size_t)(n - 1)*R3 + p] = make_float2(float(std::cos(a)), float(std::sin(a)));
				}
			}
			d.twA = static_cast<float2 *>(upload(ta.data(), ta.size()*sizeof(float2)));
			d.twB = static_cast<float2 *>(upload(tb.data(), tb.size()*sizeof(float2)));
			std::vector<float4> ta4((size_t)8*MA), tb4((size_t)8*R3);
			for (int i = 0; i < 8; ++i) {
				for (int p = 0; p < MA; ++p) {
					const float2 lo = ta[(size_t)(2*i)*MA + p], hi = (2*i + 1 < 15) ? ta[(size_t)(2*i + 1)*MA + p] : make_float2(1.f, 0.f);
					ta4[(size_t)i*MA + p] = make_float4(lo.x, lo.y, hi.x, hi.y);
				}
				for (int p = 0; p < R3; ++p) {
					const float2 lo = tb[(size_t)(2*i)*R3 + p], hi = (2*i + 1 < 15) ? tb[(size_t)(2*i + 1)*R3 + p] : make_float2(1.f, 0.f);
					tb4[(size_t)i*R3 + p] = make_float4(lo.x, lo.y, hi.x, hi.y);
				}
			}
			std::vector<float4> ta6((size_t)3*MA);
			for (int p = 0; p < MA; ++p) {
				auto tw = [&](int n) { return ta[(size_t)(n - 1)*MA + p]; };
				ta6[p] = make_float4(tw(1).x, tw(1).y, tw(2).x, tw(2).y);
				ta6[(size_t)MA + p] = make_float4(tw(3).x, tw(3).y, tw(4).x, tw(4).y);
				ta6[(size_t)2*MA + p] = make_float4(tw(8).x, tw(8).y, tw(12).x, tw(12).y);
			}
			d.twA6 = static_cast<float4 *>(upload(ta6.data(), ta6.size()*sizeof(float4)));
			d.twA4 = static_cast<float4 *>(upload(ta4.data(), ta4.size()*sizeof(float4)));
			d.twB4 = static_cast<float4 *>(upload(tb4.data(), tb4.size()*sizeof(float4)));
		}
	}
	d.wprod = static_cast<float *>(upload(wprod.data(), B*sizeof(float)));

	const size_t bandRows = (size_t)S*C*M;
	d.stInput = devAlloc<float2>(bandRows);
	d.stPrev = devAlloc<float2>(bandRows);
	// carried state that the fp16 option narrows (the typed pointers then address half-sized allocations; every access goes
	// through the accessors at the top of smst_kernels_common.h)
	const size_t stateScale = halfState ? 2 : 1;
	d.stOut = reinterpret_cast<float2 *>(devAlloc<unsigned char>(bandRows*sizeof(float2)/stateScale));
	d.stEnergy = reinterpret_cast<float *>(devAlloc<unsigned char>(bandRows*sizeof(float)/stateScale + 16)); // (+16: PrevEnergy::at reads 8 bytes at an element)
	SMST_HIP(hipMemset(d.stEnergy, 0, bandRows*sizeof(float)/stateScale + 16));
	d.histPitch = 2*d.histLen;
	d.carryPitch = d.carryLen + 2*I;
	d.hist = devAlloc<float>((size_t)S*C*d.histPitch);
	dZeroEmit = devAlloc<EmitDesc>(S); // (settleCarry)
	SMST_HIP(hipMemset(dZeroEmit, 0, (size_t)S*sizeof(EmitDesc)));
	for (int h = 0; h < 2; ++h) {
		d.histBase[h] = devAlloc<int>((size_t)S);
		SMST_HIP(hipMemset(d.histBase[h], 0, (size_t)S*sizeof(int)));
		d.carrySum[h] = reinterpret_cast<float *>(devAlloc<unsigned char>((size_t)S*C*d.carryPitch*sizeof(float)/stateScale));
		d.carryWp[h] = devAlloc<float>((size_t)S*d.carryPitch);
		d.carryBase[h] = devAlloc<int>((size_t)S);
		SMST_HIP(hipMemset(d.carryBase[h], 0, (size_t)S*sizeof(int)));
	}
	d.wpHeadLen = ((B + I - 1)/I + 2)*I;
	d.wpHead = devAlloc<float>((size_t)S*d.wpHeadLen);
	d.stFreq = devAlloc<float>((size_t)S*2);
	dParams = devAlloc<StreamParams>(S);
	d.params = d.paramsPeaks = d.paramsForm0 = d.paramsForm2 = dParams;
	dEnergy = devAlloc<float>((size_t)S*kEnergyParts);
	for (int i = 0; i < 2; ++i) {
		callSets[i].inSamples = devAlloc<int>(S);
		callSets[i].outSamples = devAlloc<int>(S);
		callSets[i].flags = devAlloc<int>(S);
		callSets[i].hInSamples = pinnedAlloc<int>(S);
		callSets[i].hOutSamples = pinnedAlloc<int>(S);
		callSets[i].hFlags = pinnedAlloc<int>(S);
		callSets[i].hEnergy = pinnedAlloc<float>((size_t)S*kEnergyParts);
		callSets[i].resetBits = devAlloc<int>(S);
		callSets[i].hResetBits = pinnedAlloc<int>(S);
		if (split) {
			callSets[i].pendHops = devAlloc<HopDesc>(S);
			callSets[i].hPendHops = pinnedAlloc<HopDesc>(S);
		}
	}
	dSeedWp = devAlloc<float>(d.carryLen);
	SMST_HIP(hipMemcpy(dSeedWp, seedCarryWp.data(), d.carryLen*sizeof(float), hipMemcpyHostToDevice));
	hopFirst.assign(S, 0);
	hopCount.assign(S, 0);
	leavesPendingV.assign(S, 0);
	ridesV.assign(S, 0);
	passV.assign(S, 0);
	dInSamples = callSets[0].inSamples;
	dOutSamples = callSets[0].outSamples;
	dFlags = callSets[0].flags;
	dAux0 = devAlloc<int>(S);
	dAux1 = devAlloc<int>(S);
	dResetBits = devAlloc<int>(S);
	dKeep = devAlloc<int>(S);
	resetBitsV.assign(S, 0);

	sched.assign(S, StreamSched());
	lastHop.assign(S, LastHop());
	// std::default_random_engine (libstdc++: minstd_rand0) of a reference instance constructed with seed + s (:39; smst_kernels_common.h: engineDraw)
	for (int s = 0; s < S; ++s) {
		const unsigned long long u = (unsigned long long)(seed + s); // `long` -> the engine's unsigned 64-bit result_type
		sched[s].seed = unsigned(u % 2147483647ull);
		if (sched[s].seed == 0) sched[s].seed = 1;
	}
	StreamParams p{};
	p.freqMultiplier = 1; p.freqTonalityLimit = 0.5f; // :513
	p.formantMultiplier = 1; p.invFormantMultiplier = 1; p.formantBaseFreq = 0; p.formantCompensation = 0; p.hasCustomMap = 0;
	params.assign(S, p);
	paramsDirty = true;

	allocateWorkspace();
	pend.assign(S, PendingBlock());
	lastSteps.assign(S, 0);
	lastStarts.assign(S, 0);
	histBase.assign(S, 0);
	carryBase.assign(S, 0);
	keepV.assign(S, 0);
	if (split) { // the block in flight: its spectra, and the tables of the run that completes it (PendingBlock, smst_engine.h)
		const size_t rows = (size_t)S*C*d.Mp;
		dPendIn = devAlloc<float2>(rows);
		dPendPrev = devAlloc<float2>(rows);
		SMST_HIP(hipMemset(dPendIn, 0, rows*sizeof(float2)));
		SMST_HIP(hipMemset(dPendPrev, 0, rows*sizeof(float2)));
		const int nSub = (S + subS - 1)/subS;
		for (int i = 0; i < 2; ++i) {
			PendSet &ps = pendSets[i];
			ps.hops = devAlloc<HopDesc>(S + kTileHops); ps.hHops = pinnedAlloc<HopDesc>(S); // (+ a tile's worth: the wavefront kernels cache a blind row of 64 descriptors per stream, hopStride 1)
			SMST_HIP(hipMemset(ps.hops, 0, (size_t)(S + kTileHops)*sizeof(HopDesc)));
			ps.emit = devAlloc<EmitDesc>(S); ps.hEmit = pinnedAlloc<EmitDesc>(S);
			ps.tileInfo = devAlloc<int>((size_t)nSub*2*subS); ps.hTileInfo = pinnedAlloc<int>((size_t)nSub*2*subS);
			ps.bits = devAlloc<int>(S); ps.hBits = pinnedAlloc<int>(S);
			ps.synthChannels = devAlloc<int>(S); ps.hSynthChannels = pinnedAlloc<int>(S);
			for (int j = 0; j < 3; ++j) { ps.prm[j] = devAlloc<StreamParams>(S); ps.hPrm[j] = pinnedAlloc<StreamParams>(S); }
			SMST_HIP(hipEventCreateWithFlags(&ps.done, hipEventDisableTiming));
			ps.used = false;
		}
		dZeroCounts = devAlloc<int>(S);
		SMST_HIP(hipMemset(dZeroCounts, 0, S*sizeof(int)));
		pendList.reserve(S);
		pendTileHas.assign((size_t)nSub*kTileHasStride, 0);
		pendMaxSpan.assign(nSub, 0);
	}
	SMST_HIP(hipDeviceSynchronize()); // the hipMemset calls above are ordered with the null stream only; the engine's streams are non-blocking
	reset();
}

Batch::~Batch() { releaseAll(); }

void Batch::releaseAll() {
	hipSetDevice(dev);
	if (st) hipStreamSynchronize(st);
	if (stChain) hipStreamSynchronize(stChain);
	if (stSynth) hipStreamSynchronize(stSynth);
	if (stGate) hipStreamSynchronize(stGate);
	for (auto &e : livePool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
	livePool.clear();
	liveEvents.clear();
	for (void *p : allocations) hipFree(p);
	allocations.clear();
	for (void *p : pinned) hipHostFree(p);
	pinned.clear();
	for (int i = 0; i < 2; ++i) {
		if (callSets[i].done) hipEventDestroy(callSets[i].done);
		if (callSets[i].tables) hipEventDestroy(callSets[i].tables);
		callSets[i].done = callSets[i].tables = nullptr;
		if (pendSets[i].done) hipEventDestroy(pendSets[i].done);
		pendSets[i].done = nullptr;
	}
	if (evStart) hipEventDestroy(evStart);
	if (evOrder) hipEventDestroy(evOrder);
	evStart = evOrder = nullptr;
	for (int i = 0; i < 3; ++i) {
		if (evFeed[i]) hipEventDestroy(evFeed[i]);
		if (evChain[i]) hipEventDestroy(evChain[i]);
		if (evOut[i]) hipEventDestroy(evOut[i]);
		if (evSynth[i]) hipEventDestroy(evSynth[i]);
		evFeed[i] = evChain[i] = evOut[i] = evSynth[i] = nullptr;
	}
	if (stGate) hipStreamDestroy(stGate);
	if (stChain) hipStreamDestroy(stChain);
	if (stSynth) hipStreamDestroy(stSynth);
	if (st) hipStreamDestroy(st);
	st = stChain = stSynth = stGate = nullptr;
}

void Batch::allocateWorkspace() {
	// Per (stream, hop, channel): 3 complex rows + 1 float4 row of M bins + one B-sample frame, plus (3-8 channels only) the
	// skewed records and (SMST_FEED_SERIAL only) the serial feed's scratch.  Sub-batch the streams so one tile workspace
	// stays under a budget.  Default: a third of the HBM that is free when the batch is created, per workspace (there are
	// two), at most 96 GiB and never more than 40 % of what is free -- fewer, larger sub-batches keep the one-wave-per-stream
	// kernels of the 3-8 channel path at more than one wave per CU (config 5: 2 sub-batches instead of 7).  If an
	// allocation still fails (another batch or torch took the memory meanwhile) the sub-batch is halved and tried again.
	double budgetGiB = 24;
	{
		size_t freeB = 0, totalB = 0;
		if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB > 0) {
			const double freeGiB = double(freeB)/(1024.0*1024.0*1024.0);
			budgetGiB = std::min(std::min(96.0, std::max(8.0, freeGiB/3.0)), 0.4*freeGiB);
		}
	}
	if (workspaceGiB > 0) budgetGiB = std::max(0.0005, workspaceGiB); // SMST_WORKSPACE_GIB (smst_switches.h)
	const bool needRecords = !fusedSupported(d) || noFuse; // the fused recurrence keeps its records in LDS
	const size_t recChunks = (size_t(recordFloats(C)) + 3)/4;
	d.recSteps = ((M + d.lag*(d.T - 1) + 63)/64)*64 + 8;
	d.recPitch = int(recChunks*64 + 16);
	d.Mp = (M + 32 + 15) & ~15; // rows start on 128-byte lines (the recurrence's writer stores aligned 64-byte groups)
	const size_t perStream = (size_t)d.T*((size_t)C*((size_t)d.Mp*(3*sizeof(float2) + sizeof(PredEntry)) + (size_t)B*sizeof(float))
	                                      + (size_t)M*(sizeof(float2) + sizeof(float)) + 3*sizeof(float))
	                         + (size_t)C*64*sizeof(float2)
	                         + (needRecords ? (size_t)d.recSteps*d.recPitch*sizeof(float4) : 0)
	                         + (d.feedSerial ? (size_t)M*64*2*sizeof(float) + (size_t)(M/2 + 2)*64*sizeof(float2) : 0);
	size_t maxStreams = size_t(budgetGiB*1024.0*1024.0*1024.0/double(perStream));
	if (maxStreams < 1) maxStreams = 1;
	subS = int(std::min<size_t>(S, maxStreams));
	if (subStreamsAsked > 0) subS = std::min(subS, subStreamsAsked);
	for (;;) {
		const size_t mark = allocations.size();
		try {
			const size_t rows = (size_t)subS*d.T*C*d.Mp;
			// two complete tile workspaces: tile i+1's feed-forward kernels run while tile i is still in the recurrence /
			// synthesis (the budget above is per workspace)
			for (int i = 0; i < 2; ++i) {
				TileBuffers &w = slots[i];
				w = TileBuffers{};
				w.Xcur = devAlloc<float2>(rows);
				w.Xprev = devAlloc<float2>(rows);
				w.PE = devAlloc<PredEntry>(rows + 2); // (+2 entries: PrevEnergy::at reads 8 bytes at an entry's third float)
				w.OUT = devAlloc<float2>(rows);
				if (needRecords) {
					w.REC = devAlloc<float4>((size_t)subS*d.recSteps*d.recPitch);
					SMST_HIP(hipMemset(w.REC, 0, (size_t)subS*d.recSteps*d.recPitch*sizeof(float4)));
				}
				w.dump = devAlloc<float2>((size_t)subS*C*64);
				w.map = devAlloc<float2>((size_t)subS*d.T*M);
				w.ratio = devAlloc<float>((size_t)subS*d.T*M);
				if (d.noFeedFusion) w.envelope = devAlloc<float>((size_t)subS*d.T*M); // (test hook: smst_batch_debug_get_formants)
				if (d.feedSerial) {
					w.energyT = devAlloc<float>((size_t)subS*M*64);
					w.smoothT = devAlloc<float>((size_t)subS*M*64);
					w.peaksT = devAlloc<float2>((size_t)subS*(M/2 + 2)*64);
				}
				w.est = devAlloc<float>((size_t)subS*d.T*2);
				w.freqEst = devAlloc<float>((size_t)subS*d.T);
				w.frames = devAlloc<float>((size_t)subS*d.T*C*B);
				if (fftNeedsScratch(M)) w.fftScratch = devAlloc<float2>(rows); // (the synthesis frames' second FFT buffer: smst_device.h)
			}
			// The continuous wavefront (kVocoderCont) finishes tile t in launch t+1, so tile t+1's analysis needs a third place to write to:
			// the spectra, the results and the frames of a PLAIN tile only, and only where that form can run at all (one sub-batch)
			slots[2] = TileBuffers{};
			dContSave = nullptr;
			if (continuous && continuousSupported(d) && fusedSupported(d) && !noFuse && subS == S) {
				TileBuffers &w = slots[2];
				w.Xcur = devAlloc<float2>(rows);
				w.Xprev = devAlloc<float2>(rows);
				w.OUT = devAlloc<float2>(rows);
				w.frames = devAlloc<float>((size_t)subS*d.T*C*B);
				if (fftNeedsScratch(M)) w.fftScratch = devAlloc<float2>(rows);
				dContSave = devAlloc<float2>((size_t)S*8*C*64);
			}
			break;
		} catch (const Error &) {
			while (allocations.size() > mark) { hipFree(allocations.back()); allocations.pop_back(); }
			(void)hipGetLastError();
			if (subS <= 1) throw;
			subS = (subS + 1)/2;
		}
	}
	wsBytes = 2*perStream*subS + (slots[2].Xcur ? (size_t)subS*d.T*C*((size_t)d.Mp*3*sizeof(float2) + (size_t)B*sizeof(float)) : 0);
}

void Batch::uploadParams() {
	if (!paramsDirty) return;
	SMST_HIP(hipMemcpyAsync(dParams, params.data(), S*sizeof(StreamParams), hipMemcpyHostToDevice, st));
	SMST_HIP(hipStreamSynchronize(st));
	paramsDirty = false;
}

// stft.reset(0.1) / Band clearing for a set of streams (bit masks, see kResetStreams): one upload + one launch, instead
// of six API calls per stream.  `bitsHost` == nullptr: `allBits` for every stream.  Not on the steady-state path (reset,
// flush, first silent block), so the small synchronous upload is fine.
// The kernels outside the tile pipeline (flush tail, pre-roll, the reset that keeps the samples of a split interval, the debug accessors) index
// the overlap-add carry from the front of its rows: a call without hops may have left the windows further in (kEmitCarried) -- one
// emission of nothing moves them back.
void Batch::settleCarry() {
	bool moved = false;
	for (int s = 0; s < S && !moved; ++s) moved = carryBase[s] != 0;
	if (!moved) return;
	DevBatch dd = d;
	dd.emit = dZeroEmit;
	dd.emitStride = 1;
	const IoArgs io{nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr};
	launchEmit(dd, io, 0, S, 0, 0, st);
	d.carryCur ^= 1;
	std::fill(carryBase.begin(), carryBase.end(), 0);
}

void Batch::resetStreams(const int *bitsHost, int allBits, const int *keepHost) {
	if (keepHost) settleCarry();
	if (bitsHost) SMST_HIP(hipMemcpy(dResetBits, bitsHost, S*sizeof(int), hipMemcpyHostToDevice));
	if (keepHost) SMST_HIP(hipMemcpy(dKeep, keepHost, S*sizeof(int), hipMemcpyHostToDevice));
	launchResetStreams(d, bitsHost ? dResetBits : nullptr, allBits, dSeedWp, st, keepHost ? dKeep : nullptr);
	for (int s = 0; s < S; ++s) if ((bitsHost ? bitsHost[s] : allBits) & 1) histBase[s] = carryBase[s] = 0; // (the host's copies of DevBatch::histBase / carryBase)
}

void Batch::reset() { // signalsmith-stretch.h:49-60
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipMemsetAsync(d.stFreq, 0, (size_t)S*2*sizeof(float), st));
	resetStreams(nullptr, 1 | 2 | 4 | 8);
	for (auto &lh : lastHop) lh = LastHop();
	std::fill(lastSteps.begin(), lastSteps.end(), 0);
	std::fill(lastStarts.begin(), lastStarts.end(), 0);
	d.histCur = 0;
	d.carryCur = 0;
	for (int s = 0; s < S; ++s) {
		const unsigned seed = seedAfterDroppedBlock(s); // reset() leaves the reference's randomEngine alone; a block in flight is dropped with the draws it has made
		pend[s] = PendingBlock(); // blockProcess = {}
		sched[s] = StreamSched();
		sched[s].seed = seed;
	}
}

// ---- parameters ------------------------------------------------------------------------------------------
template <typename F> static void forStreams(int S, int stream, F &&f) {
	if (stream < 0) { for (int s = 0; s < S; ++s) f(s); }
	else if (stream < S) f(stream);
	else throw Error("stream index out of range");
}
void Batch::setTransposeFactor(int stream, float multiplier, float tonalityLimit) { // :107-115
	forStreams(S, stream, [&](int s) {
		freezePendingParams(s);
		params[s].freqMultiplier = multiplier;
		params[s].freqTonalityLimit = (tonalityLimit > 0) ? tonalityLimit/std::sqrt(multiplier) : 1.0f;
		params[s].hasCustomMap = 0;
	});
	paramsDirty = true;
}
void Batch::setTransposeSemitones(int stream, float semitones, float tonalityLimit) { // :116-118
	setTransposeFactor(stream, float(std::pow(2, semitones/12)), tonalityLimit);
}
void Batch::setFormantFactor(int stream, float multiplier, bool compensatePitch) { // :124-128
	forStreams(S, stream, [&](int s) {
		freezePendingParams(s);
		params[s].formantMultiplier = multiplier;
		params[s].invFormantMultiplier = 1/multiplier;
		params[s].formantCompensation = compensatePitch ? 1 : 0;
	});
	paramsDirty = true;
}
void Batch::setFormantSemitones(int stream, float semitones, bool compensatePitch) { // :129-131
	setFormantFactor(stream, float(std::pow(2, semitones/12)), compensatePitch);
}
void Batch::setFormantBase(int stream, float baseFreq) { // :133-135
	forStreams(S, stream, [&](int s) { freezePendingParams(s); params[s].formantBaseFreq = baseFreq; });
	paramsDirty = true;
}
void Batch::setFreqMapTable(int stream, const float *table, int n) { // table form of :120-122
	// Split computation: the steps of a block in flight that have already run keep the map they saw -- flag, length AND knots: every stream
	// has kMapSlots table rows, StreamParams.mapSlot names the live one, and a new table goes to a row that neither findPeaks' (:874) nor
	// updateFormants step 2's (:1020) latched parameters of the block in flight refer to (freezePendingParams).  The reference replaces its
	// std::function as a whole (:120-122): a step that ran before the call evaluated the old function, a step that runs after it the new one.
	forStreams(S, stream, [&](int s) { freezePendingParams(s); });
	if (n <= 0 || !table) {
		forStreams(S, stream, [&](int s) { params[s].hasCustomMap = 0; });
		paramsDirty = true;
		return;
	}
	if (n < 2) throw Error("frequency-map table needs at least 2 points");
	if (stream >= S) throw Error("stream index out of range");
	SMST_HIP(hipSetDevice(dev));
	// Every stream keeps ITS OWN table, knot for knot (StreamParams.mapLen points, linear between them, extrapolated beyond: the
	// reference evaluates each instance's own function, :874, :1020); the batch's array only has one row pitch, the longest table
	// seen so far -- a longer table makes the array grow and the existing rows are copied as they are.  (Rounds 2-3 re-evaluated the
	// stored rows on the longer table's grid, which cuts the corner at every old knot: a stream's map could change because ANOTHER
	// stream was given a longer table.)  When no stream holds a table any more the next one starts afresh.
	auto latched = [&](int s, int slot) { // a step of the stream's block in flight ran with the table in this row
		if (!split || !pend[s].valid) return false;
		const PendingBlock &pb = pend[s];
		return (pb.frozenPeaks && pb.peaks.hasCustomMap && pb.peaks.mapSlot == slot) || (pb.frozenForm2 && pb.form2.hasCustomMap && pb.form2.mapSlot == slot);
	};
	bool anyCustom = false;
	for (int s = 0; s < S; ++s) {
		anyCustom = anyCustom || params[s].hasCustomMap;
		for (int slot = 0; slot < kMapSlots; ++slot) anyCustom = anyCustom || latched(s, slot);
	}
	if (!anyCustom) d.mapTableLen = 0;
	if (n > d.mapTableLen) {
		std::vector<float> grown((size_t)S*kMapSlots*n, 0.0f);
		for (size_t row = 0; row < (size_t)S*kMapSlots && d.mapTableLen > 0; ++row) // (rows that hold no table are zeros either way)
			std::copy(hostMapTable.begin() + row*d.mapTableLen, hostMapTable.begin() + (row + 1)*d.mapTableLen, grown.begin() + row*n);
		SMST_HIP(hipStreamSynchronize(st)); // kernels of earlier calls may still read the old array
		if (dMapTable) devFree(dMapTable);
		dMapTable = devAlloc<float>((size_t)S*kMapSlots*n);
		hostMapTable.swap(grown);
		d.mapTableLen = n;
		d.mapTable = dMapTable;
	}
	const int pitch = d.mapTableLen;
	forStreams(S, stream, [&](int s) {
		int slot = 0;
		while (slot < kMapSlots - 1 && latched(s, slot)) ++slot; // at most two rows are latched: one is always free
		std::copy(table, table + n, hostMapTable.begin() + ((size_t)s*kMapSlots + slot)*pitch);
		params[s].hasCustomMap = 1;
		params[s].mapLen = n;
		params[s].mapSlot = slot;
	});
	SMST_HIP(hipStreamSynchronize(st)); // kernels of earlier calls may still read the old table
	SMST_HIP(hipMemcpy(dMapTable, hostMapTable.data(), hostMapTable.size()*sizeof(float), hipMemcpyHostToDevice));
	paramsDirty = true;
}

void Batch::checkLaunch(const char *what) {
	if (!checkLaunches) return;
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess) throw Error(std::string("launch failed (") + what + "): " + hipGetErrorString(e), true);
}

template <typename F> void Batch::timed(double &acc, F &&f) {
	if (!profiling) { f(); return; }
	hipEvent_t a, b;
	SMST_HIP(hipEventCreate(&a));
	SMST_HIP(hipEventCreate(&b));
	SMST_HIP(hipEventRecord(a, st));
	f();
	SMST_HIP(hipEventRecord(b, st));
	SMST_HIP(hipEventSynchronize(b));
	float ms = 0;
	SMST_HIP(hipEventElapsedTime(&ms, a, b));
	acc += ms;
	hipEventDestroy(a);
	hipEventDestroy(b);
}
void Batch::growLivePool(size_t pairs) {
	// timing events WITHOUT the default system-scope fences: those drain and flush at every record and slowed the
	// recurrence from 1.2 to 1.7 ms per launch (and the step by 7 %)
	while (livePool.size() < pairs) {
		hipEvent_t a = nullptr, b = nullptr;
		SMST_HIP(hipEventCreateWithFlags(&a, hipEventDisableSystemFence));
		SMST_HIP(hipEventCreateWithFlags(&b, hipEventDisableSystemFence));
		livePool.emplace_back(a, b);
	}
}
void Batch::enableProfiling(int mode) {
	profiling = mode == 1;
	liveTiming = mode == 2;
	if (liveTiming) {
		SMST_HIP(hipSetDevice(dev));
		growLivePool(1024); // 128 steps of the headline workload between two takeTimings(): no event is created inside process()
	}
}
BatchTimings Batch::takeTimings() {
	if (!liveEvents.empty()) {
		SMST_HIP(hipSetDevice(dev));
		SMST_HIP(hipStreamSynchronize(st));
		for (auto &e : liveEvents) {
			float ms = 0;
			SMST_HIP(hipEventElapsedTime(&ms, e.first, e.second));
			timings.chainLiveMs += ms;
			++timings.chainLiveLaunches;
		}
		liveEvents.clear(); // the pairs go back to the pool
	}
	BatchTimings t = timings;
	timings = BatchTimings();
	return t;
}
void Batch::synchronize() {
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipStreamSynchronize(st));
}

// ---- process ----------------------------------------------------------------------------------------------
template <typename T> static void ensureSize(std::vector<T> &v, size_t n, long &events) {
	if (n > v.capacity()) ++events;
	v.resize(n);
}

void Batch::waitForStream(hipStream_t other) {
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipEventRecord(evOrder, other));
	SMST_HIP(hipStreamWaitEvent(st, evOrder, 0));
	SMST_HIP(hipStreamWaitEvent(stGate, evOrder, 0));
}
void Batch::signalStream(hipStream_t other) {
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipEventRecord(evOrder, st));
	SMST_HIP(hipStreamWaitEvent(other, evOrder, 0));
}

// ---- split computation: the reference's step schedule -----------------------------------------------------
StepLayout Batch::stepLayout(unsigned flags) const { // the order of signalsmith-stretch.h:332-400 / :642-812, counted as :304-318 / :620-632 count it
	StepLayout l;
	int i = 0;
	const bool nw = (flags & HOP_NEW_SPECTRUM) != 0;
	if (nw) {
		if (flags & HOP_REANALYSE_PREV) { l.reanalyse0 = i; i += C + 1; } // analyseSteps() = one per channel (pinned by tests/golden/split_events), + the copy
		l.analyse0 = i; i += C;
		l.copyIn = i; i += 1;
		i += C; // rotation
	}
	if (flags & HOP_MAPPED) { i += 3; l.peaks = i; i += 1; } // smoothEnergy x 3, findPeaks
	i += 1;                                                   // output map
	if (flags & HOP_FORMANTS) { l.form0 = i; l.form2 = i + 2; i += 3; }
	i += C;                                                   // preliminary prediction
	l.main0 = i; i += 8;                                     // splitMainPrediction
	if (nw) { l.prevCopy = i; i += 1; }
	l.spectrum = i; i += 1;
	l.synth0 = i; i += C;                                     // synthesiseSteps() = one per channel, the first adds the window product
	l.steps = i;
	return l;
}
size_t Batch::stepsExecuted(size_t steps, size_t k) const { // after k samples of the interval, :321-325 (fp32 as there)
	const float processRatio = float(k)/float(size_t(I));
	return std::min<size_t>(steps, size_t((float(steps) + 0.999f)*processRatio));
}
// The random engine and the block in flight (:616, :749, :769): a block beyond 2x draws its 2M - 2 time factors inside the chunks of the main
// prediction, in bin order, WHEN those chunks run.  So the stream's engine moves on by a whole block when the block in flight finally runs, and
// -- if reset(), a silent call or configure() drops it (blockProcess = {}) -- by the draws of the chunks that had run by then only.
static unsigned lcgPower(unsigned long long n) { // 16807^n mod (2^31 - 1)
	unsigned long long r = 1, b = 16807;
	const unsigned long long m = 2147483647ull;
	while (n) {
		if (n & 1) r = r*b % m;
		b = b*b % m;
		n >>= 1;
	}
	return unsigned(r);
}
unsigned Batch::seedAfterDroppedBlock(int s) const {
	const PendingBlock &pb = pend[s];
	const unsigned seed = sched[s].seed;
	if (!split || !pb.valid || !(pb.flags & HOP_RANDOM_TF)) return seed;
	const StepLayout l = stepLayout(pb.flags);
	const size_t e = stepsExecuted(size_t(l.steps), sched[s].samplesSinceLast);
	const size_t chunks = e > size_t(l.main0) ? std::min<size_t>(8, e - size_t(l.main0)) : 0;
	const size_t b0 = size_t(M)*chunks/8; // bins [0, b0) have drawn: one draw per bin for b > 0, one for b < M - 1
	if (b0 == 0) return seed;
	const unsigned long long draws = (b0 >= size_t(M)) ? 2ull*M - 2 : 2ull*b0 - 1;
	return unsigned((unsigned long long)seed*lcgPower(draws) % 2147483647ull);
}

// A setter is about to change params[s]: the steps of the block in flight that have already run saw the OLD values (:874, :982, :1020)
void Batch::freezePendingParams(int s) {
	if (!split) return;
	PendingBlock &pb = pend[s];
	if (!pb.valid) return;
	const StepLayout l = stepLayout(pb.flags);
	const size_t e = stepsExecuted(size_t(l.steps), sched[s].samplesSinceLast);
	if (l.peaks >= 0 && e > size_t(l.peaks) && !pb.frozenPeaks) { pb.peaks = params[s]; pb.frozenPeaks = true; }
	if (l.form0 >= 0 && e > size_t(l.form0) && !pb.frozenForm0) { pb.form0 = params[s]; pb.frozenForm0 = true; }
	if (l.form2 >= 0 && e > size_t(l.form2) && !pb.frozenForm2) { pb.form2 = params[s]; pb.frozenForm2 = true; }
}

// The blocks in flight of the streams in `pendList` run to their end: one tile of one hop per stream whose spectra come from the
// pending buffers; it consumes no input and emits no sample -- the frame goes into the overlap-add ring at the point where the block's
// interval ends (outPos = -samples since the block began; split computation delays every frame by one interval, :292-296).
// synthChannels (host, per stream, may be null): a flush() caught the block between two synthesis steps (:397-399).
void Batch::runPendingBlocks(const int *synthChannels) {
	if (pendList.empty()) return;
	uploadParams();
	pendCur ^= 1;
	PendSet &ps = pendSets[pendCur];
	if (ps.used) SMST_HIP(hipEventSynchronize(ps.done));
	const int nSub = (S + subS - 1)/subS;
	std::memset(ps.hHops, 0, S*sizeof(HopDesc));
	std::memset(ps.hEmit, 0, S*sizeof(EmitDesc));
	std::memset(ps.hTileInfo, 0, (size_t)nSub*2*subS*sizeof(int));
	std::fill(pendTileHas.begin(), pendTileHas.end(), 0);
	std::fill(pendMaxSpan.begin(), pendMaxSpan.end(), 0);
	bool anyFrozen = false, anyZeroPrev = false;
	for (int s : pendList) {
		const PendingBlock &pb = pend[s];
		HopDesc &hd = ps.hHops[s];
		hd.flags = pb.flags | HOP_PREANALYSED;
		hd.timeFactor = pb.timeFactor;
		hd.seed = pb.seed;
		hd.startBin = pb.startBin;
		hd.outPos = -int(sched[s].samplesSinceLast);
		const bool nw = (pb.flags & HOP_NEW_SPECTRUM) != 0;
		hd.inSrc = nw ? 0 : SRC_STATE;
		hd.prevSrc = (nw && (pb.flags & HOP_REANALYSE_PREV)) ? SRC_REANALYSED : SRC_STATE;
		EmitDesc &ed = ps.hEmit[s];
		ed.firstHopPos = hd.outPos;
		ed.hopCount = 1;
		const int sub = s/subS, sl = s%subS;
		int *info = ps.hTileInfo + (size_t)sub*2*subS;
		info[sl] = 1;
		info[subS + sl] = nw ? 0 : -1;
		unsigned char *th = pendTileHas.data() + (size_t)sub*kTileHasStride;
		th[0] = 1;
		if (pb.flags & HOP_MAPPED) th[1] = 1;
		if (pb.flags & HOP_FORMANTS) { th[2] = 1; th[10] = 1; } // (a block in flight may carry latched parameters: the three-kernel form decides per stream)
		if (pb.flags & HOP_RANDOM_TF) th[4] = 1;
		if (pb.startBin > 0) th[7] = 1;
		th[8] = 1;
		anyFrozen = anyFrozen || pb.frozenPeaks || pb.frozenForm0 || pb.frozenForm2;
		anyZeroPrev = anyZeroPrev || pb.zeroPrevAfter;
		LastHop &lh = lastHop[s];
		lh.slot = sub & 1;
		lh.local = 0;
		lh.subLocal = sl;
		lh.mapped = (pb.flags & HOP_MAPPED) != 0;
		lh.formants = (pb.flags & HOP_FORMANTS) != 0;
	}
	SMST_HIP(hipMemcpyAsync(ps.hops, ps.hHops, S*sizeof(HopDesc), hipMemcpyHostToDevice, st));
	SMST_HIP(hipMemcpyAsync(ps.emit, ps.hEmit, S*sizeof(EmitDesc), hipMemcpyHostToDevice, st));
	SMST_HIP(hipMemcpyAsync(ps.tileInfo, ps.hTileInfo, (size_t)nSub*2*subS*sizeof(int), hipMemcpyHostToDevice, st));
	if (anyFrozen) { // some step of some block ran before a setter: that step keeps the values it saw
		for (int s = 0; s < S; ++s) ps.hPrm[0][s] = ps.hPrm[1][s] = ps.hPrm[2][s] = params[s];
		for (int s : pendList) {
			const PendingBlock &pb = pend[s];
			if (pb.frozenPeaks) ps.hPrm[0][s] = pb.peaks;
			if (pb.frozenForm0) ps.hPrm[1][s] = pb.form0;
			if (pb.frozenForm2) ps.hPrm[2][s] = pb.form2;
		}
		for (int j = 0; j < 3; ++j) SMST_HIP(hipMemcpyAsync(ps.prm[j], ps.hPrm[j], S*sizeof(StreamParams), hipMemcpyHostToDevice, st));
		d.paramsPeaks = ps.prm[0]; d.paramsForm0 = ps.prm[1]; d.paramsForm2 = ps.prm[2];
	}
	if (synthChannels) {
		for (int s = 0; s < S; ++s) ps.hSynthChannels[s] = -1;
		for (int s : pendList) ps.hSynthChannels[s] = synthChannels[s];
		SMST_HIP(hipMemcpyAsync(ps.synthChannels, ps.hSynthChannels, S*sizeof(int), hipMemcpyHostToDevice, st));
	}
	d.hops = ps.hops;
	d.emit = ps.emit;
	d.hopStride = 1;
	d.emitStride = 1;
	const IoArgs io{nullptr, nullptr, 0, 0, 0, 0, dZeroCounts, dZeroCounts};
	runTiles(TileRun{&io, 1, 1, pendTileHas.data(), pendMaxSpan.data(), ps.tileInfo, true, synthChannels ? ps.synthChannels : nullptr, false});
	d.paramsPeaks = d.paramsForm0 = d.paramsForm2 = dParams;
	if (anyZeroPrev) {
		for (int s = 0; s < S; ++s) ps.hBits[s] = 0;
		for (int s : pendList) if (pend[s].zeroPrevAfter) ps.hBits[s] = 4;
		SMST_HIP(hipMemcpyAsync(ps.bits, ps.hBits, S*sizeof(int), hipMemcpyHostToDevice, st));
		launchResetStreams(d, ps.bits, 0, dSeedWp, st);
	}
	SMST_HIP(hipEventRecord(ps.done, st));
	ps.used = true;
	for (int s : pendList) {
		if (pend[s].flags & HOP_RANDOM_TF) sched[s].seed = unsigned((unsigned long long)sched[s].seed*lcgHopJump % 2147483647ull); // its draws happen now
		pend[s] = PendingBlock();
	}
	pendList.clear();
}

// The tiles of a call (or of a run of blocks in flight): analysis, feed-forward passes, recurrence, synthesis and emission, pipelined
// over three HIP streams and two workspaces.
void Batch::runTiles(const TileRun &run) {
	const IoArgs &io = *run.io;
	const int nTiles = run.nTiles;
	if (run.carriedOnly) { // no stream fires a hop: the front of the carry is the output, and what is left stays where it is
		timed(timings.emitMs, [&] { launchEmitCarried(d, io, st); if (profiling) ++timings.emitLaunches; });
		checkLaunch("emission of the carried sums");
		return;
	}
	std::fill(carryBase.begin(), carryBase.end(), 0); // every tile writes every stream's carry from the front of its rows
	const int carryFirst = d.carryCur;
	// Runs of two or more tiles that the continuous wavefront can take (kVocoderCont) go to it, the tiles around them -- the first tile
	// after a reset, whose first hop draws random time factors; a tile with a pitch map -- tile by tile; the segments follow each other on `st`.
	int t = 0;
	while (t < nTiles) {
		int e = t;
		while (e < nTiles && continuousApplies(run, e)) ++e;
		if (e - t >= 2) { runTilesContinuous(run, t, e, carryFirst); t = e; continue; }
		e = std::max(e, t + 1);
		for (;;) { // extend the tile-by-tile segment up to the next run of two
			int r = e;
			while (r < nTiles && continuousApplies(run, r)) ++r;
			if (e >= nTiles || r - e >= 2) break;
			e = std::max(r, e + 1);
		}
		runTilesRange(run, t, e, carryFirst);
		t = e;
	}
	d.carryCur = (carryFirst + nTiles) & 1;
}

void Batch::runTilesRange(const TileRun &run, int tile0, int tile1, int carryFirst) {
	const IoArgs &io = *run.io;
	const int T = d.T, nTiles = run.nTiles, maxHops = run.maxHops;
	const int nSub = (S + subS - 1)/subS;
	// Three HIP streams: `st` runs the feed-forward kernels of tile q, `stChain` the recurrence of tile q (a few
	// hundred waves, instruction-issue bound), `stSynth` synthesis + emission.  Two workspaces alternate, so the bulk
	// kernels of the next tile fill the machine while the recurrence of the current one is in flight.
	const bool serial = profiling || !overlap || run.pendingRun;
	const bool singleHop = singleHopSupported(d) && !noSingleHop;
	hipStream_t sF = st, sC = serial ? st : stChain, sS = serial ? st : stSynth;
	if (!serial) {
		SMST_HIP(hipEventRecord(evStart, st));
		SMST_HIP(hipStreamWaitEvent(sC, evStart, 0));
		SMST_HIP(hipStreamWaitEvent(sS, evStart, 0));
	}
	int q = tile0; // (the workspace of tile t of a single sub-batch is slots[t & 1] whatever segment it runs in: debugGetMap relies on it)
	for (int sub = 0; sub < nSub; ++sub) {
		const int sBase = sub*subS;
		const int ns = std::min(subS, S - sBase);
		for (int t = tile0; t < tile1; ++t, ++q) {
			const unsigned char *th = run.tileHas + (size_t)(sub*nTiles + t)*kTileHasStride;
			const int hopBase = t*T;
			const int tileHops = std::min(T, std::max(1, maxHops - hopBase));
			const int slot = q & 1;
			const bool plain = !(th[1] || th[2]);
			const bool fused = fusedSupported(d) && !noFuse; // mono/stereo: records stay in LDS (kVocoder)
			const TileBuffers &w = slots[slot];
			DevBatch dd = d;
			dd.Xcur = w.Xcur; dd.Xprev = w.Xprev; dd.PE = w.PE; dd.OUT = w.OUT; dd.REC = w.REC; dd.dump = w.dump;
			dd.map = w.map; dd.ratio = w.ratio; dd.envelope = w.envelope; dd.energyT = w.energyT; dd.smoothT = w.smoothT; dd.peaksT = w.peaksT; dd.est = w.est; dd.freqEst = w.freqEst; dd.frames = w.frames; dd.fftScratch = w.fftScratch;
			dd.carryCur = (carryFirst + t) & 1;
			dd.nHops = run.dTileInfo + ((size_t)(sub*nTiles + t)*2)*subS;
			dd.lastNewHop = dd.nHops + subS;
			if (!serial && q - tile0 >= 2) { // this workspace was last used by tile q-2
				SMST_HIP(hipStreamWaitEvent(sF, evChain[slot], 0));
				SMST_HIP(hipStreamWaitEvent(sF, evSynth[slot], 0));
			}
			if (!serial && fused && !plain && q - tile0 >= 1) SMST_HIP(hipStreamWaitEvent(sF, evChain[slot ^ 1], 0)); // pass A reads the carried state
			if (th[0]) {
				if (th[8]) timed(timings.otherMs, [&] { launchPendingToTile(dd, sBase, ns, dPendIn, dPendPrev, sF); }); // blocks that began in an earlier call: their spectra are waiting
				if (th[3]) timed(timings.analyseMs, [&] { launchAnalyse(dd, io, sBase, ns, hopBase, tileHops, th[5] != 0, th[6] != 0, sF); if (profiling) ++timings.analyseLaunches; });
				bool passADone = false;
				if (th[1] || th[2]) timed(timings.feedMs, [&] { passADone = launchFeed(dd, sBase, ns, hopBase, tileHops, th[2] != 0, th[10] != 0, sF); });
				timed(timings.predictMs, [&] {
					if (fused) launchPredictFused(dd, sBase, ns, hopBase, tileHops, plain, passADone, sF);
					else launchPredict(dd, sBase, ns, hopBase, tileHops, plain, passADone, sF);
					if (profiling) ++timings.predictLaunches;
				});
				// the carried feed-forward state (Band.input/.prevInput, Prediction.energy) may only move on once every
				// reader of the OLD state has run: in the fused path the producers inside kVocoder still read it
				if (!fused) timed(timings.otherMs, [&] { launchCarryFeed(dd, sBase, ns, hopBase, th[2] != 0, sF); });
			}
			checkLaunch("analysis / feed-forward kernels");
			if (!serial) {
				SMST_HIP(hipEventRecord(evFeed[slot], sF));
				SMST_HIP(hipStreamWaitEvent(sC, evFeed[slot], 0));
			}
			// synthesis + overlap-add + emission in one kernel where the geometry and the batch allow it
			const bool emitted = th[0] && !th[8] && synthEmitApplies(dd, ns, tileHops); // (th[8]: a hop that began before the call's first sample -- kSynthTeams + kEmit place its frame)
			const bool earlySynth = emitted || tileHops == 1;
			if (th[0]) {
				hipEvent_t liveA = nullptr, liveB = nullptr;
				if (liveTiming && !serial) {
					if (liveEvents.size() == livePool.size()) growLivePool(livePool.size() + 64); // only if a run outgrows what enableProfiling() made
					liveA = livePool[liveEvents.size()].first;
					liveB = livePool[liveEvents.size()].second;
					SMST_HIP(hipEventRecord(liveA, sC));
				}
				timed(timings.chainMs, [&] {
					// (th[7]: a block that a flush() interrupted inside its main prediction, HopDesc.startBin > 0.  Every form that forms its
					// records through computeRecord honours it (all-zero records below the start bin); the staged / line-aligned producers of
					// kVocoder do not go through it, so such a tile takes the gathering form -- `bounded` false)
					if (fused && tileHops == 1 && singleHop && !noAcross && acrossSupported(dd) && !th[7]) launchVocoderAcross(dd, sBase, ns, hopBase, plain, sC);
					else if (fused && tileHops == 1 && singleHop) launchVocoderOne(dd, sBase, ns, hopBase, plain, sC);
					else if (fused) launchVocoder(dd, sBase, ns, hopBase, plain, !th[4] && !th[7], sC);
					else launchChain(dd, sBase, ns, hopBase, sC);
					if (profiling) ++timings.chainLaunches;
				});
				if (liveA) {
					SMST_HIP(hipEventRecord(liveB, sC));
					liveEvents.emplace_back(liveA, liveB);
				}
				// synthesis needs the recurrence's rows only: the hand-over of the tile's last rows to the carried state (two copies at HBM rate --
				// 250 us of a 4096-stream hop quantum) runs beside it, behind the event synthesis waits for.  Where synthesis is a grid of
				// per-frame workgroups and more recurrence launches follow (config 5) the hand-over goes first, as before: it is what the NEXT
				// recurrence waits for, and behind a machine full of synthesis workgroups it would start late (config 5: 99.6 -> 101 ms)
				timed(timings.otherMs, [&] {
					if (run.dSynthChannels || !earlySynth) { // (the carried output state takes the rows as the recurrence left them)
						launchCarryOut(dd, sBase, ns, sC);
						if (run.dSynthChannels) launchMaskOutRows(dd, sBase, ns, run.dSynthChannels, sC);
					}
					if (!earlySynth && fused) launchCarryFeed(dd, sBase, ns, hopBase, th[2] != 0, sC);
					if (!serial) SMST_HIP(hipEventRecord(evOut[slot], sC));
					if (earlySynth && fused) launchCarryFeed(dd, sBase, ns, hopBase, th[2] != 0, sC);
					if (earlySynth && !run.dSynthChannels) launchCarryOut(dd, sBase, ns, sC);
				});
			}
			checkLaunch("bin recurrence");
			// kSynthEmitTeams' window products depend on nothing the recurrence writes: queued in front of the wait for it
			if (emitted) timed(timings.otherMs, [&] { launchEmitProducts(dd, sBase, ns, t, sS); });
			if (!serial) {
				SMST_HIP(hipEventRecord(evChain[slot], sC));
				SMST_HIP(hipStreamWaitEvent(sS, th[0] ? evOut[slot] : evChain[slot], 0));
			}
			if (th[0]) timed(timings.synthMs, [&] {
				if (emitted) launchSynthEmit(dd, io, sBase, ns, t, sS);
				else launchSynth(dd, sBase, ns, hopBase, tileHops, sS);
				if (profiling) ++timings.synthLaunches;
			});
			if (!emitted) timed(timings.emitMs, [&] { launchEmit(dd, io, sBase, ns, t, run.maxSpan[(size_t)sub*nTiles + t], sS); if (profiling) ++timings.emitLaunches; });
			checkLaunch("synthesis / emission");
			if (!serial) SMST_HIP(hipEventRecord(evSynth[slot], sS));
		}
	}
	if (!serial) { // everything the caller can observe is ordered on `st` again
		for (int i = 0; i < 2 && i < q - tile0; ++i) {
			SMST_HIP(hipStreamWaitEvent(st, evChain[i], 0));
			SMST_HIP(hipStreamWaitEvent(st, evSynth[i], 0));
		}
	}
}

// Whether a tile can be part of a continuous wavefront (kVocoderCont; runs of two or more such tiles are): the geometries of the
// line-aligned producers, one sub-batch, the tile plain, bounded, whole-line -- what launchVocoder sends to the aligned form -- with a
// new spectrum in every hop (the rows of a later tile then never read the carried feed-forward state).
bool Batch::continuousApplies(const TileRun &run, int t) const {
	if (!slots[2].Xcur || run.pendingRun || run.carriedOnly || subS != S) return false;
	if (!(d.L == 4 || d.alignAll)) return false; // (as launchVocoder chooses the aligned producers)
	const unsigned char *th = run.tileHas + (size_t)t*kTileHasStride;
	return th[0] && !th[1] && !th[2] && !th[4] && !th[7] && !th[8] && !th[9];
}

// The same pipeline as runTiles -- analysis of tile t+1, recurrence, synthesis + emission over three HIP streams -- with the recurrence as
// launch t of the continuous wavefront: it begins tile t and finishes tile t-1, so synthesis runs one tile later and the workspaces rotate
// over three slots (launch t reads the spectra of tiles t-1 and t while tile t+1 is analysed).  The carried state is read by launch 0 only
// (the call's first hop); it is written once per stream, from its last tile, behind the launch that finishes that tile.
void Batch::runTilesContinuous(const TileRun &run, int tile0, int tile1, int carryFirst) {
	const IoArgs &io = *run.io;
	const int T = d.T, maxHops = run.maxHops;
	const bool serial = profiling || !overlap;
	hipStream_t sF = st, sC = serial ? st : stChain, sS = serial ? st : stSynth;
	if (!serial) {
		SMST_HIP(hipEventRecord(evStart, st));
		SMST_HIP(hipStreamWaitEvent(sC, evStart, 0));
		SMST_HIP(hipStreamWaitEvent(sS, evStart, 0));
	}
	const int P = continuousPeriod(d);
	auto tileView = [&](int t) {
		const TileBuffers &w = slots[t%3];
		DevBatch dd = d;
		dd.Xcur = w.Xcur; dd.Xprev = w.Xprev; dd.OUT = w.OUT; dd.frames = w.frames; dd.fftScratch = w.fftScratch;
		dd.PE = nullptr; dd.REC = nullptr; dd.map = nullptr; dd.ratio = nullptr; // (plain tiles: never touched)
		dd.carryCur = (carryFirst + t) & 1;
		dd.nHops = run.dTileInfo + ((size_t)t*2)*subS;
		dd.lastNewHop = dd.nHops + subS;
		return dd;
	};
	auto tileHops = [&](int t) { return std::min(T, std::max(1, maxHops - t*T)); };
	auto lastOfSomeStream = [&](int t) {
		if (t == tile1 - 1) return true; // (the tiles behind the segment, if any, start from the carried state)
		for (int s = 0; s < S; ++s) if (hopCount[s] > 0 && (hopCount[s] - 1)/T == t) return true;
		return false;
	};
	// tile t is complete (its last rows ran in the launch just enqueued on sC): hand-over to the carried state, synthesis + emission
	auto finishTile = [&](int t, int launch) {
		const DevBatch dd = tileView(t);
		const bool emitted = synthEmitApplies(dd, S, tileHops(t));
		if (lastOfSomeStream(t)) // (only a stream's last tile goes to the carried state)
		 timed(timings.otherMs, [&] {
			launchCarryFeed(dd, 0, S, t*T, false, sC);
			launchCarryOut(dd, 0, S, sC);
		});
		if (!serial) SMST_HIP(hipEventRecord(evChain[t%3], sC)); // (everything that reads tile t's workspace on sC is in front of this)
		if (emitted) timed(timings.otherMs, [&] { launchEmitProducts(dd, 0, S, t, sS); });
		if (!serial) SMST_HIP(hipStreamWaitEvent(sS, evOut[launch%3], 0));
		timed(timings.synthMs, [&] {
			if (emitted) launchSynthEmit(dd, io, 0, S, t, sS);
			else launchSynth(dd, 0, S, t*T, tileHops(t), sS);
			if (profiling) ++timings.synthLaunches;
		});
		if (!emitted) timed(timings.emitMs, [&] { launchEmit(dd, io, 0, S, t, run.maxSpan[t], sS); if (profiling) ++timings.emitLaunches; });
		checkLaunch("synthesis / emission");
		if (!serial) SMST_HIP(hipEventRecord(evSynth[t%3], sS));
	};
	for (int t = tile0; t < tile1; ++t) {
		const unsigned char *th = run.tileHas + (size_t)t*kTileHasStride;
		const DevBatch dd = tileView(t);
		if (!serial && t >= tile0 + 3) { // this workspace held tile t-3: its spectra were last read by launch t-2, its results by the synthesis of tile t-3
			SMST_HIP(hipStreamWaitEvent(sF, evChain[t%3], 0));
			SMST_HIP(hipStreamWaitEvent(sF, evSynth[t%3], 0));
		}
		timed(timings.analyseMs, [&] { launchAnalyse(dd, io, 0, S, t*T, tileHops(t), th[5] != 0, th[6] != 0, sF); if (profiling) ++timings.analyseLaunches; });
		checkLaunch("analysis");
		if (!serial) {
			SMST_HIP(hipEventRecord(evFeed[t%3], sF));
			SMST_HIP(hipStreamWaitEvent(sC, evFeed[t%3], 0));
		}
		ContArgs a{};
		const TileBuffers &w0 = slots[(t + 2)%3], &w1 = slots[t%3]; // tile t-1, tile t
		a.Xcur[0] = w0.Xcur; a.Xprev[0] = w0.Xprev; a.OUT[0] = w0.OUT;
		a.Xcur[1] = w1.Xcur; a.Xprev[1] = w1.Xprev; a.OUT[1] = w1.OUT;
		a.tileInfo = run.dTileInfo + ((size_t)tile0*2)*subS; // the kernel counts tiles, hops and blocks from the segment's first tile
		a.tileStride = 2*subS;
		a.nTiles = tile1 - tile0;
		a.tile = t - tile0;
		a.hopBase = tile0*T;
		a.period = P;
		a.n0 = P*a.tile;
		a.n1 = (t == tile1 - 1) ? P*(a.tile + 1) + 62 : P*(a.tile + 1); // the last launch runs until row 63 of the last tile is through
		a.save = dContSave;
		a.writerWave = contWriterWave;
		hipEvent_t liveA = nullptr, liveB = nullptr;
		if (liveTiming && !serial) {
			if (liveEvents.size() == livePool.size()) growLivePool(livePool.size() + 64);
			liveA = livePool[liveEvents.size()].first;
			liveB = livePool[liveEvents.size()].second;
			SMST_HIP(hipEventRecord(liveA, sC));
		}
		timed(timings.chainMs, [&] { launchVocoderContinuous(d, a, 0, S, sC); if (profiling) ++timings.chainLaunches; });
		if (liveA) {
			SMST_HIP(hipEventRecord(liveB, sC));
			liveEvents.emplace_back(liveA, liveB);
		}
		checkLaunch("bin recurrence (continuous)");
		if (!serial) SMST_HIP(hipEventRecord(evOut[t%3], sC));
		if (t > tile0) finishTile(t - 1, t);
	}
	finishTile(tile1 - 1, tile1 - 1);
	if (!serial) {
		for (int i = 0; i < 3 && i < tile1 - tile0; ++i) { // everything the caller can observe is ordered on `st` again
			SMST_HIP(hipStreamWaitEvent(st, evChain[i], 0));
			SMST_HIP(hipStreamWaitEvent(st, evSynth[i], 0));
		}
	}
}

void Batch::process(const float *in, long long inSS, long long inCS, const int *inSamples,
                    float *out, long long outSS, long long outCS, const int *outSamples, const unsigned char *active) {
	SMST_HIP(hipSetDevice(dev));
	const auto hostT0 = std::chrono::steady_clock::now();
	auto msSince = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
	uploadParams();
	// This call's tables go into the set that the call before the previous one used (its kernels must have finished);
	// everything up to the first tile launch runs on `stGate`, so the host-side scheduling of this call overlaps the
	// kernels of the previous call that are still queued on `st`.  All staging is pinned and owned by the set, so the
	// uploads are asynchronous and the ONLY host synchronisation of a call is the silence-gate readback.
	callCur ^= 1;
	CallSet &cs = callSets[callCur];
	if (cs.used) { const auto t = std::chrono::steady_clock::now(); SMST_HIP(hipEventSynchronize(cs.done)); hostTimes.waitTablesMs += msSince(t); }
	dInSamples = cs.inSamples; dOutSamples = cs.outSamples; dFlags = cs.flags;
	const int T = d.T;
	int *nIn = cs.hInSamples, *nOut = cs.hOutSamples;
	int maxOut = 0, maxIn = 0;
	for (int s = 0; s < S; ++s) {
		bool on = !active || active[s];
		nIn[s] = on ? inSamples[s] : 0;
		nOut[s] = on ? outSamples[s] : 0;
		if (nIn[s] < 0 || nOut[s] < 0) throw Error("negative sample count");
		maxOut = std::max(maxOut, nOut[s]);
		maxIn = std::max(maxIn, nIn[s]);
	}
	SMST_HIP(hipMemcpyAsync(dInSamples, nIn, S*sizeof(int), hipMemcpyHostToDevice, stGate));
	SMST_HIP(hipMemcpyAsync(dOutSamples, nOut, S*sizeof(int), hipMemcpyHostToDevice, stGate));
	IoArgs io{in, out, inSS, inCS, outSS, outCS, dInSamples, dOutSamples};

	// K5: silence gate needs the input energy on the host (one 64-byte-per-stream readback per call)
	launchEnergy(d, io, 0, S, maxIn, dEnergy, stGate);
	SMST_HIP(hipMemcpyAsync(cs.hEnergy, dEnergy, (size_t)S*kEnergyParts*sizeof(float), hipMemcpyDeviceToHost, stGate));
	{ const auto t = std::chrono::steady_clock::now(); SMST_HIP(hipStreamSynchronize(stGate)); hostTimes.waitGateMs += msSince(t); }

	// K0, pass 1: silence gate (signalsmith-stretch.h:231-278) and the number of hops each stream fires in this call
	int *passFlags = cs.hFlags;
	int *clearBits = cs.hResetBits; // per CALL (pinned + its own device copy): a shared buffer could be overwritten by the next call's upload before this call's kResetStreams has run
	bool anyPass = false, anyClear = false;
	int maxHops = 0;
	pendList.clear();
	for (int s = 0; s < S; ++s) {
		passFlags[s] = 0;
		hopFirst[s] = 0;
		hopCount[s] = 0;
		leavesPendingV[s] = 0;
		ridesV[s] = 0;
		clearBits[s] = 0;
		if (active && !active[s]) continue;
		lastHop[s].slot = -1; // smst_batch_debug_get_map reports the newest hop of THIS call only
		StreamSched &sc = sched[s];
		float e = 0;
		for (int p = 0; p < kEnergyParts; ++p) e += cs.hEnergy[(size_t)s*kEnergyParts + p];
		if (e < kNoiseFloor) { // :240-278
			if (sc.silenceCounter >= size_t(2*B)) {
				if (sc.silenceFirst) {
					sc.silenceFirst = false;
					sc.seed = seedAfterDroppedBlock(s); // ... which drops the block in flight, with the draws it has made
					sc.samplesSinceLast = SIZE_MAX; // blockProcess = {}
					pend[s] = PendingBlock();
					clearBits[s] = 2 | 4 | 8;       // Band.input / .prevInput / .output := 0
					anyClear = true;
				}
				passFlags[s] = 1;
				anyPass = true;
				continue; // history is still updated below (copyInput, :270)
			} else {
				sc.silenceCounter += size_t(nIn[s]);
			}
		} else {
			sc.silenceCounter = 0;
			sc.silenceFirst = true;
		}
		const int first = (sc.samplesSinceLast >= size_t(I)) ? 0 : int(size_t(I) - sc.samplesSinceLast);
		hopFirst[s] = first;
		const int starts = (nOut[s] > first) ? (nOut[s] - first + I - 1)/I : 0; // blocks that begin in this call (:281)
		lastStarts[s] = starts;
		if (split) {
			// split computation: a block is finished when its interval is (:321-325).  The block in flight from earlier calls runs now if
			// this call reaches the end of its interval; the last block that begins here stays in flight unless its interval ends here too
			// (a plain one becomes the stream's first hop of this call; one that a flush() interrupted or whose steps saw different parameters runs by itself first)
			ridesV[s] = 0;
			if (pend[s].valid && nOut[s] >= first) {
				const PendingBlock &pb = pend[s];
				if (pb.startBin > 0 || pb.zeroPrevAfter || pb.frozenPeaks || pb.frozenForm0 || pb.frozenForm2) pendList.push_back(s);
				else ridesV[s] = 1;
			}
			const int complete = (nOut[s] > first) ? (nOut[s] - first)/I : 0;
			leavesPendingV[s] = starts > complete;
			hopCount[s] = complete + ridesV[s];
		} else {
			hopCount[s] = starts;
		}
		maxHops = std::max(maxHops, hopCount[s]);
	}
	if (anyClear) { // asynchronous: the masks travel on `st`, ahead of the launch that reads them
		SMST_HIP(hipMemcpyAsync(cs.resetBits, clearBits, S*sizeof(int), hipMemcpyHostToDevice, st));
		launchResetStreams(d, cs.resetBits, 0, dSeedWp, st);
	}
	if (!pendList.empty()) runPendingBlocks(nullptr);
	const int nTiles = std::max(1, (maxHops + T - 1)/T);
	// One row of T hop descriptors per stream and tile -- except in the real-time calling pattern, where no stream fires more than one hop per
	// call: the single-hop kernels index hop 0 only, so a row is ONE descriptor (4096 streams: 131 KB to clear and upload per 128-frame
	// quantum instead of 8.4 MB -- a third of the quantum's cost).  (The wavefront kernels load a row of 64 blindly: they keep the full rows.)
	const bool compactHops = maxHops <= 1 && singleHopSupported(d) && !noSingleHop && fusedSupported(d) && !noFuse;
	const int hopStride = compactHops ? 1 : nTiles*T;
	const int nSub = (S + subS - 1)/subS;

	// per-call tables: pinned staging and device copies grow only when a call needs more hops than any earlier call -- and
	// then BOTH sets grow, so that the call after this one (which uses the other set) does not allocate either.  The other
	// set's old tables may still be read by the previous call's kernels: they are retired and freed when that set is next used.
	for (void *q : cs.retiredDevice) devFree(q);
	for (void *q : cs.retiredPinned) pinnedFree(q);
	cs.retiredDevice.clear();
	cs.retiredPinned.clear();
	const size_t needHops = (size_t)S*hopStride, needEmit = (size_t)S*nTiles, needInfo = (size_t)nSub*nTiles*2*subS;
	for (int which = 0; which < 2; ++which) {
		CallSet &t = callSets[which ? callCur ^ 1 : callCur];
		const bool mine = which == 0;
		auto retire = [&](void *dev, void *host) {
			if (mine) { devFree(dev); pinnedFree(host); }
			else { if (dev) t.retiredDevice.push_back(dev); if (host) t.retiredPinned.push_back(host); }
		};
		if (needHops > t.hopsCap) {
			retire(t.hops, t.hHops);
			t.hopsCap = needHops + needHops/4;
			t.hops = devAlloc<HopDesc>(t.hopsCap);
			t.hHops = pinnedAlloc<HopDesc>(t.hopsCap);
		}
		if (needEmit > t.emitCap) {
			retire(t.emit, t.hEmit);
			t.emitCap = needEmit + needEmit/4;
			t.emit = devAlloc<EmitDesc>(t.emitCap);
			t.hEmit = pinnedAlloc<EmitDesc>(t.emitCap);
		}
		if (needInfo > t.tileInfoCap) {
			retire(t.tileInfo, t.hTileInfo);
			t.tileInfoCap = needInfo + needInfo/4;
			t.tileInfo = devAlloc<int>(t.tileInfoCap);
			t.hTileInfo = pinnedAlloc<int>(t.tileInfoCap);
		}
	}
	dHops = cs.hops; dEmit = cs.emit; dTileInfo = cs.tileInfo;
	HopDesc *hopsAll = cs.hHops;
	EmitDesc *emitAll = cs.hEmit;
	int *tileInfo = cs.hTileInfo; // layout: [sub][tile][2][subS] (nHops, lastNewHop)
	std::memset(hopsAll, 0, needHops*sizeof(HopDesc));
	std::memset(tileInfo, 0, needInfo*sizeof(int));
	ensureSize(maxSpanV, (size_t)nSub*nTiles, allocEvents);
	ensureSize(tileHasV, (size_t)nSub*nTiles*kTileHasStride, allocEvents); // (smst_types.h: kTileHasStride)
	std::fill(maxSpanV.begin(), maxSpanV.end(), 0);
	std::fill(tileHasV.begin(), tileHasV.end(), 0);

	// K0, pass 2: block scheduler, exactly as signalsmith-stretch.h:280-319 does it per stream, straight into the tables
	bool anyPendAnalysis = false, pendInCall = false, pendLate = false;
	for (int s = 0; s < S; ++s) {
		const int sub = s/subS, sl = s%subS;
		HopDesc *list = hopsAll + (size_t)s*hopStride;
		const int nh = hopCount[s];
		const int rides = ridesV[s]; // split computation: the block in flight from an earlier call is this call's hop 0 ...
		const int nStart = nh + (leavesPendingV[s] ? 1 : 0); // ... and the last block that begins here may stay in flight
		if (split) cs.hPendHops[s] = HopDesc{};
		int lastNew = -1;
		if (rides) {
			const PendingBlock &pb = pend[s];
			HopDesc &hd = list[0];
			hd.flags = pb.flags | HOP_PREANALYSED;
			hd.timeFactor = pb.timeFactor;
			hd.seed = pb.seed;
			hd.outPos = -int(sched[s].samplesSinceLast); // it began that many samples ago: its frame lands where its interval ends (:292-296)
			const bool nw = (pb.flags & HOP_NEW_SPECTRUM) != 0;
			hd.inSrc = nw ? 0 : SRC_STATE;
			hd.prevSrc = (nw && (pb.flags & HOP_REANALYSE_PREV)) ? SRC_REANALYSED : SRC_STATE;
			if (nw) lastNew = 0;
			if (pb.flags & HOP_RANDOM_TF) sched[s].seed = unsigned((unsigned long long)sched[s].seed*lcgHopJump % 2147483647ull); // its draws happen now
			pend[s] = PendingBlock();
		}
		if (nStart > rides) {
			StreamSched &sc = sched[s];
			const StreamParams &prm = params[s];
			const bool mapped = prm.hasCustomMap || prm.freqMultiplier != 1; // :300
			const bool formants = prm.formantMultiplier != 1 || (prm.formantCompensation && mapped); // :310
			int o = hopFirst[s];
			for (int j = rides; j < nStart; ++j, o += I) {
				HopDesc inFlight{};
				HopDesc &hd = (j < nh) ? list[j] : inFlight;
				int inputOffset = int(std::round(o*float(nIn[s])/nOut[s])); // :288 (fp32 on purpose)
				int inputInterval = inputOffset - sc.prevInputOffset;
				sc.prevInputOffset = inputOffset;
				hd.inputOffset = inputOffset;
				hd.outPos = o;
				unsigned flags = HOP_ACTIVE;
				const bool newSpectrum = sc.didSeek || inputInterval > 0; // :299
				bool reanalyse = false;
				if (newSpectrum) {
					flags |= HOP_NEW_SPECTRUM;
					reanalyse = sc.didSeek || std::abs(inputInterval - I) > 1; // :303
					if (reanalyse) flags |= HOP_REANALYSE_PREV;
				}
				if (mapped) flags |= HOP_MAPPED;
				if (formants) flags |= HOP_FORMANTS;
				float tf = sc.didSeek ? sc.seekTimeFactor : float(I)/std::max<float>(1, float(inputInterval)); // :312
				sc.didSeek = false;
				tf = std::max<float>(tf, 1/kMaxCleanStretch); // :638
				if (tf > kMaxCleanStretch) flags |= HOP_RANDOM_TF; // :639
				hd.timeFactor = tf;
				hd.flags = flags;
				lastSteps[s] = stepLayout(flags).steps;
				hd.seed = sc.seed; // the engine's state before this hop's draws
				if ((flags & HOP_RANDOM_TF) && j < nh) sc.seed = unsigned((unsigned long long)sc.seed*lcgHopJump % 2147483647ull); // 2M - 2 draws later (a block left in flight draws when it runs: seedAfterDroppedBlock)
				if (j == nh) {
					// in flight at the end of the call: analysed now, from the input as it stands (:293 stashes it), into the pending buffers;
					// everything else when its interval is complete -- or when a flush() needs to know how far it has come
					PendingBlock &pb = pend[s];
					pb = PendingBlock();
					pb.valid = true;
					pb.flags = flags;
					pb.timeFactor = tf;
					pb.seed = hd.seed;
					HopDesc &ph = cs.hPendHops[s];
					ph.inputOffset = inputOffset;
					ph.flags = flags & (HOP_ACTIVE | HOP_NEW_SPECTRUM | HOP_REANALYSE_PREV);
					if (newSpectrum) {
						anyPendAnalysis = true;
						for (int which = 0; which < (reanalyse ? 2 : 1); ++which) (analysisWindowInCall(d.B, d.M, d.I, inputOffset, which, nIn[s]) ? pendInCall : pendLate) = true;
					}
					continue;
				}
				const int tile = j/T;
				const bool lastInTile = lastNew >= 0 && lastNew/T == tile;
				hd.inSrc = newSpectrum ? j%T : (lastInTile ? lastNew%T : SRC_STATE);
				if (newSpectrum && reanalyse) hd.prevSrc = SRC_REANALYSED;
				else hd.prevSrc = lastInTile ? lastNew%T : SRC_STATE;
				if (newSpectrum) lastNew = j;
			}
			sc.samplesSinceLast = size_t(nOut[s] - (o - I));
		} else if (!passFlags[s] && !(active && !active[s])) {
			StreamSched &sc = sched[s];
			if (sc.samplesSinceLast != SIZE_MAX) sc.samplesSinceLast += size_t(nOut[s]);
		}
		if (!(active && !active[s]) && !passFlags[s]) sched[s].prevInputOffset -= nIn[s]; // :419
		for (int t = 0; t < nTiles; ++t) {
			const int h0 = t*T, h1 = std::min(nh, h0 + T);
			const int cnt = std::max(0, h1 - h0);
			EmitDesc ed{};
			const int total = passFlags[s] ? 0 : nOut[s];
			if (t == 0) ed.nLo = 0; else ed.nLo = (h0 < nh) ? list[h0].outPos : total;
			ed.nHi = (h1 < nh && cnt > 0) ? list[h1].outPos : total;
			if (cnt == 0 && t > 0) ed.nLo = ed.nHi = total;
			ed.firstHopPos = cnt > 0 ? list[h0].outPos : 0;
			ed.hopCount = cnt;
			emitAll[(size_t)s*nTiles + t] = ed;
			int *info = tileInfo + ((size_t)(sub*nTiles + t)*2)*subS;
			info[sl] = cnt;
			int lastNewLocal = -1;
			unsigned char *th = tileHasV.data() + (size_t)(sub*nTiles + t)*kTileHasStride;
			for (int h = h0; h < h1; ++h) {
				const unsigned f = list[h].flags;
				if (f & HOP_PREANALYSED) th[8] = 1;
				if ((f & HOP_NEW_SPECTRUM) && (f & HOP_PREANALYSED)) lastNewLocal = h - h0;
				if ((f & HOP_NEW_SPECTRUM) && !(f & HOP_PREANALYSED)) {
					lastNewLocal = h - h0; th[3] = 1;
					// analysis frames whose window lies in this call's input ([5], taken by kAnalyseTeams) / reaches into the history ([6])
					for (int which = 0; which < ((f & HOP_REANALYSE_PREV) ? 2 : 1); ++which) th[analysisWindowInCall(d.B, d.M, d.I, list[h].inputOffset, which, nIn[s]) ? 5 : 6] = 1;
				}
				th[0] = 1;
				if (!(f & HOP_NEW_SPECTRUM)) th[9] = 1; // (a hop that re-uses the spectrum before it: the continuous wavefront asks for a new one per hop)
				if (f & HOP_MAPPED) th[1] = 1;
				if (f & HOP_FORMANTS) { th[2] = 1; if ((f & HOP_PREANALYSED) || params[s].formantBaseFreq <= 0) th[10] = 1; } // [10]: a stream that estimates its base frequency (:929-966)
				if (f & HOP_RANDOM_TF) th[4] = 1;
			}
			info[subS + sl] = lastNewLocal;
			if (cnt > 0) {
				LastHop &lh = lastHop[s];
				lh.slot = (sub*nTiles + t) & 1;
				lh.local = cnt - 1;
				lh.subLocal = sl;
				lh.mapped = (list[h1 - 1].flags & HOP_MAPPED) != 0;
				lh.formants = (list[h1 - 1].flags & HOP_FORMANTS) != 0;
			}
			int &span = maxSpanV[(size_t)sub*nTiles + t];
			span = std::max(span, ed.nHi - ed.nLo);
		}
	}
	SMST_HIP(hipMemcpyAsync(dHops, hopsAll, needHops*sizeof(HopDesc), hipMemcpyHostToDevice, stGate));
	SMST_HIP(hipMemcpyAsync(dEmit, emitAll, needEmit*sizeof(EmitDesc), hipMemcpyHostToDevice, stGate));
	SMST_HIP(hipMemcpyAsync(dTileInfo, tileInfo, needInfo*sizeof(int), hipMemcpyHostToDevice, stGate));
	if (anyPass) SMST_HIP(hipMemcpyAsync(dFlags, passFlags, S*sizeof(int), hipMemcpyHostToDevice, stGate));
	if (anyPendAnalysis) SMST_HIP(hipMemcpyAsync(cs.pendHops, cs.hPendHops, S*sizeof(HopDesc), hipMemcpyHostToDevice, stGate));
	// the kernels below are ordered after the uploads by an event, not by the host
	SMST_HIP(hipEventRecord(cs.tables, stGate));
	SMST_HIP(hipStreamWaitEvent(st, cs.tables, 0));

	d.hops = dHops;
	d.emit = dEmit;
	d.hopStride = hopStride;
	d.emitStride = nTiles;

	// no stream fires a hop (most quanta of a real-time host): the output is the front of the carried sums, and -- while the rows have room --
	// what is left of them stays in place (kEmitCarried)
	bool carriedOnly = maxHops == 0 && carriedEmit;
	for (int s = 0; s < S && carriedOnly; ++s) carriedOnly = carryBase[s] + (passFlags[s] ? 0 : nOut[s]) + d.carryLen <= d.carryPitch;
	if (carriedOnly) for (int s = 0; s < S; ++s) carryBase[s] += passFlags[s] ? 0 : nOut[s];
	runTiles(TileRun{&io, nTiles, maxHops, tileHasV.data(), maxSpanV.data(), dTileInfo, false, nullptr, carriedOnly});
	if (anyPendAnalysis) timed(timings.analyseMs, [&] {
		// the blocks left in flight: their spectra (Band.input and, where :303 asks for it, the re-analysed Band.prevInput) into the pending
		// buffers, laid out as a one-hop tile over ALL streams
		DevBatch dp = d;
		dp.hops = cs.pendHops;
		dp.hopStride = 1;
		dp.T = 1;
		dp.Xcur = dPendIn;
		dp.Xprev = dPendPrev;
		launchAnalyse(dp, io, 0, S, 0, 1, pendInCall, pendLate, st);
		if (profiling) ++timings.analyseLaunches;
	});
	timed(timings.otherMs, [&] {
		if (anyPass) launchPassThrough(d, io, dFlags, maxOut, st);
		// the input history slides (kHistory): the host follows each stream's window to size the launch
		int span = 0;
		for (int s = 0; s < S; ++s) {
			const int n = nIn[s];
			if (histBase[s] + d.histLen + n <= d.histPitch) { histBase[s] += n; span = std::max(span, n); }
			else { histBase[s] = 0; span = std::max(span, d.histLen); }
		}
		launchHistory(d, io, span, st);
	});
	d.histCur ^= 1;
	SMST_HIP(hipEventRecord(cs.done, st));
	cs.used = true;
	SMST_HIP(hipGetLastError());
	hostTimes.callMs += msSince(hostT0);
	++hostTimes.calls;
}

// ---- seek -------------------------------------------------------------------------------------------------
void Batch::seek(const float *in, long long inSS, long long inCS, const int *inSamples, const double *rates, const unsigned char *active) {
	SMST_HIP(hipSetDevice(dev));
	// pinned staging of its own ([S] sample counts, [S] flags, [S] history lengths, energy partials): one host
	// synchronisation (the energy readback that decides about the silence counter, :159-162)
	if (!hSeek) {
		hSeek = pinnedAlloc<int>((size_t)3*S);
		hSeekEnergy = pinnedAlloc<float>((size_t)S*kEnergyParts);
	} else {
		SMST_HIP(hipStreamSynchronize(st)); // an earlier seek's uploads may still be in flight
	}
	int *nIn = hSeek, *flags = hSeek + S, *histN = hSeek + 2*S;
	for (int s = 0; s < S; ++s) {
		const bool on = !active || active[s];
		flags[s] = on ? 1 : 0;
		nIn[s] = on ? inSamples[s] : 0;
		histN[s] = d.histLen;
		if (nIn[s] < 0) throw Error("negative sample count");
	}
	SMST_HIP(hipMemcpyAsync(dInSamples, nIn, S*sizeof(int), hipMemcpyHostToDevice, st));
	SMST_HIP(hipMemcpyAsync(dFlags, flags, S*sizeof(int), hipMemcpyHostToDevice, st));
	SMST_HIP(hipMemcpyAsync(dAux0, histN, S*sizeof(int), hipMemcpyHostToDevice, st));
	IoArgs io{in, nullptr, inSS, inCS, 0, 0, dInSamples, dOutSamples};
	launchSeekHistory(d, io, dFlags, st);
	d.histCur ^= 1;
	// energy of the copied part only (:144-154) = energy over the new history (the zero padding adds nothing)
	for (int s = 0; s < S; ++s) if (flags[s]) histBase[s] = 0;
	IoArgs ioE{d.hist, nullptr, (long long)C*d.histPitch, (long long)d.histPitch, 0, 0, dAux0, dOutSamples}; // (the streams that seek: their windows are at the front of the rows now)
	launchEnergy(d, ioE, 0, S, d.histLen, dEnergy, st);
	SMST_HIP(hipMemcpyAsync(hSeekEnergy, dEnergy, (size_t)S*kEnergyParts*sizeof(float), hipMemcpyDeviceToHost, st));
	SMST_HIP(hipStreamSynchronize(st));
	for (int s = 0; s < S; ++s) {
		if (!flags[s]) continue;
		float e = 0;
		for (int p = 0; p < kEnergyParts; ++p) e += hSeekEnergy[(size_t)s*kEnergyParts + p];
		StreamSched &sc = sched[s];
		if (e >= kNoiseFloor) { // :159-162
			sc.silenceCounter = 0;
			sc.silenceFirst = true;
		}
		sc.didSeek = true; // :163
		const double rate = rates ? rates[s] : 1.0;
		sc.seekTimeFactor = (rate*I > 1) ? float(1/rate) : float(I); // :164
	}
}

// ---- flush ------------------------------------------------------------------------------------------------
void Batch::flush(float *out, long long outSS, long long outCS, const int *outSamples, const float *rates, const unsigned char *active) {
	SMST_HIP(hipSetDevice(dev));
	std::vector<int> blockOut(S, 0), blockIn(S, 0), tail(S, -1), tailOff(S, 0), outOff(S, 0);
	std::vector<unsigned char> runBlock(S, 0), on(S, 0);
	bool anyBlock = false;
	int maxIn = 0;
	for (int s = 0; s < S; ++s) {
		if (active && !active[s]) continue;
		on[s] = 1;
		const int n = outSamples[s];
		if (n < 0) throw Error("negative sample count");
		const int outputBlock = std::max(0, n - I); // :439
		const float rate = rates ? rates[s] : 0.0f;
		if (outputBlock > 0) {
			runBlock[s] = 1;
			anyBlock = true;
			blockOut[s] = outputBlock;
			blockIn[s] = int(outputBlock*rate); // :440
			maxIn = std::max(maxIn, blockIn[s]);
		}
		tail[s] = n - outputBlock;
		outOff[s] = outputBlock;
	}
	if (anyBlock) {
		if ((size_t)maxIn + 1 > zerosCapacity) {
			if (dZeros) { SMST_HIP(hipStreamSynchronize(st)); devFree(dZeros); }
			zerosCapacity = (size_t)maxIn + 1 + 4096;
			dZeros = devAlloc<float>(zerosCapacity);
			// process() reads this on another stream (the silence gate runs on stGate): complete the fill before going on
			SMST_HIP(hipMemsetAsync(dZeros, 0, zerosCapacity*sizeof(float), st));
			SMST_HIP(hipStreamSynchronize(st));
		}
		process(dZeros, 0, 0, blockIn.data(), out, outSS, outCS, blockOut.data(), runBlock.data());
	}
	// Split computation, between two interval boundaries: how far has the block in flight come?  (:321-325 spread its steps over the
	// interval; flush() reads the REAL ring, one interval ahead of the stashed one that process() emits from, then stft.reset(0.1) and
	// prevInput = output = 0, :442-463 -- and the block's remaining steps run afterwards, on what the flush left.  Pinned step by step by
	// tests/golden/split_events.)
	std::vector<int> synthChannels;
	pendList.clear();
	for (int s = 0; s < S; ++s) {
		if (!on[s] || !split || !pend[s].valid) continue;
		PendingBlock &pb = pend[s];
		const StepLayout l = stepLayout(pb.flags);
		const size_t e = stepsExecuted(size_t(l.steps), sched[s].samplesSinceLast);
		if (e > size_t(l.synth0)) {
			// the spectrum is complete and e - synth0 channels have been synthesised into the ring (all of them: the block is finished): run
			// the block now; the other channels' frames never come -- stft.reset() clears the spectrum they would be made from
			if (synthChannels.empty()) synthChannels.assign(S, -1);
			synthChannels[s] = (e >= size_t(l.steps)) ? -1 : int(e) - l.synth0;
			pendList.push_back(s);
		} else if (e >= size_t(l.spectrum)) {
			// output (and, after `.input -> .prevInput`, prevInput) were complete and are zeroed; what remains is a silent frame: its window product
			pb.startBin = M;
			pb.zeroPrevAfter = true;
		} else if (e > size_t(l.main0)) {
			// e - main0 chunks of the main prediction ran: those bins are zero now, the later chunks start from zeros (:725-726)
			pb.startBin = std::max(pb.startBin, int(size_t(M)*(e - size_t(l.main0))/8));
		} else if (l.copyIn >= 0 && e > size_t(l.analyse0) && e <= size_t(l.copyIn)) {
			// e - analyse0 channels were analysed, not yet copied into Band.input: stft.reset() cleared their spectra (:356-373)
			const int n = std::min(C, int(e) - l.analyse0);
			SMST_HIP(hipMemsetAsync(dPendIn + (size_t)s*C*d.Mp, 0, (size_t)n*d.Mp*sizeof(float2), st));
		}
	}
	if (!pendList.empty()) runPendingBlocks(synthChannels.data());
	for (int s = 0; s < S; ++s) {
		if (!on[s]) continue;
		const StreamSched &sc = sched[s];
		// where the L1 output ring the reference reads here sits relative to our carry (split: the "ahead" ring)
		tailOff[s] = (split && sc.samplesSinceLast != SIZE_MAX && sc.samplesSinceLast < size_t(I)) ? int(size_t(I) - sc.samplesSinceLast) : 0;
	}
	SMST_HIP(hipMemcpyAsync(dOutSamples, tail.data(), S*sizeof(int), hipMemcpyHostToDevice, st));
	SMST_HIP(hipMemcpyAsync(dAux0, tailOff.data(), S*sizeof(int), hipMemcpyHostToDevice, st));
	SMST_HIP(hipMemcpyAsync(dAux1, outOff.data(), S*sizeof(int), hipMemcpyHostToDevice, st));
	SMST_HIP(hipStreamSynchronize(st));
	IoArgs io{nullptr, out, 0, 0, outSS, outCS, dInSamples, dOutSamples};
	settleCarry();
	launchFlushTail(d, io, dAux0, dAux1, st);
	// stft.reset(0.1) + zero prevInput/output (:456-463).  Split computation: the samples up to the end of the interval still come from
	// the stashed ring (:407-415), which the reset does not touch -- the fresh ring begins behind them
	for (int s = 0; s < S; ++s) { resetBitsV[s] = on[s] ? (1 | 4 | 8) : 0; keepV[s] = on[s] ? tailOff[s] : 0; if (on[s]) lastHop[s] = LastHop(); }
	resetStreams(resetBitsV.data(), 0, split ? keepV.data() : nullptr);
	SMST_HIP(hipGetLastError());
}

// ---- outputSeek -------------------------------------------------------------------------------------------
void Batch::outputSeek(const float *in, long long inSS, long long inCS, const int *inputLengths) {
	// signalsmith-stretch.h:173-204
	reset();
	const int outLat = outputLatency(), inLat = inputLatency();
	std::vector<int> seekSamples(S), surplus(S), preOut(S, outLat);
	std::vector<double> rates(S);
	for (int s = 0; s < S; ++s) {
		surplus[s] = std::max(inputLengths[s] - inLat, 0);
		float playbackRate = surplus[s]/float(outLat);
		seekSamples[s] = inputLengths[s] - surplus[s];
		rates[s] = playbackRate;
	}
	seek(in, inSS, inCS, seekSamples.data(), rates.data());
	// pre-roll output into scratch [S][C][outLat]
	const size_t need = (size_t)S*C*outLat;
	if (need > scratchOutCapacity) {
		if (dScratchOut) { SMST_HIP(hipStreamSynchronize(st)); devFree(dScratchOut); }
		scratchOutCapacity = need;
		dScratchOut = devAlloc<float>(need);
	}
	// offset input per stream: seekSamples differ per stream, so pass per-stream start through a shifted base when uniform,
	// otherwise run stream groups with equal offsets
	std::vector<int> order(S);
	for (int s = 0; s < S; ++s) order[s] = s;
	std::vector<unsigned char> mask(S);
	std::vector<int> done(S, 0);
	for (int s0 = 0; s0 < S; ++s0) {
		if (done[s0]) continue;
		std::fill(mask.begin(), mask.end(), 0);
		for (int s = s0; s < S; ++s) if (!done[s] && seekSamples[s] == seekSamples[s0]) { mask[s] = 1; done[s] = 1; }
		process(in + seekSamples[s0], inSS, inCS, surplus.data(), dScratchOut, (long long)C*outLat, outLat, preOut.data(), mask.data());
	}
	// "put the thing down, flip it and reverse it" (:198-203): negate, reverse, add into the output ring
	std::vector<int> off(S, 0);
	for (int s = 0; s < S; ++s) {
		const StreamSched &sc = sched[s];
		off[s] = (split && sc.samplesSinceLast != SIZE_MAX && sc.samplesSinceLast < size_t(I)) ? int(size_t(I) - sc.samplesSinceLast) : 0;
	}
	SMST_HIP(hipMemcpyAsync(dAux0, off.data(), S*sizeof(int), hipMemcpyHostToDevice, st));
	SMST_HIP(hipStreamSynchronize(st));
	settleCarry();
	launchAddPreRoll(d, dScratchOut, outLat, dAux0, st);
	SMST_HIP(hipGetLastError());
}

// ---- copy -------------------------------------------------------------------------------------------------
void Batch::copyStateFrom(Batch &o) {
	if (o.S != S || o.C != C || o.B != B || o.I != I || o.split != split || o.halfState != halfState) throw Error("copyStateFrom: the two batches differ in geometry");
	SMST_HIP(hipSetDevice(o.dev));
	SMST_HIP(hipStreamSynchronize(o.st));
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipStreamSynchronize(st));
	const size_t bandRows = (size_t)S*C*M, scale = halfState ? 2 : 1;
	auto copy = [&](void *dst, const void *src, size_t bytes) { SMST_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDefault)); };
	copy(d.stInput, o.d.stInput, bandRows*sizeof(float2));
	copy(d.stPrev, o.d.stPrev, bandRows*sizeof(float2));
	copy(d.stOut, o.d.stOut, bandRows*sizeof(float2)/scale);
	copy(d.stEnergy, o.d.stEnergy, bandRows*sizeof(float)/scale);
	copy(d.hist, o.d.hist, (size_t)S*C*d.histPitch*sizeof(float));
	histBase = o.histBase;
	carryBase = o.carryBase;
	for (int h = 0; h < 2; ++h) {
		copy(d.histBase[h], o.d.histBase[h], (size_t)S*sizeof(int));
		copy(d.carrySum[h], o.d.carrySum[h], (size_t)S*C*d.carryPitch*sizeof(float)/scale);
		copy(d.carryWp[h], o.d.carryWp[h], (size_t)S*d.carryPitch*sizeof(float));
		copy(d.carryBase[h], o.d.carryBase[h], (size_t)S*sizeof(int));
	}
	copy(d.stFreq, o.d.stFreq, (size_t)S*2*sizeof(float));
	d.histCur = o.d.histCur;
	d.carryCur = o.d.carryCur;
	sched = o.sched;
	params = o.params;
	paramsDirty = true;
	pend = o.pend;
	lastSteps = o.lastSteps; // smst_block_steps of a clone = the original's newest block
	lastStarts = o.lastStarts;
	if (split) {
		copy(dPendIn, o.dPendIn, (size_t)S*C*d.Mp*sizeof(float2));
		copy(dPendPrev, o.dPendPrev, (size_t)S*C*d.Mp*sizeof(float2));
	}
	if (o.d.mapTableLen > 0) {
		if (dMapTable) devFree(dMapTable);
		dMapTable = devAlloc<float>((size_t)S*kMapSlots*o.d.mapTableLen);
		hostMapTable = o.hostMapTable;
		SMST_HIP(hipMemcpy(dMapTable, hostMapTable.data(), hostMapTable.size()*sizeof(float), hipMemcpyHostToDevice));
		d.mapTableLen = o.d.mapTableLen;
		d.mapTable = dMapTable;
	}
	for (auto &lh : lastHop) lh = LastHop();
}

void Batch::inheritAcrossConfigure(Batch &o) {
	if (o.S != S) throw Error("inheritAcrossConfigure: stream counts differ");
	SMST_HIP(hipSetDevice(o.dev));
	SMST_HIP(hipStreamSynchronize(o.st));
	std::vector<float> freq((size_t)S*2);
	SMST_HIP(hipMemcpy(freq.data(), o.d.stFreq, freq.size()*sizeof(float), hipMemcpyDeviceToHost));
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipStreamSynchronize(st));
	SMST_HIP(hipMemcpy(d.stFreq, freq.data(), freq.size()*sizeof(float), hipMemcpyHostToDevice));
	for (int s = 0; s < S; ++s) {
		const StreamSched &from = o.sched[s];
		StreamSched &to = sched[s];
		to.seed = o.seedAfterDroppedBlock(s); // (configure() drops a block in flight: blockProcess = {}, :89)
		to.prevInputOffset = from.prevInputOffset;
		to.didSeek = from.didSeek;
		to.seekTimeFactor = from.seekTimeFactor;
		to.silenceCounter = from.silenceCounter;
		to.silenceFirst = from.silenceFirst;
	}
}

// ---- test hooks -------------------------------------------------------------------------------------------
void Batch::debugGetState(int stream, int which, float *dst) {
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipStreamSynchronize(st));
	const size_t off = (size_t)stream*C*M, n = (size_t)C*M;
	if (halfState && (which == 2 || which == 3)) { // fp16 storage: widen on the host (energy is stored as its square root)
		const size_t count = which == 2 ? 2*n : n;
		std::vector<half_t> tmp(count);
		const half_t *src = which == 2 ? reinterpret_cast<const half_t *>(d.stOut) + 2*off : reinterpret_cast<const half_t *>(d.stEnergy) + off;
		SMST_HIP(hipMemcpy(tmp.data(), src, count*sizeof(half_t), hipMemcpyDeviceToHost));
		for (size_t i = 0; i < count; ++i) { const float v = float(tmp[i]); dst[i] = which == 3 ? v*v : v; }
		return;
	}
	if (which == 3) {
		SMST_HIP(hipMemcpy(dst, d.stEnergy + off, n*sizeof(float), hipMemcpyDeviceToHost));
		return;
	}
	const float2 *src = which == 0 ? d.stInput : (which == 1 ? d.stPrev : d.stOut);
	SMST_HIP(hipMemcpy(dst, src + off, n*sizeof(float2), hipMemcpyDeviceToHost));
}
void Batch::debugSetState(int stream, int which, const float *src) {
	if (stream < 0 || stream >= S || which < 0 || which > 3) throw Error("debugSetState: bad stream / selector");
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipStreamSynchronize(st));
	const size_t off = (size_t)stream*C*M, n = (size_t)C*M;
	if (halfState && (which == 2 || which == 3)) {
		const size_t count = which == 2 ? 2*n : n;
		std::vector<half_t> tmp(count);
		for (size_t i = 0; i < count; ++i) tmp[i] = half_t(which == 3 ? std::sqrt(std::max(src[i], 0.0f)) : src[i]);
		half_t *dst = which == 2 ? reinterpret_cast<half_t *>(d.stOut) + 2*off : reinterpret_cast<half_t *>(d.stEnergy) + off;
		SMST_HIP(hipMemcpy(dst, tmp.data(), count*sizeof(half_t), hipMemcpyHostToDevice));
		return;
	}
	if (which == 3) {
		SMST_HIP(hipMemcpy(d.stEnergy + off, src, n*sizeof(float), hipMemcpyHostToDevice));
		return;
	}
	float2 *dst = which == 0 ? d.stInput : (which == 1 ? d.stPrev : d.stOut);
	SMST_HIP(hipMemcpy(dst + off, src, n*sizeof(float2), hipMemcpyHostToDevice));
}
void Batch::debugSetCarry(int stream, const float *sums, const float *products) {
	if (stream < 0 || stream >= S) throw Error("debugSetCarry: bad stream");
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipStreamSynchronize(st));
	settleCarry();
	SMST_HIP(hipStreamSynchronize(st));
	// [C] rows of B+I values, carryPitch apart on the device
	const size_t n = (size_t)C*d.carryLen, off = (size_t)stream*C*d.carryPitch, CL = d.carryLen, CP = d.carryPitch;
	if (halfState) {
		std::vector<half_t> tmp(n);
		for (size_t i = 0; i < n; ++i) tmp[i] = half_t(sums[i]);
		SMST_HIP(hipMemcpy2D(reinterpret_cast<half_t *>(d.carrySum[d.carryCur]) + off, CP*sizeof(half_t), tmp.data(), CL*sizeof(half_t), CL*sizeof(half_t), C, hipMemcpyHostToDevice));
	} else {
		SMST_HIP(hipMemcpy2D(d.carrySum[d.carryCur] + off, CP*sizeof(float), sums, CL*sizeof(float), CL*sizeof(float), C, hipMemcpyHostToDevice));
	}
	SMST_HIP(hipMemcpy(d.carryWp[d.carryCur] + (size_t)stream*CP, products, CL*sizeof(float), hipMemcpyHostToDevice));
}
bool Batch::debugGetMap(int stream, float *dst) {
	if (stream < 0 || stream >= S) throw Error("debugGetMap: bad stream");
	const LastHop &lh = lastHop[stream];
	if (lh.slot < 0 || !lh.mapped) return false;
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipStreamSynchronize(st));
	SMST_HIP(hipMemcpy(dst, slots[lh.slot].map + ((size_t)lh.subLocal*d.T + lh.local)*M, (size_t)M*sizeof(float2), hipMemcpyDeviceToHost));
	return true;
}
bool Batch::debugGetFormants(int stream, float *ratio, float *envelope, float *freqEstimate) {
	if (stream < 0 || stream >= S) throw Error("debugGetFormants: bad stream");
	const LastHop &lh = lastHop[stream];
	if (lh.slot < 0 || !lh.formants || !slots[lh.slot].envelope) return false;
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipStreamSynchronize(st));
	const size_t row = (size_t)lh.subLocal*d.T + lh.local;
	SMST_HIP(hipMemcpy(ratio, slots[lh.slot].ratio + row*M, (size_t)M*sizeof(float), hipMemcpyDeviceToHost));
	SMST_HIP(hipMemcpy(envelope, slots[lh.slot].envelope + row*M, (size_t)M*sizeof(float), hipMemcpyDeviceToHost));
	SMST_HIP(hipMemcpy(freqEstimate, slots[lh.slot].freqEst + row, sizeof(float), hipMemcpyDeviceToHost));
	return true;
}
void Batch::debugGetCarry(int stream, float *sums, float *products) {
	SMST_HIP(hipSetDevice(dev));
	SMST_HIP(hipStreamSynchronize(st));
	settleCarry();
	SMST_HIP(hipStreamSynchronize(st));
	const size_t n = (size_t)C*d.carryLen, off = (size_t)stream*C*d.carryPitch, CL = d.carryLen, CP = d.carryPitch;
	if (halfState) {
		std::vector<half_t> tmp(n);
		SMST_HIP(hipMemcpy2D(tmp.data(), CL*sizeof(half_t), reinterpret_cast<const half_t *>(d.carrySum[d.carryCur]) + off, CP*sizeof(half_t), CL*sizeof(half_t), C, hipMemcpyDeviceToHost));
		for (size_t i = 0; i < n; ++i) sums[i] = float(tmp[i]);
	} else {
		SMST_HIP(hipMemcpy2D(sums, CL*sizeof(float), d.carrySum[d.carryCur] + off, CP*sizeof(float), CL*sizeof(float), C, hipMemcpyDeviceToHost));
	}
	SMST_HIP(hipMemcpy(products, d.carryWp[d.carryCur] + (size_t)stream*CP, CL*sizeof(float), hipMemcpyDeviceToHost));
}

} // namespace smst
