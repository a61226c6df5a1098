// What the reference-shaped API costs: N independent SignalsmithStretch<float> objects called from ONE thread through
// the C++ drop-in header (include/signalsmith-stretch/signalsmith-stretch.h: one single-stream engine per object, host buffers, a
// synchronous copy in and out per call) against the SAME N streams through smst_batch_* (host buffers as well: the PCIe copies are in both).
// Prints one JSON object.  usage: bench_dropin [streams=64] [seconds per call=1] [calls=6]
#include "signalsmith-stretch/signalsmith-stretch.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
	const int N = argc > 1 ? atoi(argv[1]) : 64;
	const double seconds = argc > 2 ? atof(argv[2]) : 1.0;
	const int calls = argc > 3 ? atoi(argv[3]) : 6;
	const int C = 2, sr = 48000;
	const int nIn = int(seconds*sr), nOut = int(nIn*1.5);
	// the bench's sine streams, one second of them, repeated per call
	std::vector<std::vector<std::vector<float>>> in(N, std::vector<std::vector<float>>(C, std::vector<float>(nIn)));
	for (int s = 0; s < N; ++s) {
		const double f1 = 110.0*std::pow(2.0, (s % 37)/12.0);
		for (int c = 0; c < C; ++c)
			for (int i = 0; i < nIn; ++i) {
				const double t = double(i)/sr;
				in[s][c][i] = float(0.4*std::sin(2*M_PI*f1*t + 0.5*c) + 0.2*std::sin(2*M_PI*3.17*f1*t));
			}
	}
	using Stretch = signalsmith::stretch::SignalsmithStretch<float>;
	double objectsSeconds = 0, batchSeconds = 0, check = 0;
	{
		std::vector<Stretch> objects;
		for (int s = 0; s < N; ++s) { objects.emplace_back(long(s)); objects[s].presetDefault(C, float(sr)); } // seed s = stream s of a batch seeded 0 (the hop after a reset draws random time factors)
		std::vector<std::vector<float>> out(C, std::vector<float>(nOut));
		for (int call = -1; call < calls; ++call) { // call -1: warm-up (first-touch allocations, kernel code upload)
			const double t0 = now();
			for (int s = 0; s < N; ++s) objects[s].process(in[s], nIn, out, nOut);
			if (call >= 0) objectsSeconds += now() - t0;
		}
		check = out[0][nOut/2]; // (the last object's channel 0)
	}
	{
		smst_batch *b = nullptr;
		if (smst_batch_create_preset_ex(&b, N, C, 0, float(sr), -1, smst_default_device(), 0, 0) != SMST_OK) { std::fprintf(stderr, "batch: %s\n", smst_last_error()); return 1; }
		std::vector<float> flatIn((size_t)N*C*nIn), flatOut((size_t)N*C*nOut);
		for (int s = 0; s < N; ++s) for (int c = 0; c < C; ++c) std::copy(in[s][c].begin(), in[s][c].end(), flatIn.begin() + ((size_t)s*C + c)*nIn);
		std::vector<int> ni(N, nIn), no(N, nOut);
		for (int call = -1; call < calls; ++call) {
			const double t0 = now();
			if (smst_batch_process(b, flatIn.data(), (long long)C*nIn, nIn, ni.data(), flatOut.data(), (long long)C*nOut, nOut, no.data(), SMST_MEM_HOST) != SMST_OK) { std::fprintf(stderr, "process: %s\n", smst_last_error()); return 1; }
			smst_batch_synchronize(b);
			if (call >= 0) batchSeconds += now() - t0;
		}
		check -= flatOut[((size_t)(N - 1)*C)*nOut + nOut/2];
		smst_batch_destroy(b);
	}
	const double samples = double(N)*C*(nIn + nOut)*calls;
	std::printf("{\"streams\": %d, \"channels\": %d, \"seconds_per_call\": %.2f, \"calls\": %d, \"objects_one_thread_Msamples_per_s\": %.1f, \"batch_api_host_buffers_Msamples_per_s\": %.1f, "
	            "\"batch_over_objects\": %.2f, \"objects_ms_per_call_per_object\": %.3f, \"same_output\": %s}\n",
	            N, C, seconds, calls, samples/objectsSeconds/1e6, samples/batchSeconds/1e6, objectsSeconds/batchSeconds, objectsSeconds/calls/N*1e3, check == 0 ? "true" : "false");
	return 0;
}
