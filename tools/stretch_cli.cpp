// Command-line front-end over the C ABI (include/smst.h): the same job as the reference's example CLI
// (cmd/main.cpp:11-86 -- WAV in, outputSeek / process / flush, WAV out; same flags and defaults), plus a batch mode
// that renders MANY files with one geometry in a single batched GPU call (the data-parallel axis of this
// implementation).
//
//   stretch_cli [--semitones=S] [--formant=S] [--formant-comp] [--formant-base=Hz] [--tonality=Hz] [--time=F]
//               [--split-computation] [--device=N] in.wav out.wav [in2.wav out2.wav ...]
//
// Files given together must share sample rate and channel count (they form one batch); lengths may differ.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/smst.h"
#include "wav_io.h"

static double flagValue(int argc, char **argv, const char *name, double fallback) {
	const std::string prefix = std::string("--") + name + "=";
	for (int i = 1; i < argc; ++i) if (!std::strncmp(argv[i], prefix.c_str(), prefix.size())) return std::atof(argv[i] + prefix.size());
	return fallback;
}
static bool hasFlag(int argc, char **argv, const char *name) {
	const std::string flag = std::string("--") + name;
	for (int i = 1; i < argc; ++i) if (flag == argv[i]) return true;
	return false;
}
#define CHECK(call) do { if ((call) != SMST_OK) { std::fprintf(stderr, "%s: %s\n", #call, smst_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
	if (hasFlag(argc, argv, "v")) { int v[3]; smst_reference_version(v); std::printf("%d.%d.%d\n", v[0], v[1], v[2]); return 0; }
	const double semitones = flagValue(argc, argv, "semitones", 0), formants = flagValue(argc, argv, "formant", 0);
	const double formantBase = flagValue(argc, argv, "formant-base", 100), tonality = flagValue(argc, argv, "tonality", 8000);
	const double time = flagValue(argc, argv, "time", 1);
	const bool formantComp = hasFlag(argc, argv, "formant-comp"), split = hasFlag(argc, argv, "split-computation");
	const int device = int(flagValue(argc, argv, "device", 0));
	std::vector<std::string> files;
	for (int i = 1; i < argc; ++i) if (std::strncmp(argv[i], "--", 2)) files.push_back(argv[i]);
	if (files.size() < 2 || files.size()%2) {
		std::fprintf(stderr, "usage: %s [flags] in.wav out.wav [in2.wav out2.wav ...]\n", argv[0]);
		return 2;
	}
	const int S = int(files.size()/2);
	std::vector<WavData> inputs(S);
	std::string error;
	for (int s = 0; s < S; ++s) {
		if (!readWav(files[2*s], inputs[s], error)) { std::fprintf(stderr, "%s\n", error.c_str()); return 1; }
		if (inputs[s].sampleRate != inputs[0].sampleRate || inputs[s].channels != inputs[0].channels) {
			std::fprintf(stderr, "all files of a batch must share sample rate and channel count\n");
			return 1;
		}
		std::printf("%s -> %s\n", files[2*s].c_str(), files[2*s + 1].c_str());
	}
	const int C = int(inputs[0].channels);
	const float sr = float(inputs[0].sampleRate);

	smst_batch *batch = nullptr;
	CHECK(smst_batch_create_preset(&batch, S, C, 0, sr, split ? 1 : 0, device, 0)); // presetDefault, cmd/main.cpp:45
	CHECK(smst_batch_set_transpose_semitones(batch, -1, float(semitones), float(tonality/sr)));  // :46
	CHECK(smst_batch_set_formant_semitones(batch, -1, float(formants), formantComp));             // :47
	CHECK(smst_batch_set_formant_base(batch, -1, float(formantBase/sr)));                         // :48
	const int inLat = smst_batch_input_latency(batch), outLat = smst_batch_output_latency(batch), interval = smst_batch_interval_samples(batch);

	// per-stream lengths of the three stages, exactly as cmd/main.cpp:55-82 computes them
	const int seekLength = smst_batch_output_seek_length(batch, float(1/time));
	std::vector<int> outLen(S), outIndex(S), inIndex(S), procIn(S), tail(S), seekLens(S, seekLength);
	int maxIn = 0, maxOut = 0;
	for (int s = 0; s < S; ++s) {
		const int n = int(inputs[s].length());
		outLen[s] = int(std::round(n*time));
		outIndex[s] = std::max(0, outLen[s] - interval);
		const int outputPos = outIndex[s] + outLat;
		const int inputPos = int(std::round(outputPos/time));
		inIndex[s] = std::max(inputPos + inLat, seekLength);
		procIn[s] = inIndex[s] - seekLength;
		tail[s] = outLen[s] - outIndex[s];
		maxIn = std::max(maxIn, std::max(inIndex[s], n));
		maxOut = std::max(maxOut, outLen[s]);
	}
	maxIn = std::max(maxIn, 1); maxOut = std::max(maxOut, 1);
	std::vector<float> in((size_t)S*C*maxIn, 0.0f), out((size_t)S*C*maxOut, 0.0f); // zero padding = inWav.resize(inputIndex), :73
	for (int s = 0; s < S; ++s) for (int c = 0; c < C; ++c)
		std::copy(inputs[s].samples[c].begin(), inputs[s].samples[c].end(), in.begin() + ((size_t)s*C + c)*maxIn);
	const long long iss = (long long)C*maxIn, ics = maxIn, oss = (long long)C*maxOut, ocs = maxOut;

	CHECK(smst_batch_output_seek(batch, in.data(), iss, ics, seekLens.data(), SMST_MEM_HOST));                                   // :58-59
	CHECK(smst_batch_process(batch, in.data() + seekLength, iss, ics, procIn.data(), out.data(), oss, ocs, outIndex.data(), SMST_MEM_HOST)); // :77-78
	// flush writes at each stream's own output offset: stage through a second buffer and splice
	std::vector<float> tails((size_t)S*C*std::max(interval, 1), 0.0f);
	CHECK(smst_batch_flush(batch, tails.data(), (long long)C*std::max(interval, 1), std::max(interval, 1), tail.data(), nullptr, SMST_MEM_HOST)); // :81-82
	for (int s = 0; s < S; ++s) for (int c = 0; c < C; ++c)
		std::copy(tails.begin() + ((size_t)s*C + c)*std::max(interval, 1), tails.begin() + ((size_t)s*C + c)*std::max(interval, 1) + tail[s],
		          out.begin() + ((size_t)s*C + c)*maxOut + outIndex[s]);

	for (int s = 0; s < S; ++s) {
		WavData result;
		result.sampleRate = inputs[s].sampleRate;
		result.channels = inputs[s].channels;
		result.samples.assign(C, std::vector<float>(outLen[s]));
		for (int c = 0; c < C; ++c) std::copy(out.begin() + ((size_t)s*C + c)*maxOut, out.begin() + ((size_t)s*C + c)*maxOut + outLen[s], result.samples[c].begin());
		if (!writeWav16(files[2*s + 1], result, error)) { std::fprintf(stderr, "%s\n", error.c_str()); return 1; }
	}
	smst_batch_destroy(batch);
	return 0;
}
