// Minimal 16-bit PCM / 32-bit float WAV reader and writer for the command-line tool (the format either side of the
// hot path: the reference CLI reads and writes 16-bit WAV, cmd/main.cpp:20-21,33-42,85).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct WavData {
	unsigned sampleRate = 48000;
	unsigned channels = 1;
	std::vector<std::vector<float>> samples; // [channel][index]
	size_t length() const { return samples.empty() ? 0 : samples[0].size(); }
};

inline bool readWav(const std::string &path, WavData &wav, std::string &error) {
	FILE *f = std::fopen(path.c_str(), "rb");
	if (!f) { error = "cannot open " + path; return false; }
	std::vector<unsigned char> bytes;
	unsigned char buf[65536];
	size_t n;
	while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
	std::fclose(f);
	auto u32 = [&](size_t o) { return uint32_t(bytes[o]) | uint32_t(bytes[o + 1]) << 8 | uint32_t(bytes[o + 2]) << 16 | uint32_t(bytes[o + 3]) << 24; };
	auto u16 = [&](size_t o) { return uint16_t(bytes[o] | bytes[o + 1] << 8); };
	if (bytes.size() < 12 || std::memcmp(bytes.data(), "RIFF", 4) || std::memcmp(bytes.data() + 8, "WAVE", 4)) { error = "not a RIFF/WAVE file"; return false; }
	unsigned format = 0, bits = 0;
	size_t pos = 12;
	bool haveFmt = false;
	while (pos + 8 <= bytes.size()) {
		const uint32_t size = u32(pos + 4);
		const size_t body = pos + 8;
		if (!std::memcmp(bytes.data() + pos, "fmt ", 4) && body + 16 <= bytes.size()) {
			format = u16(body); wav.channels = u16(body + 2); wav.sampleRate = u32(body + 4); bits = u16(body + 14);
			if (format == 0xFFFE && body + 26 <= bytes.size()) format = u16(body + 24); // WAVE_FORMAT_EXTENSIBLE sub-format
			haveFmt = true;
		} else if (!std::memcmp(bytes.data() + pos, "data", 4)) {
			if (!haveFmt || wav.channels == 0) { error = "data chunk before fmt chunk"; return false; }
			const size_t avail = std::min<size_t>(size, bytes.size() - body);
			const size_t frameBytes = size_t(bits/8)*wav.channels;
			if (!((format == 1 && (bits == 16 || bits == 24)) || (format == 3 && bits == 32))) { error = "unsupported sample format"; return false; }
			const size_t frames = avail/frameBytes;
			wav.samples.assign(wav.channels, std::vector<float>(frames));
			for (size_t i = 0; i < frames; ++i) {
				for (unsigned c = 0; c < wav.channels; ++c) {
					const size_t o = body + i*frameBytes + c*(bits/8);
					float v;
					if (format == 3) { uint32_t w = u32(o); std::memcpy(&v, &w, 4); }
					else if (bits == 16) v = float(int16_t(u16(o)))/32768.0f;
					else v = float(int32_t(uint32_t(bytes[o]) << 8 | uint32_t(bytes[o + 1]) << 16 | uint32_t(bytes[o + 2]) << 24) >> 8)/8388608.0f;
					wav.samples[c][i] = v;
				}
			}
			return true;
		}
		pos = body + size + (size & 1);
	}
	error = "no data chunk";
	return false;
}

inline bool writeWav16(const std::string &path, const WavData &wav, std::string &error) {
	FILE *f = std::fopen(path.c_str(), "wb");
	if (!f) { error = "cannot create " + path; return false; }
	const uint32_t frames = uint32_t(wav.length()), dataBytes = frames*wav.channels*2;
	auto put32 = [&](uint32_t v) { unsigned char b[4] = {(unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16), (unsigned char)(v >> 24)}; std::fwrite(b, 1, 4, f); };
	auto put16 = [&](uint16_t v) { unsigned char b[2] = {(unsigned char)v, (unsigned char)(v >> 8)}; std::fwrite(b, 1, 2, f); };
	std::fwrite("RIFF", 1, 4, f); put32(36 + dataBytes); std::fwrite("WAVEfmt ", 1, 8, f);
	put32(16); put16(1); put16(uint16_t(wav.channels)); put32(wav.sampleRate); put32(wav.sampleRate*wav.channels*2); put16(uint16_t(wav.channels*2)); put16(16);
	std::fwrite("data", 1, 4, f); put32(dataBytes);
	for (uint32_t i = 0; i < frames; ++i) {
		for (unsigned c = 0; c < wav.channels; ++c) {
			float v = wav.samples[c][i]*32768.0f;
			v = std::fmin(32767.0f, std::fmax(-32768.0f, std::round(v)));
			put16(uint16_t(int16_t(v)));
		}
	}
	std::fclose(f);
	return true;
}
