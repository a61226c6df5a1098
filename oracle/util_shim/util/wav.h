// TEST INFRASTRUCTURE ONLY (oracle): minimal stand-in for `geraintluff/util` wav.h exposing what cmd/main.cpp:31-42,
// 73-85 uses: read/write of 16-bit PCM WAV, `channels`, `sampleRate`, `length()`, `resize()`, `offset`, `wav[c][i]`.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

class Wav {
	std::vector<double> interleaved;
public:
	struct Result {
		bool ok;
		std::string reason;
		const Result &warn() const {
			if (!ok) std::cerr << "WAV error: " << reason << "\n";
			return *this;
		}
		operator bool() const { return ok; }
	};
	unsigned sampleRate = 48000;
	unsigned channels = 1;
	size_t offset = 0;

	size_t length() const { return channels ? interleaved.size()/channels - offset : 0; }
	void resize(size_t frames) { interleaved.resize((frames + offset)*channels, 0.0); }

	struct Channel {
		Wav &wav;
		unsigned channel;
		double &operator[](size_t i) { return wav.interleaved[(i + wav.offset)*wav.channels + channel]; }
	};
	Channel operator[](unsigned c) { return Channel{*this, c}; }

	Result read(const std::string &path) {
		FILE *f = std::fopen(path.c_str(), "rb");
		if (!f) return {false, "cannot open " + path};
		std::vector<unsigned char> bytes;
		unsigned char buf[65536];
		size_t n;
		while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
		std::fclose(f);
		auto u32 = [&](size_t o) { return uint32_t(bytes[o]) | uint32_t(bytes[o + 1]) << 8 | uint32_t(bytes[o + 2]) << 16 | uint32_t(bytes[o + 3]) << 24; };
		auto u16 = [&](size_t o) { return uint16_t(bytes[o] | bytes[o + 1] << 8); };
		if (bytes.size() < 12 || std::memcmp(bytes.data(), "RIFF", 4) || std::memcmp(bytes.data() + 8, "WAVE", 4)) return {false, "not RIFF/WAVE"};
		unsigned bits = 0;
		size_t pos = 12;
		while (pos + 8 <= bytes.size()) {
			uint32_t size = u32(pos + 4);
			size_t body = pos + 8;
			if (!std::memcmp(bytes.data() + pos, "fmt ", 4)) {
				channels = u16(body + 2);
				sampleRate = u32(body + 4);
				bits = u16(body + 14);
			} else if (!std::memcmp(bytes.data() + pos, "data", 4)) {
				if (bits != 16) return {false, "only 16-bit PCM"};
				size_t frames = std::min<size_t>(size, bytes.size() - body)/(2*channels);
				interleaved.resize(frames*channels);
				for (size_t i = 0; i < frames*channels; ++i) interleaved[i] = double(int16_t(u16(body + 2*i)))/32768.0;
				offset = 0;
				return {true, ""};
			}
			pos = body + size + (size & 1);
		}
		return {false, "no data chunk"};
	}
	Result write(const std::string &path) {
		FILE *f = std::fopen(path.c_str(), "wb");
		if (!f) return {false, "cannot create " + path};
		uint32_t frames = uint32_t(interleaved.size()/channels), dataBytes = frames*channels*2;
		auto put32 = [&](uint32_t v) { unsigned char b[4] = {(unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16), (unsigned char)(v >> 24)}; std::fwrite(b, 1, 4, f); };
		auto put16 = [&](uint16_t v) { unsigned char b[2] = {(unsigned char)v, (unsigned char)(v >> 8)}; std::fwrite(b, 1, 2, f); };
		std::fwrite("RIFF", 1, 4, f); put32(36 + dataBytes); std::fwrite("WAVEfmt ", 1, 8, f);
		put32(16); put16(1); put16(uint16_t(channels)); put32(sampleRate); put32(sampleRate*channels*2); put16(uint16_t(channels*2)); put16(16);
		std::fwrite("data", 1, 4, f); put32(dataBytes);
		for (size_t i = 0; i < size_t(frames)*channels; ++i) {
			double v = std::round(interleaved[i]*32768.0);
			v = std::fmin(32767.0, std::fmax(-32768.0, v));
			put16(uint16_t(int16_t(v)));
		}
		std::fclose(f);
		return {true, ""};
	}
};
