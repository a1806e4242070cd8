// wav.h -- minimal RIFF/WAVE reader + PCM16 writer for the `zen` command line tool.
// Stands where the reference links vendor/libnyquist (94 kLoC of decoders); only the conversions Zen
// actually relies on are reproduced, with the same arithmetic:
//   PCM16 -> float : s / 32767.f                 (libnyquist Common.h:288,296)
//   stereo -> mono : (L + R) / 2.0f              (libnyquist Common.h:669-675, zen/offline.h:106-113)
//   float -> PCM16 : (int16) lroundf(s * 32767.f), no dither (libnyquist Common.cpp:332-337)
//   file layout    : 44-byte canonical header, data, pad byte if odd (libnyquist Encoders.cpp:117-196)
#ifndef ZG_CLI_WAV_H
#define ZG_CLI_WAV_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace zen {
namespace wav {
	struct AudioData {
		int channelCount = 0;
		int sampleRate = 0;
		double lengthSeconds = 0;
		std::size_t frameSize = 0; // bytes per frame, as libnyquist reports it
		std::vector<float> samples; // interleaved
	};

	inline uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
	inline uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

	inline void load(AudioData& out, const std::string& path)
	{
		std::ifstream f(path, std::ios::binary);
		if (!f)
			throw std::runtime_error("cannot open " + path);
		std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
		if (buf.size() < 12 || memcmp(buf.data(), "RIFF", 4) || memcmp(buf.data() + 8, "WAVE", 4))
			throw std::runtime_error(path + ": not a RIFF/WAVE file");
		std::size_t pos = 12;
		int format = 0, bits = 0;
		const unsigned char* data = nullptr;
		std::size_t data_bytes = 0;
		// Chunk sizes come from the file: every read below is bounded by what the buffer really holds.
		while (pos + 8 <= buf.size()) {
			const std::size_t sz = rd32(&buf[pos + 4]);
			const std::size_t avail = buf.size() - (pos + 8); // bytes of this chunk's body that exist
			const unsigned char* body = buf.data() + pos + 8;
			if (!memcmp(&buf[pos], "fmt ", 4) && sz >= 16) {
				if (avail < 16)
					throw std::runtime_error(path + ": truncated fmt chunk");
				format = rd16(body);
				out.channelCount = rd16(body + 2);
				out.sampleRate = (int)rd32(body + 4);
				out.frameSize = rd16(body + 12);
				bits = rd16(body + 14);
				if (format == 0xFFFE && sz >= 26 && avail >= 26) // WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
					format = rd16(body + 24);
			}
			else if (!memcmp(&buf[pos], "data", 4)) {
				data = body;
				data_bytes = std::min<std::size_t>(sz, avail);
			}
			if (sz > avail)
				break; // the last chunk runs past the end of the file
			pos += 8 + sz + (sz & 1);
		}
		if (!data || !bits)
			throw std::runtime_error(path + ": missing fmt/data chunk");
		if (out.channelCount <= 0 || out.sampleRate <= 0)
			throw std::runtime_error(path + ": bad channel count or sample rate");
		if (bits % 8 != 0 || bits < 8)
			throw std::runtime_error(path + ": unsupported WAV encoding (bits per sample must be a multiple of 8)");
		const std::size_t n = data_bytes / (std::size_t)(bits / 8);
		out.samples.resize(n);
		if (format == 1 && bits == 16) {
			for (std::size_t i = 0; i < n; ++i)
				out.samples[i] = (float)(int16_t)rd16(data + 2 * i) / 32767.f;
		}
		else if (format == 1 && bits == 24) {
			for (std::size_t i = 0; i < n; ++i) {
				int32_t v = data[3 * i] | (data[3 * i + 1] << 8) | (data[3 * i + 2] << 16);
				if (v & 0x800000)
					v |= ~0xFFFFFF;
				out.samples[i] = (float)v / 8388608.f;
			}
		}
		else if (format == 1 && bits == 32) {
			for (std::size_t i = 0; i < n; ++i)
				out.samples[i] = (float)(int32_t)rd32(data + 4 * i) / 2147483648.f;
		}
		else if (format == 3 && bits == 32) {
			memcpy(out.samples.data(), data, n * 4);
		}
		else {
			throw std::runtime_error(path + ": unsupported WAV encoding (need PCM 16/24/32 or float32)");
		}
		out.lengthSeconds = (double)(n / out.channelCount) / (double)out.sampleRate;
	}

	inline void stereo_to_mono(const float* stereo, float* mono, std::size_t n_interleaved)
	{
		for (std::size_t i = 0, j = 0; i + 1 < n_interleaved; i += 2, ++j)
			mono[j] = (stereo[i] + stereo[i + 1]) / 2.0f;
	}

	inline void encode_pcm16_mono(const std::vector<float>& x, int sampleRate, const std::string& path)
	{
		std::ofstream f(path, std::ios::binary);
		if (!f)
			throw std::runtime_error("cannot write " + path);
		const uint32_t data_bytes = (uint32_t)(x.size() * 2);
		const uint32_t pad = data_bytes & 1;
		unsigned char h[44];
		auto w32 = [&](int o, uint32_t v) { h[o] = v & 255; h[o + 1] = (v >> 8) & 255; h[o + 2] = (v >> 16) & 255; h[o + 3] = v >> 24; };
		auto w16 = [&](int o, uint16_t v) { h[o] = v & 255; h[o + 1] = v >> 8; };
		memcpy(h, "RIFF", 4);
		w32(4, 36 + data_bytes + pad);
		memcpy(h + 8, "WAVEfmt ", 8);
		w32(16, 16);
		w16(20, 1);
		w16(22, 1);
		w32(24, (uint32_t)sampleRate);
		w32(28, (uint32_t)sampleRate * 2);
		w16(32, 2);
		w16(34, 16);
		memcpy(h + 36, "data", 4);
		w32(40, data_bytes);
		f.write((const char*)h, 44);
		std::vector<int16_t> pcm(x.size());
		for (std::size_t i = 0; i < x.size(); ++i)
			pcm[i] = (int16_t)lroundf(x[i] * 32767.f);
		f.write((const char*)pcm.data(), data_bytes);
		if (pad)
			f.put(0);
	}
} // namespace wav
} // namespace zen

#endif
