// test_wav.cpp -- zen_amd/cli/wav.h against well-formed, truncated and malformed RIFF headers (CPU only;
// built with -fsanitize=address,undefined by tests/test_sanitizers.py so that an over-read is an error,
// not luck).  Conversions checked: PCM16 -> float s/32767 (libnyquist Common.h:288), stereo -> mono
// (L+R)/2 (Common.h:669-675), float -> PCM16 round trip.
#include "../../zen_amd/cli/wav.h"

#include <cstdio>
#include <functional>

static int fails = 0;
#define CHECK(c)                                                     \
	do {                                                             \
		if (!(c)) {                                                  \
			std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); \
			++fails;                                                 \
		}                                                            \
	} while (0)

static std::vector<unsigned char> header(uint16_t fmt, uint16_t ch, uint32_t fs, uint16_t bits, uint32_t data_bytes,
                                         uint32_t fmt_size = 16)
{
	std::vector<unsigned char> h;
	auto p32 = [&](uint32_t v) { for (int i = 0; i < 4; ++i) h.push_back((v >> (8 * i)) & 255); };
	auto p16 = [&](uint16_t v) { h.push_back(v & 255); h.push_back(v >> 8); };
	auto tag = [&](const char* t) { h.insert(h.end(), t, t + 4); };
	tag("RIFF"); p32(36 + data_bytes); tag("WAVE"); tag("fmt "); p32(fmt_size);
	p16(fmt); p16(ch); p32(fs); p32(fs * ch * bits / 8); p16((uint16_t)(ch * bits / 8)); p16(bits);
	tag("data"); p32(data_bytes);
	return h;
}

static std::string write_tmp(const std::string& dir, const char* name, const std::vector<unsigned char>& b)
{
	const std::string path = dir + "/" + name;
	std::ofstream f(path, std::ios::binary);
	f.write((const char*)b.data(), (std::streamsize)b.size());
	return path;
}

static bool throws(const std::function<void()>& fn)
{
	try {
		fn();
	}
	catch (const std::runtime_error&) {
		return true;
	}
	return false;
}

int main(int argc, char** argv)
{
	const std::string dir = argc > 1 ? argv[1] : "/tmp";
	zen::wav::AudioData a;
	{ // well-formed stereo PCM16
		auto b = header(1, 2, 44100, 16, 8);
		const int16_t s[4] = {32767, -32767, 16384, 0};
		b.insert(b.end(), (const unsigned char*)s, (const unsigned char*)s + 8);
		zen::wav::load(a, write_tmp(dir, "ok.wav", b));
		CHECK(a.channelCount == 2 && a.sampleRate == 44100 && a.samples.size() == 4);
		CHECK(a.samples[0] == 1.0f && a.samples[1] == -1.0f && a.samples[2] == 16384 / 32767.f);
		std::vector<float> mono(2);
		zen::wav::stereo_to_mono(a.samples.data(), mono.data(), 4);
		CHECK(mono[0] == 0.0f && mono[1] == (16384 / 32767.f + 0.0f) / 2.0f);
		zen::wav::encode_pcm16_mono(mono, 44100, dir + "/rt.wav");
		zen::wav::AudioData r;
		zen::wav::load(r, dir + "/rt.wav");
		CHECK(r.channelCount == 1 && r.samples.size() == 2 && r.samples[1] == (float)(int16_t)lroundf(mono[1] * 32767.f) / 32767.f);
	}
	{ // data chunk claims more bytes than the file holds: clipped, not over-read
		auto b = header(1, 1, 8000, 16, 1000);
		b.push_back(1); b.push_back(0); b.push_back(2); b.push_back(0);
		zen::wav::load(a, write_tmp(dir, "short_data.wav", b));
		CHECK(a.samples.size() == 2);
	}
	{ // file ends right after a chunk header / inside the fmt body
		auto b = header(1, 1, 8000, 16, 0);
		b.resize(12 + 8); // "fmt " + size, no body
		CHECK(throws([&] { zen::wav::load(a, write_tmp(dir, "trunc_fmt0.wav", b)); }));
		b = header(1, 1, 8000, 16, 0);
		b.resize(12 + 8 + 10);
		CHECK(throws([&] { zen::wav::load(a, write_tmp(dir, "trunc_fmt10.wav", b)); }));
		b = header(0xFFFE, 1, 8000, 16, 0, 40); // extensible header whose 40-byte body is cut after 16 bytes
		b.resize(12 + 8 + 16);
		CHECK(throws([&] { zen::wav::load(a, write_tmp(dir, "trunc_ext.wav", b)); })); // no data chunk follows
	}
	{ // bits per sample 1..7 used to divide by zero; 12 is not a whole number of bytes
		for (uint16_t bits : {(uint16_t)4, (uint16_t)7, (uint16_t)12}) {
			auto b = header(1, 1, 8000, bits, 4);
			b.resize(b.size() + 4);
			CHECK(throws([&] { zen::wav::load(a, write_tmp(dir, "bits.wav", b)); }));
		}
	}
	{ // zero channels, zero / negative sample rate
		auto b = header(1, 0, 8000, 16, 4);
		b.resize(b.size() + 4);
		CHECK(throws([&] { zen::wav::load(a, write_tmp(dir, "ch0.wav", b)); }));
		b = header(1, 1, 0, 16, 4);
		b.resize(b.size() + 4);
		CHECK(throws([&] { zen::wav::load(a, write_tmp(dir, "fs0.wav", b)); }));
		b = header(1, 1, 0x80000000u, 16, 4);
		b.resize(b.size() + 4);
		CHECK(throws([&] { zen::wav::load(a, write_tmp(dir, "fsneg.wav", b)); }));
	}
	{ // not RIFF, empty, unsupported encoding
		CHECK(throws([&] { zen::wav::load(a, write_tmp(dir, "empty.wav", {})); }));
		CHECK(throws([&] { zen::wav::load(a, write_tmp(dir, "junk.wav", std::vector<unsigned char>(64, 'x'))); }));
		auto b = header(2 /* ADPCM */, 1, 8000, 16, 4);
		b.resize(b.size() + 4);
		CHECK(throws([&] { zen::wav::load(a, write_tmp(dir, "adpcm.wav", b)); }));
		CHECK(throws([&] { zen::wav::load(a, dir + "/does_not_exist.wav"); }));
	}
	if (fails) {
		std::printf("%d checks failed\n", fails);
		return 1;
	}
	std::printf("wav checks passed\n");
	return 0;
}
