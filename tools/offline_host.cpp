// offline_host.cpp -- HPRIOffline<Backend::GPU>::process(std::vector<float>) timed exactly as the reference's CLI times it
// (zen/offline.h:141-147: t1; all_out = hpss.process(audio); t2): the by-value copy of the clip, the three result vectors
// and the host <-> device copies are all inside.  Beside it the floor the C++ signature sets on any implementation: copying
// one n-float vector and value-initialising three the plain way (page faults of fresh memory, one thread), measured the same way;
// and the same call from a caller that moves its clip in.
//   g++ -O2 -std=c++17 -I include -I zen_amd/libzen tools/offline_host.cpp -o /tmp/offline_host -L zen_amd -lzen -lzen_hip -Wl,-rpath,$PWD/zen_amd
//   /tmp/offline_host [seconds = 3600] [reps = 3]
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <libzen/hps.h>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
	const double seconds = argc > 1 ? atof(argv[1]) : 3600.0;
	const int reps = argc > 2 ? atoi(argv[2]) : 3;
	const float fs = 44100.0f;
	const std::size_t n = (std::size_t)(seconds * fs);
	std::vector<float> audio(n);
	unsigned lcg = 12345u;
	for (std::size_t i = 0; i < n; ++i) { // sines + a click every quarter second + a little noise
		lcg = lcg * 1664525u + 1013904223u;
		const float t = (float)(i % 441000) / fs;
		float v = 0.2f * (sinf(6.2831853f * 220.0f * t) + sinf(6.2831853f * 440.0f * t)) + 0.01f * ((float)(lcg >> 8) / 8388608.0f - 1.0f);
		if (i % 11025 < 200)
			v += 0.9f * expf(-(float)(i % 11025) / 44.0f) * ((float)(lcg >> 8) / 8388608.0f - 1.0f);
		audio[i] = v;
	}
	zen::hps::HPRIOffline<zen::Backend::GPU> hpss(fs, 4096, 256, 2.0f, 2.0f);
	std::array<std::vector<float>, 3> all_out;
	all_out = hpss.process(audio); // first call: staging buffers, the engines' buffers
	std::vector<double> ms, floor_ms, moved_ms;
	for (int r = 0; r < reps; ++r) {
		all_out = {};
		const double t1 = now_ms();
		all_out = hpss.process(audio);
		const double t2 = now_ms();
		ms.push_back(t2 - t1);
	}
	for (int r = 0; r < reps; ++r) { // a caller that does not need its clip afterwards (zen_amd/cli/main.cpp): no by-value copy
		all_out = {};
		std::vector<float> mine(audio);
		const double t1 = now_ms();
		all_out = hpss.process(std::move(mine));
		const double t2 = now_ms();
		moved_ms.push_back(t2 - t1);
	}
	double chk = 0;
	for (std::size_t i = 0; i < std::min<std::size_t>(n, 4096); ++i)
		chk += std::fabs(all_out[1][i]);
	all_out = {};
	for (int r = 0; r < reps; ++r) { // what the signature costs before any separation happens
		const double t1 = now_ms();
		{
			std::vector<float> by_value(audio);
			std::vector<float> a(n), b(n), c(n);
			asm volatile("" : : "r"(by_value.data()), "r"(a.data()), "r"(b.data()), "r"(c.data()) : "memory");
		}
		floor_ms.push_back(now_ms() - t1);
	}
	std::sort(ms.begin(), ms.end());
	std::sort(floor_ms.begin(), floor_ms.end());
	std::sort(moved_ms.begin(), moved_ms.end());
	printf("{\"api\": \"zen::hps::HPRIOffline<GPU>::process(std::vector<float>)\", \"seconds\": %.1f, \"samples\": %zu, \"reps\": %d, "
	       "\"ms_min\": %.3f, \"ms_median\": %.3f, \"x_realtime\": %.1f, \"moved_ms_min\": %.3f, \"moved_x_realtime\": %.1f, "
	       "\"plain_vectors_floor_ms\": %.3f, \"checksum\": %.6g}\n",
	       seconds, n, reps, ms.front(), ms[ms.size() / 2], seconds / (1e-3 * ms.front()), moved_ms.front(),
	       seconds / (1e-3 * moved_ms.front()), floor_ms.front(), chk);
	return 0;
}
