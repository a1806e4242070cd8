// test_median_net_host.cpp -- the sorting / selection networks of zen_amd/csrc/median_net.h run on the CPU
// (plain g++, no GPU): the header's device functions are compiled as ordinary inline functions behind a few
// shims.  Checks, against a brute-force median (the semantics of MedianFilterCPU, libzen/mfilt.h:270-342,
// replicate border):
//   * znet::medians<W, T>          for every odd W <= 63 the kernels instantiate
//   * znet::pyramid16              (sorted block + sorted halves + sorted quarters)
//   * znet::mid16_of_two_sorted16
//   * znet::medians47_shared       with its Shared47 filled from the NEIGHBOURS' pyramids, exactly as
//                                  median47_dpp_kernel does through DPP shifts (median47.hip)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define ZNET_HOST_TEST 1
#define __device__
#define __forceinline__ inline
using std::max;
using std::min;
static inline int __float_as_int(float f)
{
	int i;
	std::memcpy(&i, &f, 4);
	return i;
}
static inline float __int_as_float(int i)
{
	float f;
	std::memcpy(&f, &i, 4);
	return f;
}
struct int4 {
	int x, y, z, w;
};
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }

#include "../../zen_amd/csrc/median_net.h"

static int fails = 0;
#define CHECK(c)                                                     \
	do {                                                             \
		if (!(c)) {                                                  \
			std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); \
			++fails;                                                 \
		}                                                            \
	} while (0)

static int brute(const std::vector<int>& x, int centre, int W)
{
	std::vector<int> w;
	const int n = (int)x.size(), mid = W / 2;
	for (int d = -mid; d <= mid; ++d)
		w.push_back(x[std::min(std::max(centre + d, 0), n - 1)]); // ippBorderRepl
	std::sort(w.begin(), w.end());
	return w[mid];
}

template <int W>
static void test_medians(std::mt19937& rng, int few_values)
{
	constexpr int T = znet::outputs_per_thread(W), NE = W + T - 1;
	for (int rep = 0; rep < 20; ++rep) {
		int e[NE], out[T];
		std::vector<int> x(NE);
		for (int i = 0; i < NE; ++i)
			x[i] = e[i] = few_values ? (int)(rng() % 5) : (int)(rng() >> 1);
		znet::medians<W, T, NE>(e, out);
		for (int g = 0; g < T; ++g) {
			std::vector<int> w(x.begin() + g, x.begin() + g + W);
			std::sort(w.begin(), w.end());
			CHECK(out[g] == w[W / 2]);
		}
	}
}

template <int W>
struct AllW {
	static void run(std::mt19937& rng)
	{
		test_medians<W>(rng, 0);
		test_medians<W>(rng, 1);
		if constexpr (W + 2 <= 63)
			AllW<W + 2>::run(rng);
	}
};

// a whole row through the 47-tap block scheme, emulating the lanes of median47_dpp_kernel
static void test_row47(std::mt19937& rng, int n_blocks, int few_values)
{
	const int n = 16 * n_blocks; // row length; thread t produces outputs 16t .. 16t+15
	std::vector<int> x(n);
	for (auto& v : x)
		v = few_values ? (int)(rng() % 7) : (int)(rng() >> 1);
	auto at = [&](int c) { return x[std::min(std::max(c, 0), n - 1)]; };
	struct Pyr {
		int raw[16], s16[16], oct[16], quad[16];
	};
	std::vector<Pyr> pyr(n_blocks + 3); // blocks u = -1 .. n_blocks + 1  ->  index u + 1
	for (int u = -1; u <= n_blocks + 1; ++u) {
		Pyr& p = pyr[u + 1];
		for (int i = 0; i < 16; ++i)
			p.raw[i] = at(16 * u - 8 + i);
		znet::pyramid16(p.raw, p.s16, p.oct, p.quad);
		int ref[16];
		std::memcpy(ref, p.raw, sizeof(ref));
		std::sort(ref, ref + 16);
		CHECK(std::memcmp(ref, p.s16, sizeof(ref)) == 0);
		for (int h = 0; h < 2; ++h) {
			std::memcpy(ref, p.raw + 8 * h, 32);
			std::sort(ref, ref + 8);
			CHECK(std::memcmp(ref, p.oct + 8 * h, 32) == 0);
		}
		for (int q = 0; q < 4; ++q) {
			std::memcpy(ref, p.raw + 4 * q, 16);
			std::sort(ref, ref + 4);
			CHECK(std::memcmp(ref, p.quad + 4 * q, 16) == 0);
		}
	}
	for (int t = 0; t < n_blocks; ++t) {
		const Pyr &own = pyr[t + 1], &nx = pyr[t + 2], &nx2 = pyr[t + 3], &pv = pyr[t];
		znet::Shared47 sh;
		znet::mid16_of_two_sorted16(own.s16, nx.s16, sh.cand);
		{
			int both[32];
			std::memcpy(both, own.s16, 64);
			std::memcpy(both + 16, nx.s16, 64);
			std::sort(both, both + 32);
			CHECK(std::memcmp(both + 8, sh.cand, 64) == 0);
		}
		for (int i = 0; i < 8; ++i) {
			sh.lo_oct[i] = pv.oct[8 + i];
			sh.hi_oct[i] = nx2.oct[i];
		}
		for (int i = 0; i < 4; ++i) {
			sh.lo_q[0][i] = pv.quad[4 + i];
			sh.lo_q[1][i] = pv.quad[12 + i];
			sh.hi_q[0][i] = nx2.quad[i];
			sh.hi_q[1][i] = nx2.quad[8 + i];
		}
		std::memcpy(sh.lo_raw, pv.raw, 64);
		std::memcpy(sh.hi_raw, nx2.raw, 64);
		int out[16];
		znet::medians47_shared(sh, out);
		for (int g = 0; g < 16; ++g)
			CHECK(out[g] == brute(x, 16 * t + g, 47));
	}
	// the border block B(n_blocks) = last eight samples + eight copies of the last one (thread 255's splice)
	{
		int w[8];
		for (int i = 0; i < 8; ++i)
			w[i] = x[n - 8 + i];
		const int c = w[7];
		std::sort(w, w + 8);
		int s[16];
		for (int i = 0; i < 8; ++i) {
			s[i] = std::min(w[i], c);
			s[8 + i] = std::max(w[i], c);
		}
		CHECK(std::memcmp(s, pyr[n_blocks + 1].s16, 64) == 0);
	}
}

int main()
{
	std::mt19937 rng(12345);
	AllW<3>::run(rng);
	for (int rep = 0; rep < 6; ++rep) {
		test_row47(rng, 8, rep & 1);
		test_row47(rng, 256, rep & 1);
	}
	test_row47(rng, 3, 0);
	if (fails) {
		std::printf("%d checks failed\n", fails);
		return 1;
	}
	std::printf("median_net host checks passed\n");
	return 0;
}
