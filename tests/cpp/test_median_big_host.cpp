// test_median_big_host.cpp -- the block-merge selection of zen_amd/csrc/median_big.h (the long frequency masks: 65 .. 257 taps)
// run on the CPU (plain g++, no GPU), as tests/cpp/test_median_net_host.cpp does for the short masks.  For every mask length
// the kernels instantiate: zbig::medians_big<W> through a loader that hands out sorted 16-blocks (rounds 2-5, and what
// rt_wide.hip's loader still does) AND through one that also hands out sorted PAIRS of blocks (round 6: the first level of
// the merge tree shared between neighbouring threads, median_big.hip srt2) -- both against a brute-force median with the
// semantics of MedianFilterCPU (libzen/mfilt.h:270-342: centred mask; the line here is long enough that no tap leaves it).
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

#include "../../zen_amd/csrc/median_big.h"

static int fails = 0;
#define CHECK(c)                                                     \
	do {                                                             \
		if (!(c)) {                                                  \
			std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); \
			++fails;                                                 \
		}                                                            \
	} while (0)

// the kernel's three images for a line of `nb` blocks: raw blocks, sorted blocks, sorted pairs (s, s + 1)
struct Images {
	std::vector<int> raw, srt, srt2;
	explicit Images(const std::vector<int>& x)
	    : raw(x)
	    , srt(x)
	{
		const size_t nb = x.size() / 16;
		for (size_t b = 0; b < nb; ++b)
			std::sort(srt.begin() + 16 * b, srt.begin() + 16 * (b + 1));
		srt2.resize(32 * (nb - 1));
		for (size_t b = 0; b + 1 < nb; ++b)
			std::merge(srt.begin() + 16 * b, srt.begin() + 16 * (b + 1), srt.begin() + 16 * (b + 1), srt.begin() + 16 * (b + 2),
			           srt2.begin() + 32 * b);
	}
};

template <int W>
struct Loader { // thread t: sorted(i) = sorted block t-a+i, rawl(j) = raw block t-a-2+j, rawr(j) = raw block t+b+j (median_big.h)
	using G = zbig::Geo<W>;
	const Images& im;
	int t;
	void sorted(int i, int* v) const { std::copy_n(im.srt.begin() + 16 * (t - G::a + i), 16, v); }
	void rawl(int j, int* v) const { std::copy_n(im.raw.begin() + 16 * (t - G::a - 2 + j), 16, v); }
	void rawr(int j, int* v) const { std::copy_n(im.raw.begin() + 16 * (t + G::b + j), 16, v); }
};
template <int W>
struct Loader32 : Loader<W> {
	using G = zbig::Geo<W>;
	void sorted32(int i, int* v) const { std::copy_n(this->im.srt2.begin() + 32 * (this->t - G::a + i), 32, v); }
};

template <int W>
static void test_w(std::mt19937& rng)
{
	using G = zbig::Geo<W>;
	static_assert(zbig::has_sorted32<Loader32<W>>::value && !zbig::has_sorted32<Loader<W>>::value, "loader traits");
	const int nb = 64; // blocks in the line; threads far enough from both ends
	for (int rep = 0; rep < 6; ++rep) {
		std::vector<int> x(16 * nb);
		for (auto& v : x)
			v = rep & 1 ? (int)(rng() % 7) : (int)(rng() >> 1); // ties, and distinct keys
		const Images im(x);
		for (int t = G::a + 2; t < nb - G::b - 2; t += 5) {
			int plain[16], shared[16];
			Loader<W> l1{im, t};
			zbig::medians_big<W>(l1, plain);
			Loader32<W> l3{{im, t}};
			zbig::medians_big<W>(l3, shared);
			for (int g = 0; g < 16; ++g) {
				const int c = 16 * t + g;
				std::vector<int> w(x.begin() + c - W / 2, x.begin() + c + W / 2 + 1);
				std::nth_element(w.begin(), w.begin() + W / 2, w.end());
				CHECK(plain[g] == w[W / 2]);
				CHECK(shared[g] == w[W / 2]);
			}
		}
	}
	std::printf("W = %d: BIG %d REST %d loose %d ok\n", W, G::BIG, G::REST, G::NX);
}

int main()
{
	std::mt19937 rng(12345);
	test_w<65>(rng);
	test_w<85>(rng);
	test_w<93>(rng);
	test_w<129>(rng);
	test_w<171>(rng);
	test_w<187>(rng);
	test_w<255>(rng);
	test_w<257>(rng);
	std::printf("%d failures\n", fails);
	return fails ? 1 : 0;
}
