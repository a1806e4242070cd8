// test_rfft_host.cpp -- zen_amd/csrc/rfft_dev.h (the Hermitian half of the radix-2 DAG on a real frame) run on the CPU:
// the header's device functions compiled as ordinary inline functions (host clang++, scalar butterflies), the N/32
// threads of a frame executed one after the other, pass by pass, on an array that stands for the LDS image.  Compared BIT
// FOR BIT with the oracle's complex transform (oracle/zen_oracle.c zo_fft_c2c: fftw.h:51-129) of the same real frame, for
// every size the engine uses, zero-padded (the analysis frame, hps.cu:456-465) and not.  An exact zero may carry either sign.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime.h>
static inline void __builtin_amdgcn_fence(int, const char*) {}
static inline void __builtin_amdgcn_wave_barrier() {}
static inline void __syncthreads() {}
static inline double __builtin_amdgcn_rsq(double x) { return 1.0 / std::sqrt(x); }

#include "../../zen_amd/csrc/rfft_dev.h"

extern "C" {
#include "../../oracle/zen_oracle.h"
}

static int fails = 0;

struct TwScalar { // the table in memory, scalar butterflies (the packed ones are gfx950 instructions)
	static constexpr bool PLAIN = true;
	static constexpr bool PACKED = false;
	const float2* p;
	float2 get(int, int, int, int idx) const { return p[idx]; }
};
struct TwScalarPre : TwScalar { // every pass's twiddles requested at its top (RPass::TwP)
	static constexpr bool PRELOAD = true;
};
struct In {
	const float* x;
	float operator()(int idx) const { return x[idx]; }
};
struct Out {
	float2* X;
	int* hits;
	void operator()(int base, int off, float2 v, int slot)
	{
		const int bin = base + off;
		if (slot < 0 || slot > 16)
			++fails;
		X[bin] = v;
		++hits[bin];
	}
};

template <int LOG2N, int PASS, bool ZU, class TW>
static void run_passes(float2* lds, const TW& tw, In& in, Out& out)
{
	using RP = zfft::RPlan<LOG2N>;
	using PS = zfft::RPass<LOG2N, PASS, ZU, TW>;
	std::vector<typename PS::Regs> regs(RP::TF);
	for (int tf = 0; tf < RP::TF; ++tf)
		PS::load(tf, lds, in, regs[tf], tw.p);
	for (int tf = 0; tf < RP::TF; ++tf)
		PS::compute(tf, lds, tw, out, true, regs[tf]);
	if constexpr (PASS + 1 < RP::P)
		run_passes<LOG2N, PASS + 1, ZU, TW>(lds, tw, in, out);
}

static bool same(float a, float b) { return a == b || (std::isnan(a) && std::isnan(b)); } // (+0 == -0)

template <int LOG2N, bool ZU, class TW = TwScalar>
static void check(unsigned seed)
{
	using RP = zfft::RPlan<LOG2N>;
	const int N = RP::N;
	std::mt19937 rng(seed);
	std::uniform_real_distribution<float> u(-1.0f, 1.0f);
	std::vector<float> x(N, 0.0f);
	for (int i = 0; i < (ZU ? N / 2 : N); ++i)
		x[i] = (rng() % 7 == 0) ? 0.0f : u(rng) * (rng() % 5 == 0 ? 1e-3f : 1.0f);
	std::vector<float> tw(N);
	zo_twiddles(tw.data(), (size_t)N);
	std::vector<float> ref(2 * (size_t)N, 0.0f);
	for (int i = 0; i < N; ++i)
		ref[2 * i] = x[i];
	zo_fft_c2c(ref.data(), (size_t)N, 0);

	std::vector<float2> lds(RP::LDS_FLOAT2, make_float2(NAN, NAN));
	std::vector<float2> X(N / 2 + 1, make_float2(NAN, NAN));
	std::vector<int> hits(N / 2 + 1, 0);
	TW t;
	t.p = reinterpret_cast<const float2*>(tw.data());
	In in{x.data()};
	Out out{X.data(), hits.data()};
	run_passes<LOG2N, 0, ZU, TW>(lds.data(), t, in, out);
	int bad = 0;
	for (int k = 0; k <= N / 2; ++k) {
		if (hits[k] != 1 || !same(X[k].x, ref[2 * k]) || !same(X[k].y, ref[2 * k + 1])) {
			if (bad < 5)
				std::printf("FAIL n=%d zu=%d bin %d: hits %d got (%.9g, %.9g) want (%.9g, %.9g)\n", N, (int)ZU, k, hits[k], X[k].x, X[k].y,
				            ref[2 * k], ref[2 * k + 1]);
			++bad;
		}
		// and the upper half of the oracle's spectrum is the conjugate of the lower one, bit for bit (what the kernels rely on)
		if (k > 0 && k < N / 2 && (!same(ref[2 * (N - k)], ref[2 * k]) || !same(ref[2 * (N - k) + 1], -ref[2 * k + 1])))
			++bad;
	}
	if (bad) {
		std::printf("FAIL n=%d zu=%d: %d bins differ\n", N, (int)ZU, bad);
		++fails;
	}
}

template <int L>
static void check_size()
{
	for (unsigned s = 1; s <= (L >= 13 ? 2u : 6u); ++s) {
		check<L, true>(s * 977 + L);
		check<L, false>(s * 131 + L);
	}
	check<L, true, TwScalarPre>(4242 + L);
	if constexpr (L < 14)
		check_size<L + 1>();
}

int main()
{
	check_size<5>();
	if (fails) {
		std::printf("%d failures\n", fails);
		return 1;
	}
	std::printf("rfft host test: all sizes 32..16384 bit-identical to the oracle's transform\n");
	return 0;
}
