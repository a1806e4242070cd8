// test_lfft_host.cpp -- zen_amd/csrc/lfft_dev.h (the transform of the single-hop kernels: 4, 8 or 16 values per thread, two
// LDS images) run on the CPU: the header's device functions compiled as ordinary inline functions (host clang++), the
// threads of a frame executed pass by pass on an array that stands for LDS -- once in lock step (all loads of a pass, then
// all its stores) and once thread after thread (a thread's stores before the next thread's loads: legal between two
// barriers, and only correct because a pass writes the image it does not read).  Compared BIT FOR BIT with the oracle's
// transform (oracle/zen_oracle.c zo_fft_c2c: fftw.h:51-129), forward (zero-padded frame, hps.cu:456-465, and full) and
// inverse (all outputs and the first half only, hps.cu:498-530).  An exact zero may carry either sign (butterfly()'s shortcuts).
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

#include "../../zen_amd/csrc/lfft_dev.h"

extern "C" {
#include "../../oracle/zen_oracle.h"
}

static int fails = 0;

struct In {
	const float2* x;
	float2 operator()(int idx, int) const { return x[idx]; }
};
template <int TF, int N>
struct Out {
	float2* X;
	int* hits;
	int tf;
	void operator()(int idx, float2 v, bool lower, int slot)
	{
		if (idx != tf + slot * TF || lower != (idx < N / 2)) // (slot numbering: idx = tf + slot * TF)
			++fails;
		X[idx] = v;
		++hits[idx];
	}
};

template <int LOG2N, int LOG2V, int PASS, bool INV, bool ZU, bool HALF, class TW>
static void run_passes(float2* lds, std::vector<TW>& tw, In& in, float2* X, int* hits, bool lockstep)
{
	using PL = zfft::LPlan<LOG2N, LOG2V>;
	using PS = zfft::LPass<LOG2N, LOG2V, PASS, INV, ZU, HALF, TW>;
	std::vector<typename PS::Regs> regs(PL::TF);
	if (lockstep) {
		for (int tf = 0; tf < PL::TF; ++tf)
			PS::load(tf, lds, in, regs[tf]);
		for (int tf = 0; tf < PL::TF; ++tf) {
			Out<PL::TF, PL::N> out{X, hits, tf};
			PS::compute(tf, lds, tw[tf], out, regs[tf]);
		}
	}
	else {
		for (int tf = PL::TF - 1; tf >= 0; --tf) {
			Out<PL::TF, PL::N> out{X, hits, tf};
			PS::load(tf, lds, in, regs[tf]);
			PS::compute(tf, lds, tw[tf], out, regs[tf]);
		}
	}
	if constexpr (PASS + 1 < PL::P)
		run_passes<LOG2N, LOG2V, PASS + 1, INV, ZU, HALF, TW>(lds, tw, in, X, hits, lockstep);
}

static bool same(float a, float b) { return a == b || (std::isnan(a) && std::isnan(b)); } // (+0 == -0)

template <int LOG2N, int LOG2V, bool INV, bool ZU, bool HALF>
static void check(unsigned seed, bool lockstep)
{
	using PL = zfft::LPlan<LOG2N, LOG2V>;
	using TW = zfft::LTwRegs<LOG2N, LOG2V>;
	const int N = PL::N;
	std::mt19937 rng(seed);
	std::uniform_real_distribution<float> u(-1.0f, 1.0f);
	std::vector<float2> x(N, make_float2(0.f, 0.f));
	for (int i = 0; i < (ZU ? N / 2 : N); ++i) {
		x[i].x = (rng() % 7 == 0) ? 0.0f : u(rng) * (rng() % 5 == 0 ? 1e-3f : 1.0f);
		x[i].y = (!INV && ZU) ? 0.0f : ((rng() % 9 == 0) ? 0.0f : u(rng));
	}
	std::vector<float> table(N);
	zo_twiddles(table.data(), (size_t)N);
	std::vector<float> ref(2 * (size_t)N);
	for (int i = 0; i < N; ++i) {
		ref[2 * i] = x[i].x;
		ref[2 * i + 1] = x[i].y;
	}
	zo_fft_c2c(ref.data(), (size_t)N, INV ? 1 : 0);

	std::vector<float2> lds(PL::LDS_FLOAT2, make_float2(NAN, NAN));
	std::vector<float2> X(N, make_float2(NAN, NAN));
	std::vector<int> hits(N, 0);
	std::vector<TW> tw(PL::TF);
	for (int tf = 0; tf < PL::TF; ++tf) {
		std::memset((void*)&tw[tf], 0xff, sizeof(TW)); // (a twiddle that is used without having been loaded is a NaN)
		tw[tf].fill(tf, reinterpret_cast<const float2*>(table.data()));
	}
	In in{x.data()};
	run_passes<LOG2N, LOG2V, 0, INV, ZU, HALF, TW>(lds.data(), tw, in, X.data(), hits.data(), lockstep);
	int bad = 0;
	for (int k = 0; k < N; ++k) {
		const int want_hits = (HALF && k >= N / 2) ? 0 : 1;
		if (hits[k] != want_hits || (want_hits && (!same(X[k].x, ref[2 * k]) || !same(X[k].y, ref[2 * k + 1])))) {
			if (bad < 5)
				std::printf("FAIL n=%d v=%d inv=%d zu=%d half=%d bin %d: hits %d got (%.9g, %.9g) want (%.9g, %.9g)\n", N, PL::V, (int)INV,
				            (int)ZU, (int)HALF, k, hits[k], X[k].x, X[k].y, ref[2 * k], ref[2 * k + 1]);
			++bad;
		}
	}
	if (bad) {
		std::printf("FAIL n=%d v=%d: %d bins differ\n", N, PL::V, bad);
		++fails;
	}
}

template <int L, int LV>
static void check_plan()
{
	for (unsigned s = 1; s <= 2; ++s) {
		check<L, LV, false, true, false>(s * 977 + L, s == 1);  // analysis: zero-padded frame, all bins
		check<L, LV, false, false, false>(s * 131 + L, s == 2); // a full frame
		check<L, LV, true, false, true>(s * 31 + L, s == 1);    // synthesis: the first half of the samples
		check<L, LV, true, false, false>(s * 57 + L, s == 2);
	}
}

template <int L>
static void check_size()
{
	check_plan<L, 2>();
	check_plan<L, 3>();
	if constexpr (L >= 8)
		check_plan<L, 4>();
	if constexpr (L < 13)
		check_size<L + 1>();
}

int main()
{
	check_size<6>();
	if (fails) {
		std::printf("%d failures\n", fails);
		return 1;
	}
	std::printf("lfft host test: sizes 64..8192 with 4, 8 and 16 values per thread bit-identical to the oracle's transform\n");
	return 0;
}
