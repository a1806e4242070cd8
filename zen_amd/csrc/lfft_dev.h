// lfft_dev.h -- fft_dev.h's transform laid out for LATENCY: one frame, one workgroup, few values per thread.
//
// The single-hop kernels transform ONE frame per call.  With fft_dev.h's plan (16 values per thread: the layout that
// gives the block kernels their throughput) a 2048-point frame is the work of 128 threads -- two wavefronts on two
// of a CU's four SIMDs, each issuing one instruction every four cycles with nobody to overlap its LDS and barrier
// waits -- and the call is bound by that instruction stream (rt_sse.hip, hop 512: 3.6 us forward, 4.5 us inverse,
// profiles/r05_rt_latency.jsonl).  Here a thread holds V = 2^LOG2V values (4, 8 or 16), the frame is spread over N / V
// threads on all four SIMDs, and a pass runs ceil-balanced LOG2V stages of the same radix-2 DIT DAG: the butterflies
// are fft_dev.h's own butterfly(), in the same order per value, so every bin is the oracle's bit for bit (the DAG
// does not depend on how its stages are grouped into passes).
//
// The image in LDS is double-buffered: pass p writes image p & 1 and pass p + 1 reads it after ONE barrier (with a
// single image a pass needs two: everybody has read before anybody writes).
//
// Twiddles: every one a thread needs in any pass, in registers (LTwRegs, as fft_dev.h's TwRegs); a resident kernel fills
// them once per launch.
#pragma once
#include "fft_dev.h"

#pragma clang fp contract(off)

namespace zfft {

template <int LOG2N, int LOG2V>
struct LPlan {
	static_assert(LOG2V >= 2 && LOG2V <= 4 && LOG2N >= 2 * LOG2V, "4, 8 or 16 values per thread");
	static constexpr int N = 1 << LOG2N, V = 1 << LOG2V;
	static constexpr int P = (LOG2N + LOG2V - 1) / LOG2V; // passes
	static constexpr int BASE = LOG2N / P, REM = LOG2N % P;
	static constexpr int r(int p) { return BASE + (p < REM ? 1 : 0); } // stages in pass p
	static constexpr int s(int p) // stages completed before pass p
	{
		int a = 0;
		for (int i = 0; i < p; ++i)
			a += r(i);
		return a;
	}
	static constexpr int RMAX = BASE + (REM ? 1 : 0);
	static constexpr int TF = N / V; // threads of the frame
#ifndef ZEN_LFFT_PAD_V8
#define ZEN_LFFT_PAD_V8 4
#endif
	static constexpr int PAD_SHIFT = LOG2V == 3 ? ZEN_LFFT_PAD_V8 : 4; // (one 8-byte slot of padding per 2^PAD_SHIFT values, fft_dev.h)
	static constexpr int IMG = N + (N >> PAD_SHIFT);   // one padded image
	static constexpr int LDS_FLOAT2 = 2 * IMG;         // the two of them
	static __device__ __forceinline__ int pad(int i) { return i + (i >> PAD_SHIFT); }
};

template <int LOG2N, int LOG2V>
struct LTwRegs {
	static constexpr bool PLAIN = true;
	static constexpr bool PACKED = false; // (one wavefront per SIMD: scalar butterflies, see fft_dev.h)
	using PL = LPlan<LOG2N, LOG2V>;
	static constexpr int NBMAX = PL::V >> PL::BASE;          // groups per thread in the pass with the fewest stages
	static constexpr int SLOTS = 1 << (PL::RMAX - 1);        // stage q of a pass loads max(1, 2^(q-2)) twiddles per group
	float2 w[PL::P][NBMAX][SLOTS];
	__device__ __forceinline__ float2 get(int pass, int i, int slot, int) const { return w[pass][i][slot]; }
	template <int PASS = 0>
	__device__ __forceinline__ void fill(int tf, const float2* __restrict__ p)
	{
		constexpr int rr = PL::r(PASS), R = 1 << rr, NB = PL::V / R, sL = PL::s(PASS), log2J = LOG2N - sL - rr;
		if constexpr (PASS > 0) { // (the first pass has k = 0: butterfly()'s TRIV shortcuts, or table entries it asks for itself)
#pragma unroll
			for (int i = 0; i < NB; ++i) {
				const int k = (tf + i * PL::TF) >> log2J;
#pragma unroll
				for (int q = 1; q <= rr; ++q) {
					const int nload = q == 1 ? 1 : (1 << (q - 2));
#pragma unroll
					for (int c = 0; c < nload; ++c)
						w[PASS][i][(q == 1 ? 0 : (1 << (q - 2))) + c] = p[(k << (LOG2N - sL - q)) + (c << (LOG2N - q))];
				}
			}
		}
		else {
			// k = 0: stage q asks for entries c << (LOG2N - q), 1 <= c < 2^(q-2) (c = 0 is 1, the upper half is derived)
#pragma unroll
			for (int i = 0; i < NB; ++i)
#pragma unroll
				for (int q = 3; q <= rr; ++q)
#pragma unroll
					for (int c = 1; c < (1 << (q - 2)); ++c)
						w[PASS][i][(1 << (q - 2)) + c] = p[c << (LOG2N - q)];
		}
		if constexpr (PASS + 1 < PL::P)
			fill<PASS + 1>(tf, p);
	}
};

// One pass of the frame's transform for thread tf < TF: load() brings the thread's V values into registers (first pass:
// in(idx, slot), idx = tf + slot * TF; later: the image the pass before wrote), compute() runs the pass's stages and writes
// the other image -- or, in the last pass, hands out(idx, X, lower, slot), idx = tf + slot * TF, lower = idx < N/2.
// ZU (first pass): the values at idx >= N/2 are zeros and are not asked for.  HALF_OUT: only idx < N/2 is handed out.
template <int LOG2N, int LOG2V, int PASS, bool INV, bool ZU, bool HALF_OUT, class TW>
struct LPass {
	using PL = LPlan<LOG2N, LOG2V>;
	static constexpr int N = PL::N, TF = PL::TF;
	static constexpr int rr = PL::r(PASS), R = 1 << rr, NB = PL::V / R;
	static constexpr int sL = PL::s(PASS);
	static constexpr int log2J = LOG2N - sL - rr, J = 1 << log2J;
	static constexpr bool FIRST = PASS == 0, LAST = PASS == PL::P - 1;
	static constexpr bool ZUP = ZU && FIRST;
	struct Regs {
		float2 v[NB][R];
	};
	template <class In>
	static __device__ __forceinline__ void load(int tf, const float2* __restrict__ lds, In& in, Regs& g)
	{
		const float2* img = lds + ((PASS + 1) & 1) * PL::IMG; // what pass PASS - 1 wrote
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			const int b = tf + i * TF;
			const int k = b >> log2J, j = b & (J - 1);
#pragma unroll
			for (int m = 0; m < R; ++m) {
				if (ZUP && m >= R / 2)
					g.v[i][m] = make_float2(0.f, 0.f);
				else if (FIRST)
					g.v[i][m] = in(m * J + j, /*slot=*/m * NB + i);
				else
					g.v[i][m] = img[PL::pad((k * R + m) * J + j)];
			}
		}
	}
	template <class Out>
	static __device__ __forceinline__ void compute(int tf, float2* __restrict__ lds, const TW& tw, Out& out, Regs& g)
	{
		float2* img = lds + (PASS & 1) * PL::IMG;
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			const int b = tf + i * TF;
			const int k = b >> log2J;
			butterfly<R, INV, ZUP, (FIRST && TW::PLAIN)>(g.v[i], k, sL, LOG2N, tw, PASS, i);
#pragma unroll
			for (int c = 0; c < R; ++c) {
				const int idx = b + c * (N / R);
				if (LAST) {
					if (!HALF_OUT || c < R / 2)
						out(idx, g.v[i][c], /*lower_half=*/c < R / 2, /*slot=*/c * NB + i);
				}
				else
					img[PL::pad(idx)] = g.v[i][c];
			}
		}
	}
};

template <int LOG2N, int LOG2V, int PASS, bool INV, bool ZU, bool HALF_OUT, class TW, class In, class Out>
__device__ __forceinline__ void lfft_passes(int tf, float2* __restrict__ lds, const TW& tw, In& in, Out& out)
{
	using PS = LPass<LOG2N, LOG2V, PASS, INV, ZU, HALF_OUT, TW>;
	typename PS::Regs g;
	PS::load(tf, lds, in, g);
	PS::compute(tf, lds, tw, out, g);
	if constexpr (!PS::LAST) {
		// the pass's image is complete before the next pass reads it; the image that pass writes was last read before this barrier
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		lfft_passes<LOG2N, LOG2V, PASS + 1, INV, ZU, HALF_OUT, TW, In, Out>(tf, lds, tw, in, out);
	}
}

// The whole transform; every one of the TF threads of the workgroup calls it (it contains barriers).  The first pass reads no
// LDS and writes image 0, the last writes none: a caller may keep data of its own in image 1 until the second pass has been
// reached (and in image 0 from the last barrier of one transform to the first write of the next).
template <int LOG2N, int LOG2V, bool INV, bool ZU, bool HALF_OUT, class TW, class In, class Out>
__device__ __forceinline__ void lfft_frame(int tf, float2* __restrict__ lds, const TW& tw, In& in, Out& out)
{
	lfft_passes<LOG2N, LOG2V, 0, INV, ZU, HALF_OUT, TW, In, Out>(tf, lds, tw, in, out);
}

} // namespace zfft
