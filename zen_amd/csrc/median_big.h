// median_big.h -- register-resident exact sliding median for odd windows 65..267 (the frequency masks of
// hop 2048/4096 at 44.1 and 48 kHz: 85, 93, 171, 187 taps; 65 and 129 taps at 16/32 kHz).  Extends the
// scheme of median_net.h:
//
//   * the row is cut into aligned 16-sample blocks; every block is sorted ONCE (by the thread that owns
//     it) and published in LDS;
//   * a thread produces 16 consecutive outputs.  The samples common to its 16 windows are NB whole blocks
//     plus rl + rr <= 30 loose samples.  It merges the first 2^k sorted blocks pairwise (Batcher odd-even
//     merges, 16 -> 32 -> 64 -> 128) into one sorted list A, sorts the loose samples and merges them with
//     the remaining blocks into a second sorted list R, and gets the 16 candidate order statistics from
//     one more (heavily pruned) odd-even merge of the relevant window of A with R (255 taps, whose second
//     list is too long for that, uses the selection identity merge(A,R)[p] = min_q max(A[p-q], R[q-1]));
//   * the 15 + 15 flank samples then go through the same selection tree as for small windows.
//
// ~210 min/max per output at 171/187 taps, all with compile-time indices (no data-dependent addressing),
// instead of ~60 wave-wide instructions per output in the sliding wave-window kernel.
#pragma once
#include <type_traits>

#include "median_net.h"

namespace zbig {

constexpr int KEY_INF = 0x7fffffff;

template <int W>
struct Geo {
	static constexpr int m = W / 2;
	static constexpr int a = (m - 15) / 16;      // whole blocks left of the thread's own block
	static constexpr int rl = (m - 15) % 16;     // loose common samples on the left
	static constexpr int b = (m + 1) / 16;       // whole blocks from the own block on
	static constexpr int rr = (m + 1) % 16;      // loose common samples on the right
	static constexpr int NB = a + b;             // whole common blocks: t-a .. t+b-1
	static constexpr int BIG = NB >= 8 ? 8 : (NB >= 4 ? 4 : (NB >= 2 ? 2 : 1));
	static constexpr int REST = NB - BIG;        // remaining whole blocks (0 .. BIG-1)
	static constexpr int NA = BIG * 16;
	static constexpr int NX = rl + rr;           // loose samples
	static constexpr int NR = REST * 16 + NX;    // second sorted list
	static constexpr int RREAL = 32 + 16 * REST; // R slots in use: 32 for the loose samples + the blocks
	static constexpr int NRP = RREAL <= 32 ? 32 : (RREAL <= 64 ? 64 : (RREAL <= 128 ? 128 : 256));
	static constexpr bool supported = (W & 1) && W >= 65 && W <= 267; // (255..267: fifteen whole blocks, 0..6 loose samples a side)
};

template <int N, int TOTAL, int OFF = 0>
struct MergeLevel {
	template <int NA>
	static __device__ __forceinline__ void run(int (&x)[NA])
	{
		znet::oe_merge<N, OFF>(x);
		if constexpr (OFF + N < TOTAL)
			MergeLevel<N, TOTAL, OFF + N>::run(x);
	}
};

// merges of R: slots past REAL hold +inf only, a merge whose second half is all +inf is the identity
template <int N, int TOTAL, int REAL, int OFF>
struct MergeLevelR {
	template <int NA>
	static __device__ __forceinline__ void run(int (&x)[NA])
	{
		if constexpr (OFF + N / 2 < REAL)
			znet::oe_merge<N, OFF>(x);
		if constexpr (OFF + N < TOTAL)
			MergeLevelR<N, TOTAL, REAL, OFF + N>::run(x);
	}
};

// The thread reads through a loader: sorted(i, v) = sorted block t-a+i (i < NB), rawl(j, v) = raw block
// t-a-2+j, rawr(j, v) = raw block t+b+j (j = 0, 1).  out[g] = median of the W samples centred on sample
// 16t+g.  The three phases (loose samples, block merge + candidates, flanks + tree) are fenced so that the
// loads of one phase are not hoisted into the previous one: peak register use stays near 128.
__device__ __forceinline__ void phase_fence() { asm volatile("" ::: "memory"); }

// A loader may also offer sorted32(i, v): blocks t-a+i and t-a+i+1 merged (i even) -- the first level of the merge tree, which
// neighbouring threads share: thread t's pairs (t-a, t-a+1), (t-a+2, t-a+3), ... are thread t+2's shifted by one.  A kernel that
// merges every adjacent pair of sorted blocks ONCE (median_big.hip: an image of sorted 32-blocks next to the sorted 16-blocks)
// saves each thread BIG/2 - 1 of its BIG/2 merges of 32.
template <class T, class = void>
struct has_sorted32 : std::false_type {};
template <class T>
struct has_sorted32<T, std::void_t<decltype(&T::sorted32)>> : std::true_type {};

template <int W, class LD>
__device__ __forceinline__ void medians_big(const LD& ld, int (&out)[16])
{
	using G = Geo<W>;
	static_assert(G::supported, "window not covered by the block-merge kernel");
	// ---- A: the 2^k whole blocks merged into one sorted list; only ranks ALO..AHI can become candidates
	constexpr int ALO = (G::m - 15 - G::NR) < 0 ? 0 : (G::m - 15 - G::NR);
	constexpr int AHI = G::m > G::NA - 1 ? G::NA - 1 : G::m;
	int An[AHI - ALO + 1];
	{
		int A[G::NA];
		if constexpr (has_sorted32<LD>::value && G::BIG >= 2) {
#pragma unroll
			for (int i = 0; i < G::BIG; i += 2) {
				int B[32];
				ld.sorted32(i, B);
#pragma unroll
				for (int j = 0; j < 32; ++j)
					A[16 * i + j] = B[j];
			}
		}
		else {
#pragma unroll
			for (int i = 0; i < G::BIG; ++i) {
				int B[16];
				ld.sorted(i, B);
#pragma unroll
				for (int j = 0; j < 16; ++j)
					A[16 * i + j] = B[j];
			}
			if constexpr (G::NA >= 32)
				MergeLevel<32, G::NA>::run(A);
		}
		if constexpr (G::NA >= 64)
			MergeLevel<64, G::NA>::run(A);
		if constexpr (G::NA >= 128)
			MergeLevel<128, G::NA>::run(A);
#pragma unroll
		for (int i = ALO; i <= AHI; ++i)
			An[i - ALO] = A[i];
	}
	phase_fence();
	// ---- R: loose samples and the remaining blocks, sorted, padded with +inf
	constexpr int NRP = G::NRP;
	int R[NRP];
	{
#pragma unroll
		for (int i = 0; i < NRP; ++i)
			R[i] = KEY_INF;
		{
			int X[32];
			{
				int L[16], Rr[16];
				ld.rawl(1, L);
				ld.rawr(0, Rr);
#pragma unroll
				for (int i = 0; i < 32; ++i)
					X[i] = KEY_INF;
#pragma unroll
				for (int i = 0; i < G::rl; ++i)
					X[i] = L[16 - G::rl + i];
#pragma unroll
				for (int i = 0; i < G::rr; ++i)
					X[G::rl + i] = Rr[i];
			}
			znet::sort_net<32>(X);
#pragma unroll
			for (int i = 0; i < 32; ++i)
				R[i] = X[i];
		}
#pragma unroll
		for (int k = 0; k < G::REST; ++k) {
			int B[16];
			ld.sorted(G::BIG + k, B);
#pragma unroll
			for (int i = 0; i < 16; ++i)
				R[32 + 16 * k + i] = B[i];
		}
		if constexpr (NRP >= 64) {
			MergeLevelR<32, NRP, G::RREAL, 32>::run(R); // pairs of blocks (slot 0 is the sorted loose samples)
			MergeLevelR<64, NRP, G::RREAL, 0>::run(R);
		}
		if constexpr (NRP >= 128)
			MergeLevelR<128, NRP, G::RREAL, 0>::run(R);
		if constexpr (NRP >= 256)
			MergeLevelR<256, NRP, G::RREAL, 0>::run(R);
	}
	// ---- candidates = ranks m-15 .. m of A u R.  At most NR samples of R precede any of them, so only the
	// window An = A[ALO..AHI] of A matters, and the ranks are m-15-ALO .. m-ALO of An u R.
	int cand[16];
	constexpr int NAN_ = AHI - ALO + 1;
	constexpr int HALF = (NAN_ <= 64 && NRP <= 64) ? 64 : 128;
	if constexpr (NAN_ <= HALF && NRP <= HALF) {
		// one odd-even merge of the two lists, each padded with +inf to HALF; the compiler prunes the
		// comparators that cannot reach the 16 outputs read below (~235 comparators at HALF = 64 instead of
		// 16 * NR max/min terms)
		int X[2 * HALF];
#pragma unroll
		for (int j = 0; j < HALF; ++j) {
			X[j] = j < NAN_ ? An[j < NAN_ ? j : 0] : KEY_INF;
			X[HALF + j] = j < NRP ? R[j < NRP ? j : 0] : KEY_INF;
		}
		znet::oe_merge<2 * HALF, 0>(X);
#pragma unroll
		for (int i = 0; i < 16; ++i)
			cand[i] = X[G::m - 15 - ALO + i];
	}
	else {
		// selection identity merge(A,R)[p] = min_q max(A[p-q], R[q-1])
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int p = G::m - 15 + i;
			int best = KEY_INF;
#pragma unroll
			for (int q = 0; q <= G::NR; ++q) { // q samples from R, p + 1 - q from A
				const int ia = p - q;          // last sample taken from A
				if (ia > G::NA - 1 || ia < -1)
					continue;                  // A or R cannot supply that many
				int term;
				if (ia < 0)
					term = R[q - 1];
				else if (q == 0)
					term = An[ia - ALO];
				else
					term = max(An[ia - ALO], R[q - 1]);
				best = min(best, term);
			}
			cand[i] = best;
		}
	}
	phase_fence();
	// ---- flanks and the selection tree
	constexpr int NE = W + 15;
	int e[NE];
	{
		int L32[32], R32[32];
		ld.rawl(0, L32);
		ld.rawl(1, L32 + 16);
		ld.rawr(0, R32);
		ld.rawr(1, R32 + 16);
#pragma unroll
		for (int q = 0; q < NE; ++q)
			e[q] = 0;
#pragma unroll
		for (int q = 0; q < 15; ++q) {
			e[q] = L32[32 - (G::rl + 15) + q];
			e[W + q] = R32[G::rr + q];
		}
	}
	znet::Node<W, 16, 0, NE, 16>::run(e, cand, out);
}

} // namespace zbig
