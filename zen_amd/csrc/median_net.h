// median_net.h -- register-resident exact sliding median for odd windows W <= 63, shared by the
// frequency- and time-direction kernels (median_net.hip).
//
// One thread produces T consecutive outputs from the W+T-1 inputs e[0..W+T-2] they depend on
// (output g is the median of e[g..g+W-1]).  Work is shared between the T overlapping windows:
//
//   core(node)  = the samples common to all windows of a node (a group of Tn consecutive windows)
//   cand(node)  = the Tn order statistics of core(node) with ranks mid-Tn+1..mid, sorted.  The median of
//                 every window of the node is one of them or one of its own <= Tn-1 extra samples.
//   root        : core = e[T-1..W-1] (W-T+1 samples), sorted once with a Batcher merge-exchange
//                 network; the compiler prunes comparators whose outputs are never used.
//   split       : a child (half the windows) has core(child) = core(node) + a chunk of Tn/2 extra
//                 samples; cand(child) = positions Tn/2..Tn-1 of merge(cand(node), sort(chunk)), obtained
//                 directly with the two-list selection identity
//                     merge(A,B)[p] = min_i max(A[p-i], B[i-1])          (v_max / v_min3 chains)
//   leaf        : Tn = 1, cand has one element: the median (the last split is a v_med3).
//
// Needs T <= mid+1 so that every candidate rank lies inside the core.  For W = 47, T = 16 this is ~55
// integer min/max per output instead of ~250 for a per-window selection network.  Samples are ordered
// through the monotone float->int key, so the output is the bit pattern of an input sample.
#pragma once
#ifndef ZNET_HOST_TEST // tests/cpp/test_median_net_host.cpp runs the networks on the CPU with its own shims
#include <hip/hip_runtime.h>
#endif

namespace znet {

__device__ __forceinline__ int f2key(float f)
{
	int b = __float_as_int(f);
	return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

// NONNEG: every sample is >= +0 (spectrogram magnitudes), whose bit patterns already order like the values
template <bool NONNEG>
__device__ __forceinline__ int to_key(float f)
{
	return NONNEG ? __float_as_int(f) : f2key(f);
}
template <bool NONNEG>
__device__ __forceinline__ float from_key(int k)
{
	return NONNEG ? __int_as_float(k) : key2f(k);
}

// N keys (a multiple of 4) between registers and 16-byte aligned LDS
template <int N>
__device__ __forceinline__ void lds_load(const int* p, int* v)
{
#pragma unroll
	for (int i = 0; i < N / 4; ++i) {
		const int4 q = *reinterpret_cast<const int4*>(p + 4 * i);
		v[4 * i] = q.x;
		v[4 * i + 1] = q.y;
		v[4 * i + 2] = q.z;
		v[4 * i + 3] = q.w;
	}
}
template <int N>
__device__ __forceinline__ void lds_store(int* p, const int* v)
{
#pragma unroll
	for (int i = 0; i < N / 4; ++i)
		*reinterpret_cast<int4*>(p + 4 * i) = make_int4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

__device__ __forceinline__ int med3i(int a, int b, int c)
{
#ifdef ZNET_HOST_TEST
	return max(min(a, b), min(max(a, b), c));
#else
	int r;
	asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
#endif
}

// Batcher's merge exchange (Knuth 5.2.2 Algorithm M) for any N, as a compile-time comparator list.
template <int N>
struct Batcher {
	static constexpr int MAXCE = N * 8 + 8;
	struct List {
		int i[MAXCE];
		int j[MAXCE];
		int n;
	};
	static constexpr List make()
	{
		List L{};
		L.n = 0;
		if (N < 2)
			return L;
		int t = 0;
		while ((1 << t) < N)
			++t;
		for (int p = 1 << (t - 1); p > 0; p >>= 1) {
			int q = 1 << (t - 1), r = 0, d = p;
			while (d > 0) {
				for (int i = 0; i < N - d; ++i) {
					if ((i & p) == r) {
						L.i[L.n] = i;
						L.j[L.n] = i + d;
						++L.n;
					}
				}
				d = q - p;
				q >>= 1;
				r = p;
			}
		}
		return L;
	}
};

template <int N>
__device__ __forceinline__ void sort_net(int (&a)[N])
{
	constexpr auto L = Batcher<N>::make();
#pragma unroll
	for (int c = 0; c < L.n; ++c) {
		const int x = a[L.i[c]], y = a[L.j[c]];
		a[L.i[c]] = min(x, y);
		a[L.j[c]] = max(x, y);
	}
}

// sort a bitonic sequence of 8 ascending (3 half-cleaner stages, 12 comparators)
__device__ __forceinline__ void bitonic_sort8(int (&v)[8])
{
#pragma unroll
	for (int d = 4; d >= 1; d >>= 1) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			if ((i & d) == 0) {
				const int x = v[i], y = v[i + d];
				v[i] = min(x, y);
				v[i + d] = max(x, y);
			}
		}
	}
}

// out[p'] = merge(A, B)[h + p'], p' < h, for sorted A (n = 2h) and sorted B (h)
template <int NA>
__device__ __forceinline__ void merge_mid(const int (&A)[NA], const int (&B)[NA / 2], int (&out)[NA / 2])
{
	constexpr int h = NA / 2;
	if constexpr (h == 1) {
		out[0] = med3i(A[0], A[1], B[0]);
	}
	else if constexpr (h == 8) {
		// ranks 8..15 of A[16] u B[8] by two bitonic half-merges (64 min/max instead of 96):
		// the 16 smallest are A[0..8) plus the 8 smallest of A[8..16) u B, of which we want the top 8.
		int d[8];
#pragma unroll
		for (int i = 0; i < 8; ++i)
			d[i] = min(A[8 + i], B[7 - i]); // the 8 smallest of A[8..16) u B (bitonic)
		bitonic_sort8(d);
#pragma unroll
		for (int i = 0; i < 8; ++i)
			out[i] = max(A[i], d[7 - i]); // the 8 largest of A[0..8) u d (bitonic)
		bitonic_sort8(out);
	}
	else {
#pragma unroll
		for (int pp = 0; pp < h; ++pp) {
			const int p = h + pp;
			int m = A[p];
#pragma unroll
			for (int i = 1; i <= h; ++i)
				m = min(m, max(A[p - i], B[i - 1]));
			out[pp] = m;
		}
	}
}

template <int W, int TN, int G0, int NE, int T>
struct Node {
	static __device__ __forceinline__ void run(const int (&e)[NE], const int (&cand)[TN], int (&out)[T])
	{
		constexpr int h = TN / 2;
		int L[h], R[h], cl[h], cr[h];
#pragma unroll
		for (int i = 0; i < h; ++i)
			L[i] = e[G0 + h - 1 + i]; // the h samples just below the node's core
		sort_net<h>(L);
		merge_mid<TN>(cand, L, cl);
		Node<W, h, G0, NE, T>::run(e, cl, out);
#pragma unroll
		for (int i = 0; i < h; ++i)
			R[i] = e[G0 + W + i]; // the h samples just above the node's core
		sort_net<h>(R);
		merge_mid<TN>(cand, R, cr);
		Node<W, h, G0 + h, NE, T>::run(e, cr, out);
	}
};

template <int W, int G0, int NE, int T>
struct Node<W, 1, G0, NE, T> {
	static __device__ __forceinline__ void run(const int (&)[NE], const int (&cand)[1], int (&out)[T])
	{
		out[G0] = cand[0];
	}
};

// largest power of two <= min(mid + 1, 16)
constexpr int outputs_per_thread(int W)
{
	const int mid = W / 2;
	int t = 1;
	while (2 * t <= mid + 1 && 2 * t <= 16)
		t *= 2;
	return t;
}

// out[g] = median(e[g..g+W-1]) for g < T; NE >= W + T - 1
template <int W, int T, int NE>
__device__ __forceinline__ void medians(const int (&e)[NE], int (&out)[T])
{
	static_assert((W & 1) && W >= 3 && W <= 63, "odd window 3..63");
	static_assert(T >= 1 && T <= W / 2 + 1, "T <= mid + 1");
	static_assert(NE >= W + T - 1, "not enough inputs");
	constexpr int mid = W / 2, NC = W - T + 1;
	int core[NC];
#pragma unroll
	for (int i = 0; i < NC; ++i)
		core[i] = e[T - 1 + i];
	sort_net<NC>(core);
	int cand[T];
#pragma unroll
	for (int i = 0; i < T; ++i)
		cand[i] = core[mid - T + 1 + i];
	Node<W, T, 0, NE, T>::run(e, cand, out);
}

} // namespace znet

// ---------------------------------------------------------------------------------------------------
// Pieces of the block-sharing 47-tap kernel (median47.hip: median47_dpp_kernel; rt_fused.hip).
namespace znet {

// Batcher odd-even merge of the two sorted halves of a[0..N) (N a power of two), as a comparator list.
template <int N>
struct OddEvenMerge {
	static constexpr int MAXCE = N * 4;
	struct List {
		int i[MAXCE];
		int j[MAXCE];
		int n;
	};
	static constexpr void rec(List& L, int lo, int n, int r)
	{
		const int step = r * 2;
		if (step < n) {
			rec(L, lo, n, step);
			rec(L, lo + r, n, step);
			for (int i = lo + r; i + r < lo + n; i += step) {
				L.i[L.n] = i;
				L.j[L.n] = i + r;
				++L.n;
			}
		}
		else {
			L.i[L.n] = lo;
			L.j[L.n] = lo + r;
			++L.n;
		}
	}
	static constexpr List make()
	{
		List L{};
		L.n = 0;
		rec(L, 0, N, 1);
		return L;
	}
};

// a[OFF..OFF+N): both halves sorted -> whole range sorted
template <int N, int OFF, int NA>
__device__ __forceinline__ void oe_merge(int (&a)[NA])
{
	constexpr auto L = OddEvenMerge<N>::make();
#pragma unroll
	for (int c = 0; c < L.n; ++c) {
		const int x = a[OFF + L.i[c]], y = a[OFF + L.j[c]];
		a[OFF + L.i[c]] = min(x, y);
		a[OFF + L.j[c]] = max(x, y);
	}
}

// cand[0..16) = ranks 8..23 of merge(A, B) for sorted A[16], B[16]
__device__ __forceinline__ void mid16_of_two_sorted16(const int (&A)[16], const int (&B)[16], int (&cand)[16])
{
	int lo[16], hi[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) { // lo: the 16 smallest (bitonic), hi: the 16 largest (bitonic)
		lo[i] = min(A[i], B[15 - i]);
		hi[i] = max(A[i], B[15 - i]);
	}
	int u[8], l[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		u[i] = max(lo[i], lo[i + 8]); // the 8 largest of lo  = ranks 8..15
		l[i] = min(hi[i], hi[i + 8]); // the 8 smallest of hi = ranks 16..23
	}
	bitonic_sort8(u);
	bitonic_sort8(l);
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		cand[i] = u[i];
		cand[8 + i] = l[i];
	}
}

} // namespace znet

namespace znet {

// Sorted-block pyramid of one 16-sample block: s16 = the block sorted, oct = its two halves sorted,
// quad = its four quarters sorted (the intermediate stages of Batcher's odd-even merge sort).
__device__ __forceinline__ void pyramid16(const int (&raw)[16], int (&s16)[16], int (&oct)[16], int (&quad)[16])
{
#pragma unroll
	for (int i = 0; i < 16; ++i)
		s16[i] = raw[i];
	oe_merge<2, 0>(s16);
	oe_merge<2, 2>(s16);
	oe_merge<2, 4>(s16);
	oe_merge<2, 6>(s16);
	oe_merge<2, 8>(s16);
	oe_merge<2, 10>(s16);
	oe_merge<2, 12>(s16);
	oe_merge<2, 14>(s16);
	oe_merge<4, 0>(s16);
	oe_merge<4, 4>(s16);
	oe_merge<4, 8>(s16);
	oe_merge<4, 12>(s16);
#pragma unroll
	for (int i = 0; i < 16; ++i)
		quad[i] = s16[i];
	oe_merge<8, 0>(s16);
	oe_merge<8, 8>(s16);
#pragma unroll
	for (int i = 0; i < 16; ++i)
		oct[i] = s16[i];
	oe_merge<16, 0>(s16);
}

// 47-tap medians of 16 consecutive outputs 16t..16t+15 from blocks shared with the neighbours.
// With x the row and B(u) = x[16u-8 .. 16u+7]:
//   A, B   : B(t), B(t+1) sorted                 (the 32 samples common to all 16 windows)
//   lo*    : from B(t-1): octU = its upper half sorted, quad1 / quad3 = its 2nd / 4th quarter sorted,
//            raw = the block as is (pairs and singles are taken from it)
//   hi*    : from B(t+2): octL = lower half sorted, quad0 / quad2 = 1st / 3rd quarter sorted, raw
// Same tree as medians<47,16>: only the sorting of the root and of the 8/4-sample chunks is replaced
// by reads of the neighbours' pyramids.
struct Shared47 {
	int cand[16];
	int lo_oct[8], hi_oct[8];
	int lo_q[2][4], hi_q[2][4];
	int lo_raw[16], hi_raw[16];
};

template <int TN, int G0>
struct Node47 {
	static __device__ __forceinline__ void run(const Shared47& s, const int (&cand)[TN], int (&out)[16])
	{
		constexpr int h = TN / 2;
		int L[h], R[h], cl[h], cr[h];
		if constexpr (TN == 16) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				L[i] = s.lo_oct[i];
				R[i] = s.hi_oct[i];
			}
		}
		else if constexpr (TN == 8) {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				L[i] = s.lo_q[G0 / 8][i];
				R[i] = s.hi_q[G0 / 8][i];
			}
		}
		else if constexpr (TN == 4) {
			L[0] = min(s.lo_raw[G0 + 2], s.lo_raw[G0 + 3]);
			L[1] = max(s.lo_raw[G0 + 2], s.lo_raw[G0 + 3]);
			R[0] = min(s.hi_raw[G0], s.hi_raw[G0 + 1]);
			R[1] = max(s.hi_raw[G0], s.hi_raw[G0 + 1]);
		}
		else {
			L[0] = s.lo_raw[G0 + 1];
			R[0] = s.hi_raw[G0];
		}
		merge_mid<TN>(cand, L, cl);
		if constexpr (h == 1)
			out[G0] = cl[0];
		else
			Node47<h, G0>::run(s, cl, out);
		merge_mid<TN>(cand, R, cr);
		if constexpr (h == 1)
			out[G0 + 1] = cr[0];
		else
			Node47<h, G0 + h>::run(s, cr, out);
	}
};

__device__ __forceinline__ void medians47_shared(const Shared47& s, int (&out)[16])
{
	Node47<16, 0>::run(s, s.cand, out);
}

} // namespace znet
