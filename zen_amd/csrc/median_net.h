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
#include <hip/hip_runtime.h>

namespace znet {

__device__ __forceinline__ int f2key(float f)
{
	int b = __float_as_int(f);
	return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

__device__ __forceinline__ int med3i(int a, int b, int c)
{
	int r;
	asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
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

// out[p'] = merge(A, B)[h + p'], p' < h, for sorted A (n = 2h) and sorted B (h)
template <int NA>
__device__ __forceinline__ void merge_mid(const int (&A)[NA], const int (&B)[NA / 2], int (&out)[NA / 2])
{
	constexpr int h = NA / 2;
	if constexpr (h == 1) {
		out[0] = med3i(A[0], A[1], B[0]);
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
