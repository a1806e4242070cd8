// median_net.hip -- the fast path of the 2-D median filter for odd masks <= 63 taps (every mask of the
// BASELINE configs up to hop 1024: 3/47, 7/23, 11/13), built on the sorting-network scheme of
// median_net.h.  Semantics as median.hip: MedianFilterCPU (libzen/mfilt.h:270-342), replicate border.
//
// Frequency direction (mask along the contiguous axis, mfilt.h:316; the 47-tap kernel of the headline
// metric).  A 256-thread workgroup owns 256*T consecutive outputs of one row (T = 16 for 47 taps: one
// whole 4096-bin row).  HBM -> LDS: every input is loaded once with aligned 16-byte loads (+ the mask
// halo), turned into its ordering key and stored in a padded row image (4 pad words per 64) that makes
// the per-thread 16-byte window reads bank-conflict free.  Each thread then pulls its W+T-1 keys with
// ds_read_b128, runs the network in registers, parks its T results in a second LDS image, and the
// workgroup writes them back with coalesced 16-byte stores.  HBM traffic = 4 B read + 4 B written per
// element (+ halo): the algorithmic minimum of SURVEY 8(d).
//
// Time direction (mask across rows, mfilt.h:311-314; 3/7/11/13 taps on the anticausal/offline path).
// No LDS: a thread owns VC adjacent columns and walks down the rows keeping the last W+T-1 samples of
// each column in registers; every row is read once per workgroup with fully coalesced (up to 16-byte)
// loads, T new rows per step.
#include "common.h"
#include "filters.h"
#include "masks.h"
#include "median_net.h"
#include "row_load.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

using znet::f2key;
using znet::key2f;

// 32-bit row addressing prepared by the host (see prepare())
struct RowMap {
	int base;    // first_row % ring_rows
	int ring;    // ring_rows
	int lo, hi;  // clamp range relative to output row 0
};

__device__ __forceinline__ int map_row(const RowMap& m, int row_rel)
{
	const int r = row_rel < m.lo ? m.lo : (row_rel > m.hi ? m.hi : row_rel);
	int idx = (m.base + r) % m.ring; // wave-uniform: scalar ALU
	if (idx < 0)
		idx += m.ring;
	return idx;
}

// -------------------------------------------------------------------------------------------------
// LDS row image of the frequency kernel: T-word chunks (one per thread) spaced STRIDE = T + PAD words
// apart.  PAD makes the 16 lanes that a ds_read_b128 services together land on 16 different 4-bank
// groups (chunk starts 20t, 12t, 4t words for T = 16, 8, 4: 5t, 3t, t mod 16 are all distinct), and
// because a thread's window is a run of whole chunks its vector v sits at the compile-time offset
// (4v/T)*STRIDE + (4v)%T from its own chunk: one base register, immediate offsets, no address math.
template <int T>
struct RowImage {
	static constexpr int PAD = T >= 8 ? 4 : 0;
	static constexpr int STRIDE = T + PAD;
	static constexpr int LOG2T = T == 16 ? 4 : (T == 8 ? 3 : 2);
	static __device__ __forceinline__ int addr(int g) { return (g >> LOG2T) * STRIDE + (g & (T - 1)); }
	static constexpr int caddr(int g) { return (g / T) * STRIDE + (g % T); }
	static constexpr int words(int n) { return ((n + T - 1) / T) * STRIDE; }
};

using znet::from_key;
using znet::to_key;

// NEIGHBOUR (47 taps only): the 32 samples common to a thread's 16 windows are exactly two aligned
// 16-sample blocks, its own and its right neighbour's own (mid + 1 = 24 = 8 mod 16).  Each thread sorts
// its own block once and receives the neighbour's sorted block through a wave-wide DPP shift (lane i <-
// lane i+1; the last lane of a wave through LDS), then takes the 16 middle ranks of the two sorted
// blocks with one bitonic half-merge: 63 + 48 comparators instead of a 191-comparator 32-sort, no extra
// LDS traffic.  Same results.
template <int W, bool NONNEG, bool NEIGHBOUR>
__global__ __launch_bounds__(256) void median_net_freq_kernel(FilterArgs a, RowMap rm, int segs_per_row, int vec_ok)
{
	constexpr int T = znet::outputs_per_thread(W), mid = W / 2;
	static_assert(T >= 4, "16-byte LDS path needs T >= 4");
	using IM = RowImage<T>;
	constexpr int MID_AL = (mid + 3) & ~3;     // image word 0 is column col0 - MID_AL (16-byte aligned)
	constexpr int DELTA = MID_AL - mid;        // a thread's window starts DELTA words into its chunk
	constexpr int NV = (DELTA + W + T - 1 + 3) / 4, NE = NV * 4;
	constexpr int OUTS = 256 * T;
	constexpr int SPANV = (255 * T) / 4 + NV;  // vectors in the image
	__shared__ __attribute__((aligned(16))) int tile[IM::words(SPANV * 4)];

	const int tid = threadIdx.x;
	const int row = blockIdx.x / segs_per_row, seg = blockIdx.x - row * segs_per_row;
	const int cols = a.cols, col0 = seg * OUTS;
	const float* __restrict__ srow =
	    a.src + (long long)blockIdx.y * a.src_stream_stride + (long long)map_row(rm, row) * cols;
	float* __restrict__ drow = a.dst + (long long)blockIdx.y * a.dst_stream_stride + (long long)row * cols;
	const int c_lo = col0 - MID_AL; // column of image word 0

	// Hermitian rows: only the outputs of columns 0..cols/2 and of the last `mid` columns are wanted
	const int herm = a.hermitian;
	const int o0 = col0 + tid * T; // this thread's first output column
	const bool wanted = !herm || o0 <= (cols >> 1) || o0 + T > cols - mid;
	// Keys: with the caller's promise (NONNEG: magnitudes) the raw bits, which order non-negative samples as they are.
	// Without the promise every sample is keyed as it is staged and every result mapped back (6 VALU instructions per element:
	// 0.687 of the HBM roof where the promise reads 0.712, 13 taps on 1024-bin rows).  -DZEN_MN_DETECT_SIGN (A/B, round 6)
	// instead stages raw bits, ORs the sign bits per wave and re-keys only workgroups that saw one: 0.673 on magnitudes -- the
	// flag's read and branch between the two barriers of a 330-instruction workgroup cost more than the keys -- and 0.574
	// against 0.640 on signed data (a window is re-keyed per thread, not per sample).  Not the default.
#ifdef ZEN_MN_DETECT_SIGN
	constexpr bool DETECT = !NONNEG;
#else
	constexpr bool DETECT = false;
#endif
	constexpr bool RAWKEYS = NONNEG || DETECT; // the image holds raw bits
	__shared__ int wneg[4];
	int sgn = 0;
	if (vec_ok) { // cols % 4 == 0 and 16-byte aligned rows: a vector is wholly inside or wholly outside
		constexpr int NLD = (SPANV + 255) / 256;
		int4 k[NLD];
#pragma unroll
		for (int i = 0; i < NLD; ++i) { // all loads in flight before the first use
			// (the last round covers SPANV - 256 (NLD - 1) vectors only: its other lanes ask for the image's last vector again --
			// one cache line -- instead of the next segment's first 4 KB, which they used to fetch and throw away: rows of
			// several segments moved up to twice their bytes through the L1)
			const int vi = tid + 256 * i < SPANV ? tid + 256 * i : SPANV - 1;
			k[i] = row_vec_keys<RAWKEYS>(srow, c_lo + 4 * vi, cols, herm);
			sgn |= k[i].x | k[i].y | k[i].z | k[i].w;
		}
#pragma unroll
		for (int i = 0; i < NLD; ++i) {
			const int vi = tid + 256 * i;
			if (vi < SPANV)
				*reinterpret_cast<int4*>(&tile[IM::addr(4 * vi)]) = k[i];
		}
	}
	else {
		for (int g = tid; g < SPANV * 4; g += 256) {
			int c = c_lo + g;
			c = c < 0 ? 0 : (c > cols - 1 ? cols - 1 : c);
			ZH_CHK(srow + c, 1);
			const int b = to_key<RAWKEYS>(srow[c]);
			sgn |= b;
			tile[IM::addr(g)] = b;
		}
	}
	if constexpr (DETECT) {
		const bool any = __ballot(sgn < 0) != 0ull;
		if ((tid & 63) == 0)
			wneg[tid >> 6] = any ? 1 : 0;
	}
	__syncthreads();
	bool neg = false; // workgroup-uniform; read BEHIND the window's own LDS reads (in front of them the compiler waited for the
	                  // flags first: a dependent LDS round trip per workgroup).  The re-keying blocks are marked unlikely: laid
	                  // out in line, the two branches a workgroup of magnitudes TOOK around them cost 4.5 % of the launch
	                  // (0.674 against 0.704 of 8 TB/s, 13 taps on 1024-bin rows; variant builds, same box)
	auto read_neg = [&]() {
		if constexpr (DETECT)
			neg = __builtin_amdgcn_readfirstlane(wneg[0] | wneg[1] | wneg[2] | wneg[3]) != 0;
	};
	auto rekey = [](int b) { return b ^ ((b >> 31) & 0x7fffffff); }; // raw bits <-> ordering key (an involution: znet::f2key)

	int out[T];
#pragma unroll
	for (int i = 0; i < T; ++i)
		out[i] = 0; // threads whose outputs nobody wants (Hermitian rows) skip the network
	const int* mine = &tile[tid * IM::STRIDE];
	if constexpr (NEIGHBOUR) {
		static_assert(W == 47 && T == 16 && DELTA == 1, "block alignment of the 47-tap mask");
		__shared__ __attribute__((aligned(16))) int edge[4][16]; // sorted block of lane 0 of waves 1..3, and of block 256
		// e[q] = x[16t-23+q]: chunk tid = left flank (words 1..15), chunk tid+1 = own block B(t),
		// chunk tid+2 = B(t+1) (comes sorted from the neighbour), chunk tid+3 = right flank (words 0..14)
		int e[W + T - 1], A[16], B[16];
		{
			int lo[16], hi[16];
			znet::lds_load<16>(mine + IM::caddr(0), lo);
			znet::lds_load<16>(mine + IM::caddr(16), A);
			znet::lds_load<16>(mine + IM::caddr(48), hi);
#pragma unroll
			for (int q = 0; q < W + T - 1; ++q)
				e[q] = 0;
#pragma unroll
			for (int q = 0; q < 15; ++q) {
				e[q] = lo[q + 1];
				e[47 + q] = hi[q];
			}
		}
		read_neg();
		if (__builtin_expect(neg, 0)) {
#pragma unroll
			for (int q = 0; q < W + T - 1; ++q)
				e[q] = rekey(e[q]);
#pragma unroll
			for (int q = 0; q < 16; ++q)
				A[q] = rekey(A[q]);
		}
		znet::sort_net<16>(A);
		const int lane = tid & 63, wave = tid >> 6;
		if (lane == 0 && wave > 0)
			znet::lds_store<16>(edge[wave - 1], A);
		if (tid == 255) { // the right neighbour of the last block belongs to nobody
			int X[16];
			znet::lds_load<16>(mine + IM::caddr(32), X);
			if (__builtin_expect(neg, 0)) {
#pragma unroll
				for (int q = 0; q < 16; ++q)
					X[q] = rekey(X[q]);
			}
			znet::sort_net<16>(X);
			znet::lds_store<16>(edge[3], X);
		}
#pragma unroll
		for (int i = 0; i < 16; ++i) // lane i <- lane i+1 (lane 63 keeps its own value: replaced below)
			B[i] = __builtin_amdgcn_update_dpp(A[i], A[i], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
		__syncthreads(); // edge blocks published; every window is in registers: the image can take the results
		if (lane == 63)
			znet::lds_load<16>(edge[wave], B);
		int cand[16];
		znet::mid16_of_two_sorted16(A, B, cand);
		znet::Node<W, T, 0, W + T - 1, T>::run(e, cand, out);
	}
	else {
		int ld[NE], e[W + T - 1];
#pragma unroll
		for (int v = 0; v < NV; ++v) {
			const int4 q = *reinterpret_cast<const int4*>(mine + IM::caddr(4 * v));
			ld[4 * v] = q.x;
			ld[4 * v + 1] = q.y;
			ld[4 * v + 2] = q.z;
			ld[4 * v + 3] = q.w;
		}
#pragma unroll
		for (int q = 0; q < W + T - 1; ++q)
			e[q] = ld[q + DELTA];
		read_neg();
		__syncthreads(); // every window is in registers: the image can take the results
		if (__builtin_expect(neg, 0)) {
#pragma unroll
			for (int q = 0; q < W + T - 1; ++q)
				e[q] = rekey(e[q]);
		}
		if (wanted)
			znet::medians<W, T, W + T - 1>(e, out);
	}
	if (__builtin_expect(neg, 0)) { // back to the samples' bits
#pragma unroll
		for (int i = 0; i < T; ++i)
			out[i] = rekey(out[i]);
	}
#pragma unroll
	for (int v = 0; v < T / 4; ++v)
		*reinterpret_cast<int4*>(&tile[tid * IM::STRIDE + 4 * v]) =
		    make_int4(out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]);
	__syncthreads();

	if (vec_ok) {
#pragma unroll
		for (int i = 0; i < T / 4; ++i) {
			const int g = 4 * tid + 1024 * i;
			const int c = col0 + g;
			if (c < cols && (!herm || c <= (cols >> 1) || c + 4 > cols - mid)) {
				const int4 k = *reinterpret_cast<const int4*>(&tile[IM::addr(g)]);
				float* dq = reinterpret_cast<float*>(__builtin_assume_aligned(drow + c, 16)); // streaming store
				ZH_CHK(dq, 4);
				__builtin_nontemporal_store(from_key<RAWKEYS>(k.x), dq);
				__builtin_nontemporal_store(from_key<RAWKEYS>(k.y), dq + 1);
				__builtin_nontemporal_store(from_key<RAWKEYS>(k.z), dq + 2);
				__builtin_nontemporal_store(from_key<RAWKEYS>(k.w), dq + 3);
			}
		}
	}
	else {
		for (int g = tid; g < OUTS; g += 256) {
			const int c = col0 + g;
			if (c < cols) {
				ZH_CHK(drow + c, 1);
				drow[c] = from_key<RAWKEYS>(tile[IM::addr(g)]);
			}
		}
	}
}

// -------------------------------------------------------------------------------------------------
// Hermitian rows (FilterArgs::hermitian; the engine's half rows) that fit one segment: of a row's cols / T chunks of
// T outputs only the first cols / (2T) + 1 (columns 0 .. cols/2) and the last ceil(mid / T) (the last `mid` columns,
// whose replicate border is not the mirror image of the first ones') are wanted -- 131 of 256 for every mask of the
// BASELINE configurations -- and the kernel above leaves the other half of its workgroup idle.  Here a 448-thread
// workgroup takes THREE rows, one thread per wanted chunk, stages only the columns those chunks read (the mirrored
// upper half comes from the stored half, row_vec_keys) and stores straight from registers: consecutive threads write
// consecutive 16-byte vectors.  Same network, same values.
constexpr int HERM_THREADS = 448;

// BITS (FilterArgs::bits_t, hard masks of non-negative rows): instead of storing P the thread compares it with the
// harmonic estimate of its bins (FilterArgs::hrows, stored half) and ORs the two mask bits of every bin into the row's
// words in LDS, in the synthesis threads' order (stft.h IstftArgs::bits_t): bin k at word k mod cols/16, bits 2*(k /
// (cols/16)); a bin of the lower half also at its mirror image's place unless that lies among the last `mid` bins, whose
// own P comes from the tail chunks.  The workgroup holds whole rows, so it writes finished words.
template <int W, bool NONNEG, bool BITS = false>
__global__ __launch_bounds__(HERM_THREADS) void median_net_freq_herm_kernel(FilterArgs a, RowMap rm, int n_lo, int n_tail,
                                                                            int lwv, int twv)
{
	static_assert(!BITS || NONNEG, "mask bits: magnitudes only");
	constexpr int T = znet::outputs_per_thread(W), mid = W / 2;
	static_assert(T >= 4, "16-byte LDS path needs T >= 4");
	constexpr int MID_AL = (mid + 3) & ~3, DELTA = MID_AL - mid;
	constexpr int NV = (DELTA + W + T - 1 + 3) / 4, NE = NV * 4;
	extern __shared__ __attribute__((aligned(16))) int himg[]; // per row: lwv vectors (columns -MID_AL ..), then twv (the tail)
	const int tid = threadIdx.x, cols = a.cols;
	const int jobs = n_lo + n_tail, rpw = HERM_THREADS / jobs, row_vecs = lwv + twv;
	const int row0 = blockIdx.x * rpw;
	const int c_t0 = cols - n_tail * T; // first column of the tail chunks
	const float* __restrict__ src = a.src + (long long)blockIdx.y * a.src_stream_stride;
	for (int v = tid; v < rpw * row_vecs; v += HERM_THREADS) {
		const int rr = v / row_vecs, vv = v - rr * row_vecs;
		const int rw = row0 + rr < a.n_out_rows ? row0 + rr : a.n_out_rows - 1; // (past the last row: staged again, never used)
		const float* srow = src + (long long)map_row(rm, rw) * cols;
		const int col = vv < lwv ? 4 * vv - MID_AL : c_t0 - MID_AL + 4 * (vv - lwv);
		*reinterpret_cast<int4*>(&himg[4 * v]) = row_vec_keys<NONNEG>(srow, col, cols, 1);
	}
	unsigned* tw = reinterpret_cast<unsigned*>(himg) + 4 * rpw * row_vecs; // BITS: rpw rows of cols/16 words
	const int tfw = cols >> 4;
	if constexpr (BITS) {
		for (int k = tid; k < rpw * tfw; k += HERM_THREADS)
			tw[k] = 0u;
	}
	__syncthreads();
	const int rr = tid / jobs, j = tid - rr * jobs;
	const bool work = rr < rpw && row0 + rr < a.n_out_rows;
	if (!BITS && !work)
		return;
	const bool tail = j >= n_lo;
	const int rra = rr < rpw ? rr : 0; // (BITS: threads without a chunk stay for the barrier below and read row 0's image)
	const int* mine = &himg[4 * (rra * row_vecs + (tail ? lwv : 0)) + T * (tail ? j - n_lo : j)]; // word 0 = chunk's first column - MID_AL
	int ld[NE], e[W + T - 1], out[T];
#pragma unroll
	for (int v = 0; v < NV; ++v) {
		const int4 q = *reinterpret_cast<const int4*>(mine + 4 * v);
		ld[4 * v] = q.x;
		ld[4 * v + 1] = q.y;
		ld[4 * v + 2] = q.z;
		ld[4 * v + 3] = q.w;
	}
#pragma unroll
	for (int q = 0; q < W + T - 1; ++q)
		e[q] = ld[q + DELTA];
	if constexpr (BITS) {
		if (work) {
			znet::medians<W, T, W + T - 1>(e, out);
			const int c0 = tail ? c_t0 + T * (j - n_lo) : T * j; // first bin of the chunk
			const float* hrow = a.hrows + (long long)blockIdx.y * a.h_stream_stride + (long long)(row0 + rr) * cols;
			float h[T]; // H of the bins (the tail's from the mirror image: H is symmetric, only its lower half is stored)
#pragma unroll
			for (int i = 0; i < T; ++i) {
				ZH_CHK(hrow + (tail ? cols - (c0 + i) : c0 + i), 1);
				h[i] = hrow[tail ? cols - (c0 + i) : c0 + i];
			}
			int log2tf = 0;
			while ((1 << log2tf) < tfw)
				++log2tf;
			unsigned* trow = tw + rr * tfw;
#pragma unroll
			for (int i = 0; i < T; ++i) {
				const int k = c0 + i;
				const float pf = __int_as_float(out[i]);
				const unsigned pm = a.need_pm && hard_mask_exact(pf, h[i] + FLT_EPSILON, a.thr_p) != 0.0f ? 1u : 0u;
				const unsigned hm = a.need_hm && hard_mask_exact(h[i], pf + FLT_EPSILON, a.thr_h) != 0.0f ? 1u : 0u;
				const unsigned code = pm | (hm << 1);
				const bool wanted = tail ? k >= cols - mid : k <= (cols >> 1);
				if (wanted && code)
					atomicOr(&trow[k & (tfw - 1)], code << (2 * (k >> log2tf)));
				const int km = cols - k; // the mirror image of a bin of the lower half
				if (!tail && k > mid && k < (cols >> 1) && code)
					atomicOr(&trow[km & (tfw - 1)], code << (2 * (km >> log2tf)));
			}
		}
		__syncthreads();
		unsigned* dst_t = a.bits_t + (long long)blockIdx.y * a.bits_t_stream_stride + (long long)row0 * tfw;
		const int rows_here = a.n_out_rows - row0 < rpw ? a.n_out_rows - row0 : rpw;
		for (int k = tid; k < rows_here * tfw; k += HERM_THREADS) {
			ZH_CHK(dst_t + k, 1);
			dst_t[k] = tw[k];
		}
		return;
	}
	znet::medians<W, T, W + T - 1>(e, out);
	float* d = a.dst + (long long)blockIdx.y * a.dst_stream_stride + (long long)(row0 + rr) * cols + (tail ? c_t0 + T * (j - n_lo) : T * j);
#pragma unroll
	for (int v = 0; v < T / 4; ++v) {
		float* dq = reinterpret_cast<float*>(__builtin_assume_aligned(d + 4 * v, 16)); // streaming store
		ZH_CHK(dq, 4);
#pragma unroll
		for (int i = 0; i < 4; ++i)
			__builtin_nontemporal_store(from_key<NONNEG>(out[4 * v + i]), dq + i);
	}
}

// -------------------------------------------------------------------------------------------------
// Time median AND frequency median of the same rows in one launch (pass 2 of the offline path: 11 / 13 taps on 1024-bin
// half rows): the harmonic estimate never reaches HBM.  The two-launch form wrote H (a row of the stored half per frame)
// from median_net_time_kernel and read it back here next to the magnitudes; now a workgroup of HERM_THREADS
//   A. computes H for GROUPS * TT consecutive rows -- thread (g, c): TT rows of four adjacent columns from WT + TT - 1
//      magnitude rows, the sorting-network scheme of median_net_time_kernel -- and leaves them in LDS;
//   B. runs median_net_freq_herm_kernel<WF, true, BITS>'s stage over the same rows, three at a time: stage the columns the
//      wanted chunks read, frequency medians, compare with H from LDS, OR the two mask bits of every bin into the rows'
//      words (the synthesis threads' order, stft.h IstftArgs::bits_t), store the finished words.
// Same comparisons on the same values: bit-identical masks ("no_median_tf": the two launches).  The magnitude rows are read
// (WT + TT - 1) / TT times in phase A and once in phase B, from L2 after the first touch; HBM sees them ~1.2 times.
template <int WT, int WF>
__global__ __launch_bounds__(HERM_THREADS) void median_tf_herm_bits_kernel(FilterArgs a, RowMap rm, int n_lo, int n_tail, int lwv, int twv)
{
	constexpr int TT = znet::outputs_per_thread(WT), midT = WT / 2, NET = WT + TT - 1, GROUPS = 3, RWG = TT * GROUPS;
	constexpr int T = znet::outputs_per_thread(WF), mid = WF / 2;
	static_assert(T >= 4, "16-byte LDS path needs T >= 4");
	constexpr int MID_AL = (mid + 3) & ~3, DELTA = MID_AL - mid;
	constexpr int NV = (DELTA + WF + T - 1 + 3) / 4, NE = NV * 4;
	extern __shared__ __attribute__((aligned(16))) int himg[]; // [3 rows of the frequency stage | their bit words | H of RWG rows]
	const int tid = threadIdx.x, cols = a.cols;
	const int jobs = n_lo + n_tail, rpw = HERM_THREADS / jobs, row_vecs = lwv + twv;
	const int tfw = cols >> 4;
	unsigned* tw = reinterpret_cast<unsigned*>(himg) + 4 * rpw * row_vecs; // rpw rows of cols/16 words
	const int hs = (cols >> 1) + 8;                                          // floats per H row (bins 0..cols/2, rounded up to vectors)
	float* Hs = reinterpret_cast<float*>(tw + rpw * tfw);
	const int row0 = blockIdx.x * RWG;
	const float* __restrict__ src = a.src + (long long)blockIdx.y * a.src_stream_stride;

	// ---- A: the time medians of rows row0 .. row0 + RWG - 1, stored half of every row
	{
		const int vecs = ((cols >> 1) + 4) >> 2; // column vectors of the stored half (bins 0..cols/2, rounded up)
		if (tid < GROUPS * vecs) {
			const int g = tid / vecs, c = (tid - g * vecs) << 2;
			const int r0 = row0 + g * TT;
			int e[4][NET];
#pragma unroll
			for (int q = 0; q < NET; ++q) { // taps r0 - midT .. r0 + midT + TT - 1 (replicate border: map_row clamps)
				ZH_CHK(src + ((long long)map_row(rm, r0 - midT + q) * cols + c), 4);
				const float4 x = *reinterpret_cast<const float4*>(src + (long long)map_row(rm, r0 - midT + q) * cols + c);
				e[0][q] = __float_as_int(x.x); // |S| >= +0: the bits are the ordering key
				e[1][q] = __float_as_int(x.y);
				e[2][q] = __float_as_int(x.z);
				e[3][q] = __float_as_int(x.w);
			}
			__builtin_amdgcn_sched_barrier(0); // all NET loads before the first comparator: the scheduler had woven the networks into
			                                   // the loads, eight dependent trips to the L2 per thread
			int out[4][TT];
#pragma unroll
			for (int v = 0; v < 4; ++v)
				znet::medians<WT, TT, NET>(e[v], out[v]);
#pragma unroll
			for (int gI = 0; gI < TT; ++gI)
				*reinterpret_cast<float4*>(Hs + (g * TT + gI) * hs + c) =
				    make_float4(__int_as_float(out[0][gI]), __int_as_float(out[1][gI]), __int_as_float(out[2][gI]), __int_as_float(out[3][gI]));
		}
	}
	// ---- B: the frequency stage, rpw rows per turn
	const int c_t0 = cols - n_tail * T; // first column of the tail chunks
	int log2tf = 0;
	while ((1 << log2tf) < tfw)
		++log2tf;
	const int rr = tid / jobs, j = tid - rr * jobs;
	const bool tail = j >= n_lo;
	const int rra = rr < rpw ? rr : 0; // (threads without a chunk stay for the barriers and read row 0's image)
	for (int rb = 0; rb < RWG; rb += rpw) {
		const int rowb = row0 + rb;
		if (rowb >= a.n_out_rows)
			break; // (workgroup-uniform)
		__syncthreads(); // H is complete (first turn); the last turn's words are stored and its image is free
		for (int v = tid; v < rpw * row_vecs; v += HERM_THREADS) {
			const int r1 = v / row_vecs, vv = v - r1 * row_vecs;
			const int rw = rowb + r1 < a.n_out_rows ? rowb + r1 : a.n_out_rows - 1; // (past the last row: staged again, never used)
			const float* srow = src + (long long)map_row(rm, rw) * cols;
			const int col = vv < lwv ? 4 * vv - MID_AL : c_t0 - MID_AL + 4 * (vv - lwv);
			*reinterpret_cast<int4*>(&himg[4 * v]) = row_vec_keys<true>(srow, col, cols, 1);
		}
		for (int k = tid; k < rpw * tfw; k += HERM_THREADS)
			tw[k] = 0u;
		__syncthreads();
		const bool work = rr < rpw && rb + rr < RWG && rowb + rr < a.n_out_rows;
		if (work) {
			const int* mine = &himg[4 * (rra * row_vecs + (tail ? lwv : 0)) + T * (tail ? j - n_lo : j)];
			int ld[NE], e[WF + T - 1], out[T];
#pragma unroll
			for (int v = 0; v < NV; ++v) {
				const int4 q = *reinterpret_cast<const int4*>(mine + 4 * v);
				ld[4 * v] = q.x;
				ld[4 * v + 1] = q.y;
				ld[4 * v + 2] = q.z;
				ld[4 * v + 3] = q.w;
			}
#pragma unroll
			for (int q = 0; q < WF + T - 1; ++q)
				e[q] = ld[q + DELTA];
			znet::medians<WF, T, WF + T - 1>(e, out);
			const int c0 = tail ? c_t0 + T * (j - n_lo) : T * j; // first bin of the chunk
			const float* hrow = Hs + (rb + rr) * hs;
			unsigned* trow = tw + rr * tfw;
			float hv[T]; // (the T harmonic estimates read together, not one in front of every comparison)
#pragma unroll
			for (int i = 0; i < T; ++i)
				hv[i] = hrow[tail ? cols - (c0 + i) : c0 + i]; // (the tail's H from the mirror image: H is symmetric)
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int i = 0; i < T; ++i) {
				const int k = c0 + i;
				const float hh = hv[i];
				const float pf = __int_as_float(out[i]);
				const unsigned pm = a.need_pm && hard_mask_exact(pf, hh + FLT_EPSILON, a.thr_p) != 0.0f ? 1u : 0u;
				const unsigned hm = a.need_hm && hard_mask_exact(hh, pf + FLT_EPSILON, a.thr_h) != 0.0f ? 1u : 0u;
				const unsigned code = pm | (hm << 1);
				const bool wanted = tail ? k >= cols - mid : k <= (cols >> 1);
				if (wanted && code)
					atomicOr(&trow[k & (tfw - 1)], code << (2 * (k >> log2tf)));
				const int km = cols - k; // the mirror image of a bin of the lower half
				if (!tail && k > mid && k < (cols >> 1) && code)
					atomicOr(&trow[km & (tfw - 1)], code << (2 * (km >> log2tf)));
			}
		}
		__syncthreads();
		unsigned* dst_t = a.bits_t + (long long)blockIdx.y * a.bits_t_stream_stride + (long long)rowb * tfw;
		int rows_here = a.n_out_rows - rowb < rpw ? a.n_out_rows - rowb : rpw;
		rows_here = RWG - rb < rows_here ? RWG - rb : rows_here;
		for (int k = tid; k < rows_here * tfw; k += HERM_THREADS) {
			ZH_CHK(dst_t + k, 1);
			dst_t[k] = tw[k];
		}
	}
}

// -------------------------------------------------------------------------------------------------
template <int VC>
struct VecT;
template <>
struct VecT<1> {
	using type = float;
};
template <>
struct VecT<2> {
	using type = float2;
};
template <>
struct VecT<4> {
	using type = float4;
};

template <int VC>
__device__ __forceinline__ void load_keys(const float* p, int (&k)[VC])
{
	ZH_CHK(p, VC);
	if constexpr (VC == 1) {
		k[0] = f2key(*p);
	}
	else if constexpr (VC == 2) {
		const float2 x = *reinterpret_cast<const float2*>(p);
		k[0] = f2key(x.x);
		k[1] = f2key(x.y);
	}
	else { // (streaming load: aligned rows, the four loads become one 16-byte instruction)
		const float* q = reinterpret_cast<const float*>(__builtin_assume_aligned(p, 16));
		k[0] = f2key(__builtin_nontemporal_load(q));
		k[1] = f2key(__builtin_nontemporal_load(q + 1));
		k[2] = f2key(__builtin_nontemporal_load(q + 2));
		k[3] = f2key(__builtin_nontemporal_load(q + 3));
	}
}

template <int VC>
__device__ __forceinline__ void store_keys(float* p, const int (&k)[VC])
{
	ZH_CHK(p, VC);
	if constexpr (VC == 4) { // streaming store
#pragma unroll
		for (int i = 0; i < 4; ++i)
			__builtin_nontemporal_store(key2f(k[i]), p + i);
		return;
	}
	if constexpr (VC == 1) {
		*p = key2f(k[0]);
	}
	else if constexpr (VC == 2) {
		*reinterpret_cast<float2*>(p) = make_float2(key2f(k[0]), key2f(k[1]));
	}
	else {
		*reinterpret_cast<float4*>(p) = make_float4(key2f(k[0]), key2f(k[1]), key2f(k[2]), key2f(k[3]));
	}
}

constexpr int time_vc(int W) { return W <= 15 ? 4 : (W <= 31 ? 2 : 1); }

// out[g] = median(e[g..g+W-1]), g < T.  Three taps: a v_med3 per output, any T; else the shared-sort scheme (T <= mid + 1)
template <int W, int T, int NE>
__device__ __forceinline__ void time_medians(const int (&e)[NE], int (&out)[T])
{
	if constexpr (W == 3) {
#pragma unroll
		for (int g = 0; g < T; ++g)
			out[g] = znet::med3i(e[g], e[g + 1], e[g + 2]);
	}
	else {
		znet::medians<W, T, NE>(e, out);
	}
}

// rows per step of the time kernel: the sorting scheme's outputs per thread; three taps (no sharing to be had): eight rows, so
// that a thread has 8 x 16 B in flight instead of 2 x 16
constexpr int time_rows(int W) { return W == 3 ? 8 : znet::outputs_per_thread(W); }

// (Round 6, measured and not kept: the NEXT step's T rows requested before this step's medians are computed and stored -- gfx9
// counts loads and stores in one in-order counter, so the wait for a step's loads first drains the previous step's stores.
// Interleaved A/B on one box, fraction of 8 TB/s: 103 360 x 1024 / 11 taps 0.661 against 0.649, but 8192^2 0.612 against 0.660,
// 16384^2 0.697 against 0.732, 3 / 7 taps 0.679 / 0.693 against 0.687 / 0.698: the 14-16 registers cost a wave of occupancy.)
template <int W, int VC>
__global__ __launch_bounds__(256) void median_net_time_kernel(FilterArgs a, RowMap rm, int rows_per_block)
{
	constexpr int T = time_rows(W), mid = W / 2, NE = W + T - 1;
	const int cols = a.cols;
	const int pitch = a.pitch ? a.pitch : cols; // floats between rows (rows of which only the first cols are filtered)
	// thread -> (column vector, block of rows), the column fastest: a row of 516 columns (the stored half of a 1024-bin
	// spectrum) is 129 vectors, and with one block of rows per workgroup the third wavefront ran for a single lane
	const int vecs = (cols + VC - 1) / VC;
	const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
	const int rb = (int)(t / vecs);
	const int c = (int)(t - (long long)rb * vecs) * VC;
	const int r0 = rb * rows_per_block;
	if (r0 >= a.n_out_rows)
		return;
	const float* __restrict__ src = a.src + (long long)blockIdx.z * a.src_stream_stride + c;
	float* __restrict__ dst = a.dst + (long long)blockIdx.z * a.dst_stream_stride + c;
	int rend = r0 + rows_per_block;
	rend = rend > a.n_out_rows ? a.n_out_rows : rend;

	int e[VC][NE];
#pragma unroll
	for (int q = 0; q < W - 1; ++q) { // taps r0-mid .. r0+mid-1
		int k[VC];
		load_keys<VC>(src + (long long)map_row(rm, r0 - mid + q) * pitch, k);
#pragma unroll
		for (int v = 0; v < VC; ++v)
			e[v][q] = k[v];
	}
	for (int rr = r0; rr < rend; rr += T) {
#pragma unroll
		for (int i = 0; i < T; ++i) { // T new rows: rr+mid .. rr+mid+T-1
			int k[VC];
			load_keys<VC>(src + (long long)map_row(rm, rr + mid + i) * pitch, k);
#pragma unroll
			for (int v = 0; v < VC; ++v)
				e[v][W - 1 + i] = k[v];
		}
		__builtin_amdgcn_sched_barrier(0); // (the T new rows requested together, before the first comparator)
		int out[VC][T];
#pragma unroll
		for (int v = 0; v < VC; ++v)
			time_medians<W, T, NE>(e[v], out[v]);
#pragma unroll
		for (int g = 0; g < T; ++g) {
			if (rr + g < rend) {
				int k[VC];
#pragma unroll
				for (int v = 0; v < VC; ++v)
					k[v] = out[v][g];
				store_keys<VC>(dst + (long long)(rr + g) * pitch, k);
			}
		}
#pragma unroll
		for (int v = 0; v < VC; ++v)
#pragma unroll
			for (int q = 0; q < W - 1; ++q)
				e[v][q] = e[v][q + T];
	}
}

// -------------------------------------------------------------------------------------------------
bool prepare(const FilterArgs& a, RowMap* rm)
{
	const long long lim = 0x3fffffff;
	if (a.ring_rows <= 0 || a.ring_rows > lim || a.n_out_rows > lim)
		return false;
	rm->ring = (int)a.ring_rows;
	rm->base = (int)(a.first_row % a.ring_rows);
	const long long lo = a.clamp_lo - a.first_row, hi = a.clamp_hi - a.first_row;
	rm->lo = (int)(lo < -lim ? -lim : (lo > lim ? lim : lo));
	rm->hi = (int)(hi < -lim ? -lim : (hi > lim ? lim : hi));
	return true;
}

template <int W>
int launch_freq(const FilterArgs& a, const RowMap& rm, hipStream_t stream, int* bits_done)
{
	constexpr int T = znet::outputs_per_thread(W);
	const int segs = (a.cols + 256 * T - 1) / (256 * T);
	const int vec_ok = (a.cols % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.src) & 15) == 0)
	                   && ((reinterpret_cast<uintptr_t>(a.dst) & 15) == 0) && (a.src_stream_stride % 4 == 0)
	                   && (a.dst_stream_stride % 4 == 0);
	dim3 grid((unsigned)((long long)a.n_out_rows * segs), (unsigned)a.n_streams);
	if (a.hermitian && !vec_ok)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "median: Hermitian rows need 16-byte aligned rows of a multiple of 4 columns");
	if (a.hermitian && segs == 1 && a.cols % (2 * T) == 0 && !g_opt_median_general) { // three rows per workgroup: median_net_freq_herm_kernel
		constexpr int mid = W / 2, MID_AL = (mid + 3) & ~3, DELTA = MID_AL - mid, NV = (DELTA + W + T - 1 + 3) / 4;
		const int n_lo = a.cols / (2 * T) + 1, n_tail = (mid + T - 1) / T, jobs = n_lo + n_tail;
		// a chunk's window starts T*j words into its piece of the image and is NV vectors long
		const int lwv = (T * (n_lo - 1)) / 4 + NV, twv = (T * (n_tail - 1)) / 4 + NV;
		if (jobs <= HERM_THREADS && a.n_out_rows < 0x3fffffff) {
			const int rpw = HERM_THREADS / jobs;
			const size_t lds = sizeof(int) * 4 * (size_t)rpw * (lwv + twv);
			dim3 g((unsigned)((a.n_out_rows + rpw - 1) / rpw), (unsigned)a.n_streams);
			// mask bits instead of P (power-of-two rows of at least 16 words: every bin's place is a shift and a mask)
			if constexpr (W == 13 || W == 11 || W == 23 || W == 21) { // both medians in one launch (FilterArgs::time_len)
				if (bits_done && a.bits_t && a.time_len && a.nonneg && (a.cols & (a.cols - 1)) == 0 && a.cols >= 256) {
					const int wt = a.time_len;
					const int tt = znet::outputs_per_thread(wt), rwg = 3 * tt;
					const size_t lds_f = lds + sizeof(unsigned) * (size_t)rpw * (a.cols >> 4) + sizeof(float) * (size_t)rwg * ((a.cols >> 1) + 8);
					dim3 gf((unsigned)((a.n_out_rows + rwg - 1) / rwg), (unsigned)a.n_streams);
					constexpr int WT_FOR = W == 13 ? 11 : (W == 11 ? 13 : 7); // the pairs of the BASELINE geometries (44.1 / 48 kHz)
					if (wt == WT_FOR) {
						hipLaunchKernelGGL((median_tf_herm_bits_kernel<WT_FOR, W>), gf, dim3(HERM_THREADS), lds_f, stream, a, rm, n_lo, n_tail, lwv, twv);
						ZH_HIP(hipGetLastError());
						*bits_done = 2;
						return ZEN_HIP_OK;
					}
				}
			}
			if (bits_done && a.bits_t && a.hrows && a.nonneg && (a.cols & (a.cols - 1)) == 0 && a.cols >= 256) {
				const size_t lds_b = lds + sizeof(unsigned) * (size_t)rpw * (a.cols >> 4);
				hipLaunchKernelGGL((median_net_freq_herm_kernel<W, true, true>), g, dim3(HERM_THREADS), lds_b, stream, a, rm, n_lo, n_tail, lwv,
				                   twv);
				ZH_HIP(hipGetLastError());
				*bits_done = 2;
				return ZEN_HIP_OK;
			}
			if (a.nonneg)
				hipLaunchKernelGGL((median_net_freq_herm_kernel<W, true>), g, dim3(HERM_THREADS), lds, stream, a, rm, n_lo, n_tail, lwv, twv);
			else
				hipLaunchKernelGGL((median_net_freq_herm_kernel<W, false>), g, dim3(HERM_THREADS), lds, stream, a, rm, n_lo, n_tail, lwv, twv);
			ZH_HIP(hipGetLastError());
			return ZEN_HIP_OK;
		}
	}
	if constexpr (W == 47) {
		if (!g_opt_no_median47_neighbour && !a.hermitian) {
			if (a.nonneg)
				hipLaunchKernelGGL((median_net_freq_kernel<W, true, true>), grid, dim3(256), 0, stream, a, rm, segs, vec_ok);
			else
				hipLaunchKernelGGL((median_net_freq_kernel<W, false, true>), grid, dim3(256), 0, stream, a, rm, segs, vec_ok);
			ZH_HIP(hipGetLastError());
			return ZEN_HIP_OK;
		}
	}
	if (a.nonneg)
		hipLaunchKernelGGL((median_net_freq_kernel<W, true, false>), grid, dim3(256), 0, stream, a, rm, segs, vec_ok);
	else
		hipLaunchKernelGGL((median_net_freq_kernel<W, false, false>), grid, dim3(256), 0, stream, a, rm, segs, vec_ok);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

// 3- and 5-tap frequency masks (hops <= 128: l_perc = 500 / (fs / nfft) is 1..5 bins): one thread per four
// outputs straight from global memory (the 2*mid halo samples come out of L1), med3 / the 5-input median
// identity med3(e, max(min(a,b), min(c,d)), min(max(a,b), max(c,d))).
template <int W, bool NONNEG>
__global__ __launch_bounds__(256) void median_tiny_freq_kernel(FilterArgs a, RowMap rm, int groups_per_row)
{
	constexpr int mid = W / 2;
	const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
	const int row = (int)(gid / groups_per_row), g = (int)(gid - (long long)row * groups_per_row);
	if (row >= a.n_out_rows)
		return;
	const int cols = a.cols, c0 = 4 * g;
	const float* __restrict__ srow =
	    a.src + (long long)blockIdx.y * a.src_stream_stride + (long long)map_row(rm, row) * cols;
	float* __restrict__ drow = a.dst + (long long)blockIdx.y * a.dst_stream_stride + (long long)row * cols;
	if (a.hermitian && c0 > (cols >> 1) && c0 + 4 <= cols - mid)
		return; // Hermitian rows: only columns 0..cols/2 and the last `mid` ones are wanted
	int k[4 + 2 * mid];
#pragma unroll
	for (int i = 0; i < 4 + 2 * mid; ++i) {
		int c = c0 - mid + i;
		c = c < 0 ? 0 : (c > cols - 1 ? cols - 1 : c); // replicate border (ippBorderRepl)
		if (a.hermitian && c > (cols >> 1))
			c = cols - c; // the stored half of a Hermitian row
		k[i] = to_key<NONNEG>(srow[c]);
	}
#pragma unroll
	for (int o = 0; o < 4; ++o) {
		int m;
		if constexpr (W == 3) {
			m = znet::med3i(k[o], k[o + 1], k[o + 2]);
		}
		else {
			const int lo = max(min(k[o], k[o + 1]), min(k[o + 2], k[o + 3]));
			const int hi = min(max(k[o], k[o + 1]), max(k[o + 2], k[o + 3]));
			m = znet::med3i(k[o + 4], lo, hi);
		}
		if (c0 + o < cols)
			drow[c0 + o] = from_key<NONNEG>(m);
	}
}

template <int W>
int launch_tiny_freq(const FilterArgs& a, const RowMap& rm, hipStream_t stream)
{
	const int groups = (a.cols + 3) / 4;
	const long long threads = (long long)a.n_out_rows * groups;
	dim3 grid((unsigned)((threads + 255) / 256), (unsigned)a.n_streams);
	if (a.nonneg)
		hipLaunchKernelGGL((median_tiny_freq_kernel<W, true>), grid, dim3(256), 0, stream, a, rm, groups);
	else
		hipLaunchKernelGGL((median_tiny_freq_kernel<W, false>), grid, dim3(256), 0, stream, a, rm, groups);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int W>
int launch_freq_guard(const FilterArgs& a, const RowMap& rm, hipStream_t stream, int* bits_done)
{
	if constexpr (W >= 7)
		return launch_freq<W>(a, rm, stream, bits_done);
	else
		return launch_tiny_freq<W>(a, rm, stream);
}

template <int W, int VC>
int launch_time_vc(const FilterArgs& a, const RowMap& rm, hipStream_t stream)
{
	constexpr int T = time_rows(W);
	const int vecs = (a.cols + VC - 1) / VC; // column vectors of a row
	// enough blocks of rows to fill the chip (>= ~1024 full workgroups), but long enough to amortise the W-1 halo
	int rpb = 256;
	// (round 6, interleaved A/B on the path shapes and the reference's squares: 512 / 768 / 2048 workgroups instead of 1024 lose
	// 1-30 %; two columns per thread -- twice the threads per row, blocks of rows twice as long, half the halo -- lose 4-20 %)
	while (rpb > 4 * T && rpb > 32 && (long long)vecs * ((a.n_out_rows + rpb - 1) / rpb) * a.n_streams < 1024LL * 256)
		rpb >>= 1;
	rpb = (rpb + T - 1) / T * T;
	const long long threads = (long long)vecs * ((a.n_out_rows + rpb - 1) / rpb);
	dim3 grid((unsigned)((threads + 255) / 256), 1, (unsigned)a.n_streams);
	hipLaunchKernelGGL((median_net_time_kernel<W, VC>), grid, dim3(256), 0, stream, a, rm, rpb);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int W>
int launch_time(const FilterArgs& a, const RowMap& rm, hipStream_t stream)
{
	constexpr int VC = time_vc(W);
	const bool al = (reinterpret_cast<uintptr_t>(a.src) % (4 * VC) == 0)
	                && (reinterpret_cast<uintptr_t>(a.dst) % (4 * VC) == 0) && a.cols % VC == 0
	                && a.src_stream_stride % VC == 0 && a.dst_stream_stride % VC == 0;
	if (VC > 1 && !al)
		return launch_time_vc<W, 1>(a, rm, stream);
	return launch_time_vc<W, VC>(a, rm, stream);
}

} // namespace

#define ZH_NET_WIDTHS(X) X(3) X(5) X(7) X(9) X(11) X(13) X(15) X(17) X(19) X(21) X(23) X(25) X(27) X(29) X(31) \
	X(33) X(35) X(37) X(39) X(41) X(43) X(45) X(47) X(49) X(51) X(53) X(55) X(57) X(59) X(61) X(63)

// returns ZEN_HIP_OK and sets *handled when the mask length / geometry is covered by the fast path
int launch_median_net(const FilterArgs& a, hipStream_t stream, bool* handled, int* bits_done)
{
	*handled = false;
	RowMap rm;
	if (a.len < 3 || a.len > 63 || !prepare(a, &rm))
		return ZEN_HIP_OK;
	const bool freq = a.direction == ZEN_HIP_FREQUENCY;
	if ((long long)a.n_out_rows * ((a.cols + 3) / 4) > 0x7fffffffLL * 256)
		return ZEN_HIP_OK;
	*handled = true;
	switch (a.len) {
#define X(W) \
	case W: return freq ? launch_freq_guard<W>(a, rm, stream, bits_done) : launch_time<W>(a, rm, stream);
		ZH_NET_WIDTHS(X)
#undef X
	default: *handled = false; return ZEN_HIP_OK;
	}
}

// true if launch_median(a) with a.time_len = time_len, a.len = freq_len, Hermitian rows of `cols` bins and bits_t set will run
// median_tf_herm_bits_kernel (the conditions of launch_freq above)
bool median_tf_fused_available(int time_len, int freq_len, int cols)
{
	if (g_opt_no_median_tf || g_opt_median_general || cols < 256 || (cols & (cols - 1)) != 0)
		return false;
	const bool combo = (freq_len == 13 && time_len == 11) || (freq_len == 11 && time_len == 13)
	                   || ((freq_len == 23 || freq_len == 21) && time_len == 7);
	if (!combo)
		return false;
	const int T = znet::outputs_per_thread(freq_len), mid = freq_len / 2;
	const int segs = (cols + 256 * T - 1) / (256 * T);
	const int n_lo = cols / (2 * T) + 1, n_tail = (mid + T - 1) / T;
	return segs == 1 && cols % (2 * T) == 0 && n_lo + n_tail <= HERM_THREADS && 3 * (((cols >> 1) + 4) >> 2) <= HERM_THREADS;
}

// Hermitian rows (FilterArgs::hermitian): the sorting-network kernels of this file and median47_dpp_kernel
// read the stored half and skip the unwanted outputs; the long-mask kernel does so for whole 4096-column segments
bool filter_supports_hermitian(int len, int cols)
{
	if (cols % 4 != 0 || cols < 32)
		return false;
	if (len >= 3 && len <= 63)
		return true;
	// the long masks of median_big.hip (hop 2048 / 4096): whole 16-bin blocks, the two pieces of the tail apart
	const bool big = len == 65 || len == 85 || len == 93 || len == 129 || len == 171 || len == 187 || len == 255 || len == 257;
	return big && cols % 32 == 0 && (cols >> 5) >= (len / 2 + 15) / 16 + 1;
}

} // namespace zen_hip_impl
