// median_big.hip -- frequency-direction median for the long percussive masks of hop 2048 / 4096
// (l_perc = 500 / (fs / nfft), libzen/hps.h:229: 85, 93, 171, 187 taps; 65, 129 at 16/32 kHz; 255), results identical to
// MedianFilterCPU (libzen/mfilt.h:270-342) and to the general wave kernel of median.hip.
// Algorithm: median_big.h.  One workgroup = 4096 consecutive outputs of one row; LDS holds the raw
// samples of the segment (with halo) and one sorted copy of every aligned 16-sample block.
#include "common.h"
#include "filters.h"
#include "masks.h"
#include "median_big.h"
#include "row_load.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

constexpr int RSTR = 20; // 16 words + 4 pad per block: ds_read_b128 of consecutive blocks is conflict free
constexpr int OUTS = 4096;

using znet::from_key;
using znet::to_key;

// Preconditions (checked by the launcher): cols % 4 == 0, rows and pointers 16-byte aligned.
#ifndef ZEN_BIG_MINB
#define ZEN_BIG_MINB 2
#endif
#ifndef ZEN_BIG_SHARE32_MAX_LDS
#define ZEN_BIG_SHARE32_MAX_LDS (80 * 1024 + 512) // two workgroups per CU (187 taps: 80 432 B); 255 / 257 taps (82 256 B) take it too: below
#endif
// nblk_main: the leading blocks of a row that are filtered (all of them; Hermitian rows: bins 0..cols/2 - 1, the
// rest is median_big_tail_kernel's).
// Two mask bits per bin (FilterArgs::bits) of a thread's sixteen outputs: P = the medians, H = the block's own samples
// (hps.cu:501-505, :535-540 through masks.h hard_mask_exact; non-negative samples: a key is the float's bits)
__device__ __forceinline__ unsigned mask_word16(const int (&p)[16], const int (&h)[16], double thr_p, double thr_h, int need_pm,
                                                int need_hm)
{
	unsigned w = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const float pf = __int_as_float(p[i]), hf = __int_as_float(h[i]);
		const unsigned pm = need_pm && hard_mask_exact(pf, hf + FLT_EPSILON, thr_p) != 0.0f ? 1u : 0u;
		const unsigned hm = need_hm && hard_mask_exact(hf, pf + FLT_EPSILON, thr_h) != 0.0f ? 1u : 0u;
		w |= (pm | (hm << 1)) << (2 * i);
	}
	return w;
}

// FilterArgs::soft_rows: the medians replaced by the percussive soft mask of their bins (as keys: a mask is >= +0), the
// harmonic one in mh[] (masks.h pmask_value / hmask_value: the very functions the synthesis would call)
__device__ __forceinline__ void soft_masks16(int (&p)[16], const int (&h)[16], int power, int need_pm, int need_hm, float (&mh)[16])
{
	const MaskCfg c{0.0f, 0.0f, 1, power, 0, 0, 0};
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const float pf = __int_as_float(p[i]), hf = __int_as_float(h[i]);
		mh[i] = need_hm ? hmask_value(hf, pf, c) : 0.0f;
		p[i] = __float_as_int(need_pm ? pmask_value(hf, pf, c) : 0.0f);
	}
}

// BITS (Hermitian rows of non-negative samples whose harmonic estimate is the row itself): the mask word of the
// thread's block is stored instead of its sixteen medians.
// SOFT (same rows): soft masks instead of the medians (soft_masks16).
template <int W, bool NONNEG, bool BITS = false, bool SOFT = false>
__global__ __launch_bounds__(256, ZEN_BIG_MINB) void median_big_kernel(FilterArgs a, int row_base, int ring, int segs_per_row,
                                                                      int nblk_main)
{
	static_assert(!BITS || NONNEG, "mask bits: magnitudes only");
	using G = zbig::Geo<W>;
	// raw image: chunk c = block c - (a+2) of the segment; sorted image: entry s = block s - a
	constexpr int NRAW = 256 + G::a + 2 + G::b + 2 - 1;
	constexpr int NSORT = 256 + G::NB - 1;
	__shared__ __attribute__((aligned(16))) int raw[NRAW * RSTR];
	__shared__ __attribute__((aligned(16))) int srt[NSORT * RSTR];
	// Round 6: every adjacent pair of sorted blocks merged ONCE (entry s = blocks s, s + 1 as one sorted list of 32): the first
	// level of every thread's merge tree -- BIG/2 merges of 32 per thread, of which all but one are a neighbour's -- becomes one
	// merge per thread and BIG/2 loads (36 words per entry: consecutive threads' 16-byte reads fall into different banks).
	// Eight-block trees only: 171 / 187 taps -3.5..4 % (80.4 KB of LDS: still two workgroups per CU); 255 / 257 taps 3.4-3.7 ->
	// 1.35-1.79 ms per 6 460 x 16 384 although the image (82.3 KB) leaves one workgroup per CU -- the shorter tree needs 72-190
	// bytes of scratch per lane instead of 1.1 KB; four-block trees (129 taps) lose 4 % (one merge saved, a barrier and an
	// LDS round trip added).
	constexpr int S2STR = 36;
#ifdef ZEN_BIG_NO_SHARE32 // (A/B: the kernel of rounds 2-5)
	constexpr bool SHARE32 = false;
#else
	constexpr bool SHARE32 = G::BIG >= 8 && sizeof(int) * (size_t)(NRAW * RSTR + NSORT * RSTR + (NSORT - 1) * S2STR) <= ZEN_BIG_SHARE32_MAX_LDS;
#endif
	__shared__ __attribute__((aligned(16))) int srt2[SHARE32 ? (NSORT - 1) * S2STR : 4];

	const int tid = threadIdx.x;
	const int cols = a.cols;
	const long long sg = blockIdx.x;
	const long long rowg = sg / segs_per_row;
	const int seg = (int)(sg - rowg * segs_per_row);
	const int st = (int)(rowg / a.n_out_rows), row = (int)(rowg - (long long)st * a.n_out_rows);
	const int col0 = seg * OUTS;
	const float* srow = a.src + (long long)st * a.src_stream_stride + (long long)((row_base + row) % ring) * cols;
	float* drow = a.dst + (long long)st * a.dst_stream_stride + (long long)row * cols;

	// ---- stage the segment + halo as ordering keys (replicate border, ippBorderRepl)
	{
		constexpr int NVEC = NRAW * 4, NLD = (NVEC + 255) / 256;
		const int c_lo = col0 - 16 * (G::a + 2);
		int4 kv[NLD]; // all loads in flight, then the image (a vector past the image reads a replicated column: harmless)
#pragma unroll
		for (int i = 0; i < NLD; ++i)
			kv[i] = row_vec_keys<NONNEG>(srow, c_lo + 4 * (tid + 256 * i), cols, a.hermitian);
#pragma unroll
		for (int i = 0; i < NLD; ++i) {
			const int vi = tid + 256 * i;
			if (vi < NVEC)
				*reinterpret_cast<int4*>(&raw[(vi >> 2) * RSTR + 4 * (vi & 3)]) = kv[i];
		}
	}
	__syncthreads();
	// ---- every block sorted once
	for (int s = tid; s < NSORT; s += 256) {
		int v[16];
		znet::lds_load<16>(&raw[(s + 2) * RSTR], v);
		znet::sort_net<16>(v);
		znet::lds_store<16>(&srt[s * RSTR], v);
	}
	__syncthreads();
	if constexpr (SHARE32) {
		for (int s = tid; s < NSORT - 1; s += 256) {
			int v[32];
			znet::lds_load<16>(&srt[s * RSTR], v);
			znet::lds_load<16>(&srt[(s + 1) * RSTR], v + 16);
			znet::oe_merge<32, 0>(v);
			znet::lds_store<32>(&srt2[s * S2STR], v);
		}
		__syncthreads();
	}
	// ---- 16 outputs per thread
	const bool wanted = (col0 >> 4) + tid < nblk_main;
	int out[16];
#pragma unroll
	for (int i = 0; i < 16; ++i)
		out[i] = 0;
	if (wanted) {
		struct Loader {
			const int* srt_t; // sorted block t-a
			const int* raw_t; // raw block t-a-2
			__device__ __forceinline__ void sorted(int i, int* v) const { znet::lds_load<16>(srt_t + i * RSTR, v); }
			__device__ __forceinline__ void rawl(int j, int* v) const { znet::lds_load<16>(raw_t + j * RSTR, v); }
			__device__ __forceinline__ void rawr(int j, int* v) const { znet::lds_load<16>(raw_t + (G::a + G::b + 2 + j) * RSTR, v); }
		};
		struct Loader32 : Loader {
			const int* srt2_t; // sorted pair (t-a, t-a+1)
			__device__ __forceinline__ void sorted32(int i, int* v) const { znet::lds_load<32>(srt2_t + i * S2STR, v); }
		};
		if constexpr (SHARE32) {
			Loader32 ld;
			ld.srt_t = &srt[tid * RSTR];
			ld.raw_t = &raw[tid * RSTR];
			ld.srt2_t = &srt2[tid * S2STR];
			zbig::medians_big<W>(ld, out);
		}
		else {
			Loader ld{&srt[tid * RSTR], &raw[tid * RSTR]};
			zbig::medians_big<W>(ld, out);
		}
	}
	if constexpr (BITS) {
		if (wanted) {
			int h[16];
			znet::lds_load<16>(&raw[(tid + G::a + 2) * RSTR], h); // the thread's own block
			ZH_CHK(a.bits + ((long long)st * a.bits_stream_stride + (long long)row * a.bits_row_words + (col0 >> 4) + tid), 1);
			a.bits[(long long)st * a.bits_stream_stride + (long long)row * a.bits_row_words + (col0 >> 4) + tid] =
			    mask_word16(out, h, a.thr_p, a.thr_h, a.need_pm, a.need_hm);
		}
		return;
	}
	int mhk[16]; // SOFT: the harmonic masks of the thread's bins (as keys), stored like the results below
	if constexpr (SOFT) {
#pragma unroll
		for (int i = 0; i < 16; ++i)
			mhk[i] = 0;
		if (wanted) {
			int h[16];
			float mh[16];
			znet::lds_load<16>(&raw[(tid + G::a + 2) * RSTR], h); // the thread's own block
			soft_masks16(out, h, a.soft_power, a.need_pm, a.need_hm, mh);
#pragma unroll
			for (int i = 0; i < 16; ++i)
				mhk[i] = __float_as_int(mh[i]);
		}
		if (a.mh_dst) { // (the sorted image is dead: it collects the harmonic masks the way the raw image collects the results)
			__syncthreads();
			znet::lds_store<16>(&srt[tid * RSTR], mhk);
		}
	}
	__syncthreads(); // all reads of the images done: the raw image now collects the results
	znet::lds_store<16>(&raw[tid * RSTR], out);
	__syncthreads();
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int g = 4 * tid + 1024 * i;
		const int c = col0 + g;
		if (c < cols && (c >> 4) < nblk_main) {
			const int4 k = *reinterpret_cast<const int4*>(&raw[(g >> 4) * RSTR + (g & 15)]);
			ZH_CHK(drow + c, 4);
			*reinterpret_cast<float4*>(drow + c) =
			    make_float4(from_key<NONNEG>(k.x), from_key<NONNEG>(k.y), from_key<NONNEG>(k.z), from_key<NONNEG>(k.w));
			if constexpr (SOFT) {
				if (a.mh_dst) {
					const int4 m = *reinterpret_cast<const int4*>(&srt[(g >> 4) * RSTR + (g & 15)]);
					float* mrow = a.mh_dst + (long long)st * a.mh_stream_stride + (long long)row * cols;
					ZH_CHK(mrow + c, 4);
					*reinterpret_cast<float4*>(mrow + c) =
					    make_float4(__int_as_float(m.x), __int_as_float(m.y), __int_as_float(m.z), __int_as_float(m.w));
				}
			}
		}
	}
}

// Hermitian rows: what the main kernel leaves out -- the block that holds bin cols/2 and the last ceil(mid/16)
// blocks (whose replicate border differs from the mirrored one).  A wavefront takes TAIL_ROWS rows, 16
// lanes each: lane 0 of a row's group takes the middle block, lanes 1.. the tail blocks (at most 8); both
// neighbourhoods of every row are staged and sorted side by side.  (One row per wavefront left 57 of 64 lanes idle
// through the selection: 0.33 ms of the 12 ms offline batch step.)
constexpr int TAIL_ROWS = 4;
template <int W, bool NONNEG, bool BITS = false, bool SOFT = false>
__global__ __launch_bounds__(64) void median_big_tail_kernel(FilterArgs a, int row_base, int ring)
{
	using G = zbig::Geo<W>;
	constexpr int NT = (G::m + 15) / 16;                // tail blocks
	static_assert(NT + 1 <= 16, "a row's blocks fit its 16 lanes");
	constexpr int HALO_L = G::a + 2, HALO_R = G::b + 1;  // raw blocks needed left / right of an output block
	constexpr int NRAW_A = HALO_L + 1 + HALO_R, NRAW_B = HALO_L + NT + HALO_R, NRAW = NRAW_A + NRAW_B;
	constexpr int NSORT_A = 1 + G::NB - 1, NSORT_B = NT + G::NB - 1, NSORT = NSORT_A + NSORT_B;
	__shared__ __attribute__((aligned(16))) int raw[TAIL_ROWS * NRAW * RSTR];
	__shared__ __attribute__((aligned(16))) int srt[TAIL_ROWS * NSORT * RSTR];
	constexpr int TBW = 12; // BITS: the words of a row behind the main kernel's (entries nfft/2 .. nfft/2 + mid, at most 9 + padding)
	__shared__ unsigned tb[TAIL_ROWS * TBW];
	const int lane = threadIdx.x;
	if (BITS && lane < TAIL_ROWS * TBW)
		tb[lane] = 0u;
	const int cols = a.cols, nblk = cols >> 4;
	const int st = blockIdx.y, row0 = blockIdx.x * TAIL_ROWS;
	const int rows_here = a.n_out_rows - row0 < TAIL_ROWS ? a.n_out_rows - row0 : TAIL_ROWS;
	const float* src_s = a.src + (long long)st * a.src_stream_stride;
	const int blkA = nblk >> 1, blkB = nblk - NT; // first output block of either piece
	for (int w0 = lane; w0 < rows_here * NRAW * 4; w0 += 256) { // four vectors per turn, their loads in flight together
		int4 kv[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int wi = w0 + 64 * u < rows_here * NRAW * 4 ? w0 + 64 * u : rows_here * NRAW * 4 - 1; // (past the end: loaded again, not stored)
			const int rr = wi / (NRAW * 4), vi = wi - rr * (NRAW * 4);
			const float* srow = src_s + (long long)((row_base + row0 + rr) % ring) * cols;
			const bool pb = vi >= NRAW_A * 4;
			const int v = pb ? vi - NRAW_A * 4 : vi;
			kv[u] = row_vec_keys<NONNEG>(srow, 16 * ((pb ? blkB : blkA) - HALO_L) + 4 * v, cols, 1);
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int wi = w0 + 64 * u;
			if (wi < rows_here * NRAW * 4) {
				const int rr = wi / (NRAW * 4), vi = wi - rr * (NRAW * 4);
				*reinterpret_cast<int4*>(&raw[(rr * NRAW + (vi >> 2)) * RSTR + 4 * (vi & 3)]) = kv[u];
			}
		}
	}
	__syncthreads();
	for (int ws = lane; ws < rows_here * NSORT; ws += 64) { // sorted entry s of a piece = raw chunk s + 2 of that piece
		const int rr = ws / NSORT, s = ws - rr * NSORT;
		const bool pb = s >= NSORT_A;
		const int chunk = pb ? NRAW_A + (s - NSORT_A) + 2 : s + 2;
		int v[16];
		znet::lds_load<16>(&raw[(rr * NRAW + chunk) * RSTR], v);
		znet::sort_net<16>(v);
		znet::lds_store<16>(&srt[(rr * NSORT + s) * RSTR], v);
	}
	__syncthreads();
	const int rr = lane >> 4, l = lane & 15;
	if (!BITS && (rr >= rows_here || l > NT))
		return;
	const bool work = rr < rows_here && l <= NT;
	const int loc = l == 0 ? 0 : (l <= NT ? l - 1 : 0); // block within the piece
	struct Loader {
		const int* srt_t;
		const int* raw_t;
		__device__ __forceinline__ void sorted(int i, int* v) const { znet::lds_load<16>(srt_t + i * RSTR, v); }
		__device__ __forceinline__ void rawl(int j, int* v) const { znet::lds_load<16>(raw_t + j * RSTR, v); }
		__device__ __forceinline__ void rawr(int j, int* v) const { znet::lds_load<16>(raw_t + (G::a + G::b + 2 + j) * RSTR, v); }
	} ld{&srt[(rr * NSORT + (l == 0 ? 0 : NSORT_A) + loc) * RSTR], &raw[(rr * NRAW + (l == 0 ? 0 : NRAW_A) + loc) * RSTR]};
	int out[16];
	if (!BITS || work)
		zbig::medians_big<W>(ld, out);
	const int c = 16 * ((l == 0 ? blkA : blkB) + loc);
	if constexpr (BITS) {
		// entry of bin idx: nfft/2 for the middle bin, nfft/2 + 1 + idx - (nfft - mid) for the last mid bins; their H is the
		// block's own (mirrored) samples.  The lanes of a row OR their bits into the row's words in LDS.
		if (work) {
			int h[16];
			znet::lds_load<16>(ld.raw_t + (G::a + 2) * RSTR, h);
			const unsigned w = mask_word16(out, h, a.thr_p, a.thr_h, a.need_pm, a.need_hm);
			const int e0 = (cols >> 1) - 16 * blkA; // entries relative to the first word behind the main kernel's
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int idx = c + i;
				const int e = l == 0 ? (i == 0 ? e0 : -1) : (idx >= cols - G::m ? e0 + 1 + idx - (cols - G::m) : -1);
				if (e >= 0)
					atomicOr(&tb[rr * TBW + (e >> 4)], ((w >> (2 * i)) & 3u) << (2 * (e & 15)));
			}
		}
		__syncthreads();
		const int nw = a.bits_row_words - blkA; // words behind the main kernel's (the row's padding included)
		for (int k = lane; k < rows_here * nw; k += 64) {
			const int r2 = k / nw, w2 = k - r2 * nw;
			ZH_CHK(a.bits + ((long long)st * a.bits_stream_stride + (long long)(row0 + r2) * a.bits_row_words + blkA + w2), 1);
			a.bits[(long long)st * a.bits_stream_stride + (long long)(row0 + r2) * a.bits_row_words + blkA + w2] = tb[r2 * TBW + w2];
		}
		return;
	}
	if constexpr (SOFT) {
		int h[16];
		float mh[16];
		znet::lds_load<16>(ld.raw_t + (G::a + 2) * RSTR, h); // the block's own samples (the tail's: mirrored = H of those bins)
		soft_masks16(out, h, a.soft_power, a.need_pm, a.need_hm, mh);
		if (a.mh_dst) {
			float* mrow = a.mh_dst + (long long)st * a.mh_stream_stride + (long long)(row0 + rr) * cols + c;
#pragma unroll
			for (int v = 0; v < 4; ++v) {
				ZH_CHK(mrow + 4 * v, 4);
				*reinterpret_cast<float4*>(mrow + 4 * v) = make_float4(mh[4 * v], mh[4 * v + 1], mh[4 * v + 2], mh[4 * v + 3]);
			}
		}
	}
	float* drow = a.dst + (long long)st * a.dst_stream_stride + (long long)(row0 + rr) * cols;
#pragma unroll
	for (int v = 0; v < 4; ++v) {
		ZH_CHK(drow + c + 4 * v, 4);
		*reinterpret_cast<float4*>(drow + c + 4 * v) =
		    make_float4(from_key<NONNEG>(out[4 * v]), from_key<NONNEG>(out[4 * v + 1]), from_key<NONNEG>(out[4 * v + 2]),
		                from_key<NONNEG>(out[4 * v + 3]));
	}
}

template <int W>
int launch_w(const FilterArgs& a, hipStream_t stream, int* bits_done)
{
	const int row_base = (int)(a.first_row % a.ring_rows);
	const int nblk_main = a.hermitian ? a.cols >> 5 : (a.cols + 15) >> 4; // blocks 0 .. cols/32 - 1 hold bins 0 .. cols/2 - 1
	const int segs = (nblk_main + 255) / 256;
	dim3 grid((unsigned)((long long)a.n_out_rows * segs * a.n_streams));
	// mask bits instead of P: Hermitian magnitude rows that are their own harmonic estimate
	if (bits_done && a.bits && a.hermitian && a.nonneg && !a.hrows && a.bits_row_words - nblk_main <= 12) {
		dim3 tgrid((unsigned)((a.n_out_rows + TAIL_ROWS - 1) / TAIL_ROWS), (unsigned)a.n_streams);
		hipLaunchKernelGGL((median_big_kernel<W, true, true>), grid, dim3(256), 0, stream, a, row_base, (int)a.ring_rows, segs, nblk_main);
		hipLaunchKernelGGL((median_big_tail_kernel<W, true, true>), tgrid, dim3(64), 0, stream, a, row_base, (int)a.ring_rows);
		ZH_HIP(hipGetLastError());
		*bits_done = 1;
		return ZEN_HIP_OK;
	}
	// soft masks instead of P (same rows)
	if (bits_done && a.soft_rows && a.hermitian && a.nonneg && !a.hrows) {
		dim3 tgrid((unsigned)((a.n_out_rows + TAIL_ROWS - 1) / TAIL_ROWS), (unsigned)a.n_streams);
		hipLaunchKernelGGL((median_big_kernel<W, true, false, true>), grid, dim3(256), 0, stream, a, row_base, (int)a.ring_rows, segs, nblk_main);
		hipLaunchKernelGGL((median_big_tail_kernel<W, true, false, true>), tgrid, dim3(64), 0, stream, a, row_base, (int)a.ring_rows);
		ZH_HIP(hipGetLastError());
		*bits_done = 3;
		return ZEN_HIP_OK;
	}
	if (a.nonneg)
		hipLaunchKernelGGL((median_big_kernel<W, true>), grid, dim3(256), 0, stream, a, row_base, (int)a.ring_rows, segs, nblk_main);
	else
		hipLaunchKernelGGL((median_big_kernel<W, false>), grid, dim3(256), 0, stream, a, row_base, (int)a.ring_rows, segs, nblk_main);
	ZH_HIP(hipGetLastError());
	if (a.hermitian) {
		dim3 tgrid((unsigned)((a.n_out_rows + TAIL_ROWS - 1) / TAIL_ROWS), (unsigned)a.n_streams);
		if (a.nonneg)
			hipLaunchKernelGGL((median_big_tail_kernel<W, true>), tgrid, dim3(64), 0, stream, a, row_base, (int)a.ring_rows);
		else
			hipLaunchKernelGGL((median_big_tail_kernel<W, false>), tgrid, dim3(64), 0, stream, a, row_base, (int)a.ring_rows);
		ZH_HIP(hipGetLastError());
	}
	return ZEN_HIP_OK;
}

// l_perc of hops 512..4096 at the usual sample rates (8..96 kHz); 255; and 257 = l_perc 256 made odd: fs / hop = 7.8125, the
// lowest rate every hop runs at (32 kHz at hop 4096 -- the CLI's default hop_h --, 16 kHz at 2048, 8 kHz at 1024)
bool median_big_available(int len)
{
	return len == 65 || len == 85 || len == 93 || len == 129 || len == 171 || len == 187 || len == 255 || len == 257;
}

} // namespace

// long frequency masks with vector-aligned geometry.  *handled = false: use the general kernel.
int launch_median_big(const FilterArgs& a, hipStream_t stream, bool* handled, int* bits_done)
{
	*handled = false;
	if (a.direction != ZEN_HIP_FREQUENCY || !median_big_available(a.len))
		return ZEN_HIP_OK;
	// Hermitian rows: the main kernel takes whole 16-bin blocks below cols/2, the tail kernel one wavefront of
	// blocks: rows of at least 64 blocks whose two pieces do not overlap
	if (a.hermitian && (a.cols % 32 != 0 || (a.cols >> 5) < (a.len / 2 + 15) / 16 + 1 || a.n_streams > 65535))
		return ZEN_HIP_OK;
	const bool vec_ok = (a.cols % 4 == 0) && a.cols >= 4 && ((reinterpret_cast<uintptr_t>(a.src) & 15) == 0)
	                    && ((reinterpret_cast<uintptr_t>(a.dst) & 15) == 0) && (a.src_stream_stride % 4 == 0)
	                    && (a.dst_stream_stride % 4 == 0);
	const long long blocks = (long long)a.n_out_rows * ((a.cols + OUTS - 1) / OUTS) * a.n_streams;
	if (!vec_ok || a.ring_rows <= 0 || a.ring_rows > 0x3fffffff || blocks > 0x7fffffffLL)
		return ZEN_HIP_OK;
	*handled = true;
	switch (a.len) {
	case 65: return launch_w<65>(a, stream, bits_done);
	case 85: return launch_w<85>(a, stream, bits_done);
	case 93: return launch_w<93>(a, stream, bits_done);
	case 129: return launch_w<129>(a, stream, bits_done);
	case 171: return launch_w<171>(a, stream, bits_done);
	case 187: return launch_w<187>(a, stream, bits_done);
	case 257: return launch_w<257>(a, stream, bits_done);
	default: return launch_w<255>(a, stream, bits_done);
	}
}

} // namespace zen_hip_impl
