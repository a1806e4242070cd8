// median47.hip -- the 47-tap frequency-direction median of the headline config (hop 1024: l_perc = 46,
// mask 47, libzen/hps.h:229 + libzen/mfilt.h:89), with the sorting work shared between neighbouring
// threads.  Same results as median_net_freq_kernel<47> (and as MedianFilterCPU, mfilt.h:270-342).
//
// Why a second kernel: on gfx950 v_min/v_max/v_min3/v_max3/v_med3 issue at half rate (~4 cycles per
// wave64 instruction), and the generic kernel spends 85 % of all SIMD cycles on ~52 of them per output
// (profiles/r01_c_median47_pmc.json).  Nearly half of those sort samples that the neighbouring threads
// sort as well.  With 47 taps and 16 outputs per thread every chunk the selection tree consumes is an
// aligned dyadic block of the row (mid + 1 = 24 is a multiple of 8):
//
//   B(u) = x[16u-8 .. 16u+7]                      one 16-sample block per thread u
//   thread t needs  B(t), B(t+1) sorted            (its 32 common samples)
//                   upper half / 2nd, 4th quarter of B(t-1), sorted;  B(t-1) raw (pairs, singles)
//                   lower half / 1st, 3rd quarter of B(t+2), sorted;  B(t+2) raw
//
// so each thread sorts ONE block (63 comparators), publishes it in place of its raw samples, and gets
// the 16 middle ranks of its 32 common samples from its own and its right neighbour's sorted block by
// one bitonic half-merge (48 comparator-equivalents) instead of a 191-comparator 32-sort.  The flanks
// (15 samples on either side) are read before the in-place publication and go through the same
// selection tree as the generic kernel.  ~44 min/max per output instead of ~52, same LDS footprint.
// (A variant that also shared the sorted halves/quarters needed 75 KB of LDS per workgroup and lost
// more to occupancy than it saved in instructions.)
#include "common.h"
#include "filters.h"
#include "median_net.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

constexpr int NCH = 259;  // chunks per row segment: blocks u = -1 .. 257  <->  chunk c = u + 1
constexpr int RSTR = 20;  // raw image: 16 words + 4 pad per chunk (5c mod 16 distinct: ds_read_b128 conflict free)
constexpr int OUTS = 4096;

using znet::from_key;
using znet::to_key;

// Preconditions (checked by the launcher): cols % 4 == 0, rows and pointers 16-byte aligned.
template <bool NONNEG>
__global__ __launch_bounds__(256) void median47_shared_kernel(FilterArgs a, int row_base, int ring, int segs_per_row)
{
	__shared__ __attribute__((aligned(16))) int raw[NCH * RSTR];

	const int tid = threadIdx.x;
	const int cols = a.cols;
	const long long total = (long long)a.n_out_rows * segs_per_row * a.n_streams;
	constexpr int NVEC = NCH * 4, NLD = (NVEC + 255) / 256;

	const float* srow = nullptr;
	float* drow = nullptr;
	int col0 = 0;
	auto locate = [&](long long sg) {
		const long long rowg = sg / segs_per_row;
		const int seg = (int)(sg - rowg * segs_per_row);
		const int st = (int)(rowg / a.n_out_rows), row = (int)(rowg - (long long)st * a.n_out_rows);
		col0 = seg * OUTS;
		srow = a.src + (long long)st * a.src_stream_stride + (long long)((row_base + row) % ring) * cols;
		drow = a.dst + (long long)st * a.dst_stream_stride + (long long)row * cols;
	};
	float4 x[NLD];
	auto issue_loads = [&]() {
		const int c_lo = col0 - 24; // column of raw chunk 0, word 0 (block u = -1)
#pragma unroll
		for (int i = 0; i < NLD; ++i) {
			int vc = c_lo + 4 * (tid + 256 * i);
			vc = vc < 0 ? 0 : (vc > cols - 4 ? cols - 4 : vc);
			x[i] = *reinterpret_cast<const float4*>(srow + vc);
		}
	};

	// persistent: a workgroup walks over row segments; the HBM loads of the next segment are in flight
	// while the current one is sorted.
	long long sg = blockIdx.x;
	if (sg < total) {
		locate(sg);
		issue_loads();
	}
	while (sg < total) {
		float* const drow_cur = drow;
		const int col0_cur = col0;
		{
			const int c_lo = col0 - 24;
#pragma unroll
			for (int i = 0; i < NLD; ++i) {
				const int vi = tid + 256 * i;
				const int vc = c_lo + 4 * vi;
				if (vi < NVEC) {
					int4 k = make_int4(to_key<NONNEG>(x[i].x), to_key<NONNEG>(x[i].y), to_key<NONNEG>(x[i].z),
					                   to_key<NONNEG>(x[i].w));
					if (vc < 0) // replicate border (ippBorderRepl)
						k = make_int4(k.x, k.x, k.x, k.x);
					else if (vc >= cols)
						k = make_int4(k.w, k.w, k.w, k.w);
					*reinterpret_cast<int4*>(&raw[(vi >> 2) * RSTR + 4 * (vi & 3)]) = k;
				}
			}
		}
		const long long next = sg + gridDim.x;
		if (next < total) {
			locate(next);
			issue_loads();
		}
		__syncthreads();

		// ---- own block + both flanks into registers, then the sorted block replaces the raw one
		constexpr int W = 47, T = 16, NE = W + T - 1;
		int e[NE]; // e[q] = x[16t-23+q]; only the flanks e[0..14], e[47..61] are filled
		int A[16];
		{
			int lo[16], hi[16];
			znet::lds_load<16>(&raw[tid * RSTR], lo);       // B(t-1) = x[16t-24 .. 16t-9]
			znet::lds_load<16>(&raw[(tid + 3) * RSTR], hi); // B(t+2) = x[16t+24 .. 16t+39]
#pragma unroll
			for (int q = 0; q < 15; ++q) {
				e[q] = lo[q + 1];
				e[47 + q] = hi[q];
			}
#pragma unroll
			for (int q = 15; q < 47; ++q)
				e[q] = 0; // the 32 common samples come from the sorted blocks instead
			znet::lds_load<16>(&raw[(tid + 1) * RSTR], A);
		}
		int extra[16];
		if (tid == 0)
			znet::lds_load<16>(&raw[257 * RSTR], extra); // B(256): right neighbour of the last thread
		__syncthreads();
		znet::sort_net<16>(A);
		znet::lds_store<16>(&raw[(tid + 1) * RSTR], A);
		if (tid == 0) {
			znet::sort_net<16>(extra);
			znet::lds_store<16>(&raw[257 * RSTR], extra);
		}
		__syncthreads();
		int cand[16], out[16];
		{
			int B[16];
			znet::lds_load<16>(&raw[(tid + 2) * RSTR], B);
			znet::mid16_of_two_sorted16(A, B, cand);
		}
		__syncthreads(); // sorted blocks are consumed: the image now collects the results
		znet::Node<W, T, 0, NE, T>::run(e, cand, out);
		znet::lds_store<16>(&raw[tid * RSTR], out);
		__syncthreads();

#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int g = 4 * tid + 1024 * i;
			const int c = col0_cur + g;
			if (c < cols) {
				const int4 k = *reinterpret_cast<const int4*>(&raw[(g >> 4) * RSTR + (g & 15)]);
				*reinterpret_cast<float4*>(drow_cur + c) = make_float4(from_key<NONNEG>(k.x), from_key<NONNEG>(k.y),
				                                                       from_key<NONNEG>(k.z), from_key<NONNEG>(k.w));
			}
		}
		__syncthreads(); // results are out of the image before the next segment is staged
		sg = next;
	}
}

} // namespace

// 47 taps, frequency direction, vector-aligned geometry.  *handled = false: use the generic kernel.
int launch_median47_shared(const FilterArgs& a, hipStream_t stream, bool* handled)
{
	*handled = false;
	if (a.len != 47 || a.direction != ZEN_HIP_FREQUENCY || !g_opt_median47_shared)
		return ZEN_HIP_OK;
	const bool vec_ok = (a.cols % 4 == 0) && a.cols >= 4 && ((reinterpret_cast<uintptr_t>(a.src) & 15) == 0)
	                    && ((reinterpret_cast<uintptr_t>(a.dst) & 15) == 0) && (a.src_stream_stride % 4 == 0)
	                    && (a.dst_stream_stride % 4 == 0);
	if (!vec_ok || a.ring_rows <= 0 || a.ring_rows > 0x3fffffff || a.n_out_rows > 0x3fffffff)
		return ZEN_HIP_OK;
	*handled = true;
	const int segs = (a.cols + OUTS - 1) / OUTS;
	const int row_base = (int)(a.first_row % a.ring_rows);
	long long blocks = (long long)a.n_out_rows * segs * a.n_streams;
	if (g_opt_median47_blocks > 0 && blocks > g_opt_median47_blocks)
		blocks = g_opt_median47_blocks; // persistent: a few workgroups per CU
	dim3 grid((unsigned)blocks);
	if (a.nonneg)
		hipLaunchKernelGGL((median47_shared_kernel<true>), grid, dim3(256), 0, stream, a, row_base, (int)a.ring_rows, segs);
	else
		hipLaunchKernelGGL((median47_shared_kernel<false>), grid, dim3(256), 0, stream, a, row_base, (int)a.ring_rows, segs);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace zen_hip_impl
