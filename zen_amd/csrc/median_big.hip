// median_big.hip -- frequency-direction median for the long percussive masks of hop 2048 / 4096
// (l_perc = 500 / (fs / nfft), libzen/hps.h:229: 85, 93, 171, 187 taps; 65, 129 at 16/32 kHz; 255), results identical to
// MedianFilterCPU (libzen/mfilt.h:270-342) and to the general wave kernel of median.hip.
// Algorithm: median_big.h.  One workgroup = 4096 consecutive outputs of one row; LDS holds the raw
// samples of the segment (with halo) and one sorted copy of every aligned 16-sample block.
#include "common.h"
#include "filters.h"
#include "median_big.h"

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
template <int W, bool NONNEG>
__global__ __launch_bounds__(256, ZEN_BIG_MINB) void median_big_kernel(FilterArgs a, int row_base, int ring, int segs_per_row)
{
	using G = zbig::Geo<W>;
	// raw image: chunk c = block c - (a+2) of the segment; sorted image: entry s = block s - a
	constexpr int NRAW = 256 + G::a + 2 + G::b + 2 - 1;
	constexpr int NSORT = 256 + G::NB - 1;
	__shared__ __attribute__((aligned(16))) int raw[NRAW * RSTR];
	__shared__ __attribute__((aligned(16))) int srt[NSORT * RSTR];

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
#pragma unroll
		for (int i = 0; i < NLD; ++i) {
			const int vi = tid + 256 * i;
			const int vc = c_lo + 4 * vi;
			if (vi < NVEC) {
				const int vcl = vc < 0 ? 0 : (vc > cols - 4 ? cols - 4 : vc);
				const float4 x = *reinterpret_cast<const float4*>(srow + vcl);
				int4 k = make_int4(to_key<NONNEG>(x.x), to_key<NONNEG>(x.y), to_key<NONNEG>(x.z), to_key<NONNEG>(x.w));
				if (vc < 0)
					k = make_int4(k.x, k.x, k.x, k.x);
				else if (vc >= cols)
					k = make_int4(k.w, k.w, k.w, k.w);
				*reinterpret_cast<int4*>(&raw[(vi >> 2) * RSTR + 4 * (vi & 3)]) = k;
			}
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
	// ---- 16 outputs per thread
	int out[16];
	{
		struct Loader {
			const int* srt_t; // sorted block t-a
			const int* raw_t; // raw block t-a-2
			__device__ __forceinline__ void sorted(int i, int* v) const { znet::lds_load<16>(srt_t + i * RSTR, v); }
			__device__ __forceinline__ void rawl(int j, int* v) const { znet::lds_load<16>(raw_t + j * RSTR, v); }
			__device__ __forceinline__ void rawr(int j, int* v) const { znet::lds_load<16>(raw_t + (G::a + G::b + 2 + j) * RSTR, v); }
		} ld{&srt[tid * RSTR], &raw[tid * RSTR]};
		zbig::medians_big<W>(ld, out);
	}
	__syncthreads(); // all reads of the images done: the raw image now collects the results
	znet::lds_store<16>(&raw[tid * RSTR], out);
	__syncthreads();
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int g = 4 * tid + 1024 * i;
		const int c = col0 + g;
		if (c < cols) {
			const int4 k = *reinterpret_cast<const int4*>(&raw[(g >> 4) * RSTR + (g & 15)]);
			*reinterpret_cast<float4*>(drow + c) =
			    make_float4(from_key<NONNEG>(k.x), from_key<NONNEG>(k.y), from_key<NONNEG>(k.z), from_key<NONNEG>(k.w));
		}
	}
}

template <int W>
int launch_w(const FilterArgs& a, hipStream_t stream)
{
	const int segs = (a.cols + OUTS - 1) / OUTS;
	const int row_base = (int)(a.first_row % a.ring_rows);
	dim3 grid((unsigned)((long long)a.n_out_rows * segs * a.n_streams));
	if (a.nonneg)
		hipLaunchKernelGGL((median_big_kernel<W, true>), grid, dim3(256), 0, stream, a, row_base, (int)a.ring_rows, segs);
	else
		hipLaunchKernelGGL((median_big_kernel<W, false>), grid, dim3(256), 0, stream, a, row_base, (int)a.ring_rows, segs);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

// l_perc of hops 512..4096 at the usual sample rates (8..96 kHz), and the longest mask the API accepts
bool median_big_available(int len)
{
	return len == 65 || len == 85 || len == 93 || len == 129 || len == 171 || len == 187 || len == 255;
}

} // namespace

// long frequency masks with vector-aligned geometry.  *handled = false: use the general kernel.
int launch_median_big(const FilterArgs& a, hipStream_t stream, bool* handled)
{
	*handled = false;
	if (a.direction != ZEN_HIP_FREQUENCY || !median_big_available(a.len) || a.hermitian)
		return ZEN_HIP_OK;
	const bool vec_ok = (a.cols % 4 == 0) && a.cols >= 4 && ((reinterpret_cast<uintptr_t>(a.src) & 15) == 0)
	                    && ((reinterpret_cast<uintptr_t>(a.dst) & 15) == 0) && (a.src_stream_stride % 4 == 0)
	                    && (a.dst_stream_stride % 4 == 0);
	const long long blocks = (long long)a.n_out_rows * ((a.cols + OUTS - 1) / OUTS) * a.n_streams;
	if (!vec_ok || a.ring_rows <= 0 || a.ring_rows > 0x3fffffff || blocks > 0x7fffffffLL)
		return ZEN_HIP_OK;
	*handled = true;
	switch (a.len) {
	case 65: return launch_w<65>(a, stream);
	case 85: return launch_w<85>(a, stream);
	case 93: return launch_w<93>(a, stream);
	case 129: return launch_w<129>(a, stream);
	case 171: return launch_w<171>(a, stream);
	case 187: return launch_w<187>(a, stream);
	default: return launch_w<255>(a, stream);
	}
}

} // namespace zen_hip_impl
