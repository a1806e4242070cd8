// box.hip -- 1-D box (mean) filter along either axis with replicate border, for gfx950.
// Semantics: BoxFilterCPU (libzen/box.h:217-288: ippiFilterBoxBorder_32f_C1R, ippBorderRepl); replaces
// the NPP calls of BoxFilterGPU::filter (libzen/box.h:188-214).  Used by the SSE variant of the engine
// (libzen/hps.cu:582-652), whose element-wise wrappers (|S|^2 -> 1/x before, (l+1)/x after) are fused
// here as sse_pre / sse_post.
//
// The sum is taken tap by tap in ascending index order, in float, then divided by the mask length:
// the same order as oracle/zen_oracle.c zo_box_filter, so results are bit-identical (a running sum
// would be faster and differently rounded).  Neighbouring threads read neighbouring addresses, the
// taps of a thread are served by L1/L2; masks are <= 23 taps on the BASELINE configs.
#include "common.h"
#include "filters.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

template <int DIR>
__global__ __launch_bounds__(256) void box_kernel(FilterArgs a)
{
	const float* __restrict__ src = a.src + (long long)blockIdx.y * a.src_stream_stride;
	float* __restrict__ dst = a.dst + (long long)blockIdx.y * a.dst_stream_stride;
	const int cols = a.cols, len = a.len, mid = len >> 1;
	const float flen = (float)len;
	const long long n = (long long)a.n_out_rows * cols;
	for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
	     i += (long long)gridDim.x * blockDim.x) {
		const long long row = i / cols;
		const int col = (int)(i - row * cols);
		const long long ar = a.first_row + row;
		float acc = 0.f;
		for (int j = 0; j < len; ++j) {
			float v;
			if (DIR == 0) {
				int c = col - mid + j;
				c = c < 0 ? 0 : (c > cols - 1 ? cols - 1 : c);
				v = src[(ar % a.ring_rows) * cols + c];
			}
			else {
				long long r = ar - mid + j;
				r = r < a.clamp_lo ? a.clamp_lo : (r > a.clamp_hi ? a.clamp_hi : r);
				if (a.causal_self && r > ar)
					r = ar;
				v = src[(r % a.ring_rows) * cols + col];
			}
			if (a.sse_pre)
				v = (1.0f / (v * v)) * 1.0F; // hps.h:91-98 powf(abs,2) then hps.h:45-56 (1/x)*1
			acc = (j == 0) ? v : acc + v;
		}
		float res = acc / flen;
		if (a.sse_post)
			res = (1.0f / res) * a.post_factor; // hps.cu:599-604
		dst[i] = res;
	}
}

} // namespace

int launch_box(const FilterArgs& a, hipStream_t stream)
{
	if (a.n_out_rows <= 0 || a.cols <= 0 || a.n_streams <= 0)
		return ZEN_HIP_OK;
	if (a.len < 1 || !(a.len & 1))
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "box: mask length %d must be odd and positive", a.len);
	const long long n = (long long)a.n_out_rows * a.cols;
	long long blocks = (n + 255) / 256;
	if (blocks > 8192)
		blocks = 8192;
	dim3 grid((unsigned)blocks, (unsigned)a.n_streams);
	if (a.direction == ZEN_HIP_FREQUENCY)
		hipLaunchKernelGGL(box_kernel<0>, grid, dim3(256), 0, stream, a);
	else
		hipLaunchKernelGGL(box_kernel<1>, grid, dim3(256), 0, stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace zen_hip_impl
