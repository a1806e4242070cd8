// box.hip -- 1-D box (mean) filter along either axis with replicate border, for gfx950.
// Semantics: BoxFilterCPU (libzen/box.h:217-288: ippiFilterBoxBorder_32f_C1R, ippBorderRepl); replaces
// the NPP calls of BoxFilterGPU::filter (libzen/box.h:188-214).  Used by the SSE variant of the engine
// (libzen/hps.cu:582-652), whose element-wise wrappers (|S|^2 -> 1/x before, (l+1)/x after) are fused
// here as sse_pre / sse_post.
//
// The sum is taken tap by tap in ascending index order, in float, then divided by the mask length:
// the same order as oracle/zen_oracle.c zo_box_filter, so results are bit-identical (a running sum
// would be faster and differently rounded).  Both tiled kernels stage their source samples in LDS once,
// with sse_pre already applied (one division per sample instead of one per tap), and every thread adds
// its taps from there:
//   frequency : a workgroup owns 1024 consecutive outputs of one row (+ the mask halo);
//   time      : a workgroup owns 64 columns x RB rows (+ the halo rows); lanes are consecutive columns.
// Masks too long for the time tile fall back to the direct kernel.
#include "common.h"
#include "exact_div.h"
#include "filters.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

// K values at once: the divisions of the SSE path are IEEE-exact in three instructions each where their operands are in range
// (exact_div.h: proven exhaustively); a wavefront that holds a zero, an infinity or a value next to the denormals takes the
// compiler's division for the batch.
template <int K>
__device__ __forceinline__ void pre_of(const float (&v)[K], int sse_pre, float (&out)[K])
{
	if (sse_pre) { // hps.h:91-98 powf(abs,2) then hps.h:45-56 (1/x)*1
		float sq[K], z[K];
#pragma unroll
		for (int i = 0; i < K; ++i)
			sq[i] = v[i] * v[i];
		zdiv::recip_batch<K>(sq, z);
#pragma unroll
		for (int i = 0; i < K; ++i)
			out[i] = z[i] * 1.0F;
	}
	else {
#pragma unroll
		for (int i = 0; i < K; ++i)
			out[i] = v[i];
	}
}
template <int K>
__device__ __forceinline__ void post_of(const float (&acc)[K], float flen, float rlen, int sse_post, float post_factor, float (&out)[K])
{
	float res[K];
	zdiv::div_const_batch<K>(acc, flen, rlen, res); // the box mean: sum / length (box.h:266-286)
	if (sse_post) { // hps.cu:599-604
		float z[K];
		zdiv::recip_batch<K>(res, z);
#pragma unroll
		for (int i = 0; i < K; ++i)
			out[i] = z[i] * post_factor;
	}
	else {
#pragma unroll
		for (int i = 0; i < K; ++i)
			out[i] = res[i];
	}
}
__device__ __forceinline__ float pre_of(float v, int sse_pre)
{
	return sse_pre ? (1.0f / (v * v)) * 1.0F : v; // (the direct-form kernel: one value at a time, the compiler's division)
}
__device__ __forceinline__ float post_of(float acc, float flen, int sse_post, float post_factor)
{
	const float res = acc / flen;
	return sse_post ? (1.0f / res) * post_factor : res;
}

constexpr int FREQ_OUTS = 1024;   // outputs per workgroup (4 per thread)
constexpr int MAX_LEN = 255;

__global__ __launch_bounds__(256) void box_freq_kernel(FilterArgs a, int segs_per_row)
{
	__shared__ float tile[FREQ_OUTS + MAX_LEN - 1];
	const int tid = threadIdx.x;
	const int row = blockIdx.x / segs_per_row, seg = blockIdx.x - row * segs_per_row;
	const int cols = a.cols, len = a.len, mid = len >> 1, col0 = seg * FREQ_OUTS;
	const long long ring_row = (a.first_row + row) % a.ring_rows; // wave-uniform
	const float* __restrict__ srow = a.src + (long long)blockIdx.y * a.src_stream_stride + ring_row * cols;
	float* __restrict__ drow = a.dst + (long long)blockIdx.y * a.dst_stream_stride + (long long)row * cols;
	const int span = FREQ_OUTS + len - 1;
	{ // all staging loads in flight together (span <= FREQ_OUTS + MAX_LEN - 1: five turns of 256)
		constexpr int NT = (FREQ_OUTS + MAX_LEN - 1 + 255) / 256;
		float v[NT];
#pragma unroll
		for (int u = 0; u < NT; ++u) {
			int c = col0 - mid + tid + 256 * u;
			c = c < 0 ? 0 : (c > cols - 1 ? cols - 1 : c); // replicate border (ippBorderRepl); past the span: not stored
			ZH_CHK(srow + c, 1);
			v[u] = srow[c];
		}
		float pv[NT];
		pre_of<NT>(v, a.sse_pre, pv);
#pragma unroll
		for (int u = 0; u < NT; ++u)
			if (tid + 256 * u < span)
				tile[tid + 256 * u] = pv[u];
	}
	__syncthreads();
	const float flen = (float)len;
	// the thread's four outputs tap by tap (ascending order each): four reads in flight, then the four sums -- one output
	// after the other was one dependent LDS round trip per tap.  (Lanes read consecutive words: no bank conflicts; an output
	// past the row reads inside the tile and is not stored.)
	constexpr int K = FREQ_OUTS / 256;
	float acc[K];
#pragma unroll
	for (int k = 0; k < K; ++k)
		acc[k] = tile[tid + 256 * k];
	for (int j = 1; j < len; ++j) {
		float t[K];
#pragma unroll
		for (int k = 0; k < K; ++k)
			t[k] = tile[tid + 256 * k + j];
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int k = 0; k < K; ++k)
			acc[k] = acc[k] + t[k];
		__builtin_amdgcn_sched_barrier(0);
	}
	float res[K];
	post_of<K>(acc, flen, 1.0f / flen, a.sse_post, a.post_factor, res);
#pragma unroll
	for (int k = 0; k < K; ++k) {
		const int o = tid + 256 * k;
		if (col0 + o < cols) {
			ZH_CHK(drow + col0 + o, 1);
			drow[col0 + o] = res[k];
		}
	}
}

// masks beyond MAX_LEN taps (any length the LDS holds): the same tile, sized by the launch, staged in a loop; the mean by the
// compiler's IEEE division (zdiv::div_const_batch is proven for lengths 1..255 only, tools/check_div.hip)
__global__ __launch_bounds__(256) void box_freq_long_kernel(FilterArgs a, int segs_per_row)
{
	extern __shared__ float ltile[];
	const int tid = threadIdx.x;
	const int row = blockIdx.x / segs_per_row, seg = blockIdx.x - row * segs_per_row;
	const int cols = a.cols, len = a.len, mid = len >> 1, col0 = seg * FREQ_OUTS;
	const long long ring_row = (a.first_row + row) % a.ring_rows;
	const float* __restrict__ srow = a.src + (long long)blockIdx.y * a.src_stream_stride + ring_row * cols;
	float* __restrict__ drow = a.dst + (long long)blockIdx.y * a.dst_stream_stride + (long long)row * cols;
	const int span = FREQ_OUTS + len - 1;
	for (int u = tid; u < span; u += 256) {
		int c = col0 - mid + u;
		c = c < 0 ? 0 : (c > cols - 1 ? cols - 1 : c);
		ZH_CHK(srow + c, 1);
		ltile[u] = pre_of(srow[c], a.sse_pre);
	}
	__syncthreads();
	const float flen = (float)len;
	constexpr int K = FREQ_OUTS / 256;
	float acc[K];
#pragma unroll
	for (int k = 0; k < K; ++k)
		acc[k] = ltile[tid + 256 * k];
	for (int j = 1; j < len; ++j) {
		float t[K];
#pragma unroll
		for (int k = 0; k < K; ++k)
			t[k] = ltile[tid + 256 * k + j];
#pragma unroll
		for (int k = 0; k < K; ++k)
			acc[k] = acc[k] + t[k];
	}
#pragma unroll
	for (int k = 0; k < K; ++k) {
		const int o = tid + 256 * k;
		if (col0 + o < cols) {
			ZH_CHK(drow + col0 + o, 1);
			drow[col0 + o] = post_of(acc[k], flen, a.sse_post, a.post_factor);
		}
	}
}

constexpr int TIME_COLS = 64;
constexpr int TIME_TILE_ROWS = 192; // 48 KB of LDS

// rows_per_block output rows x 64 columns per workgroup; tile row i holds absolute source row
// clamp(first + i - mid) of the block's first output row `first`.
__global__ __launch_bounds__(256) void box_time_kernel(FilterArgs a, int rows_per_block)
{
	__shared__ float tile[TIME_TILE_ROWS * TIME_COLS];
	const int tid = threadIdx.x, lane_col = tid & (TIME_COLS - 1), rl = tid >> 6;
	const int cols = a.cols, len = a.len, mid = len >> 1;
	const int col = blockIdx.x * TIME_COLS + lane_col;
	const int q0 = blockIdx.y * rows_per_block; // first output row of the block
	const int nq = min(rows_per_block, a.n_out_rows - q0);
	const float* __restrict__ src = a.src + (long long)blockIdx.z * a.src_stream_stride;
	float* __restrict__ dst = a.dst + (long long)blockIdx.z * a.dst_stream_stride;
	const long long ar0 = a.first_row + q0;
	const int nrows = nq + len - 1;
	if (col < cols) {
		// eight rows per turn, their loads in flight together (one row per turn was a trip to memory per row: up to 36 in
		// a row); a row past the tile reads the tile's last row again and is not stored
		for (int i0 = rl; i0 < nrows; i0 += 32) {
			float v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const int i = i0 + 4 * u < nrows ? i0 + 4 * u : nrows - 1;
				long long r = ar0 - mid + i;
				r = r < a.clamp_lo ? a.clamp_lo : (r > a.clamp_hi ? a.clamp_hi : r);
				ZH_CHK(src + ((r % a.ring_rows) * cols + col), 1);
				v[u] = src[(r % a.ring_rows) * cols + col];
			}
			float pv[8];
			pre_of<8>(v, a.sse_pre, pv);
#pragma unroll
			for (int u = 0; u < 8; ++u)
				if (i0 + 4 * u < nrows)
					tile[(i0 + 4 * u) * TIME_COLS + lane_col] = pv[u];
		}
	}
	__syncthreads();
	if (col >= cols)
		return;
	const float flen = (float)len;
	// causal_self: taps beyond the output's own row are that row again (hps.h:265-268): tile row q + mid
	const int jmax = a.causal_self ? mid : len - 1;
	// four output rows of the thread at a time, tap by tap (ascending order each): four reads in flight, then the four sums
	for (int qb = rl; qb < nq; qb += 16) {
		int q[4];
		float acc[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			q[u] = qb + 4 * u < nq ? qb + 4 * u : nq - 1; // (a row past the block: the last one again, not stored)
			acc[u] = tile[q[u] * TIME_COLS + lane_col];
		}
		for (int j = 1; j < len; ++j) {
			const int jj = j < jmax ? j : jmax;
			float t[4];
#pragma unroll
			for (int u = 0; u < 4; ++u)
				t[u] = tile[(q[u] + jj) * TIME_COLS + lane_col];
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int u = 0; u < 4; ++u)
				acc[u] = acc[u] + t[u];
			__builtin_amdgcn_sched_barrier(0);
		}
		float res[4];
		post_of<4>(acc, flen, 1.0f / flen, a.sse_post, a.post_factor, res);
#pragma unroll
		for (int u = 0; u < 4; ++u)
			if (qb + 4 * u < nq) {
				ZH_CHK(dst + ((long long)(q0 + q[u]) * cols + col), 1);
				dst[(long long)(q0 + q[u]) * cols + col] = res[u];
			}
	}
}

// direct form: every output gathers its taps from global memory (any mask length)
template <int DIR>
__global__ __launch_bounds__(256) void box_direct_kernel(FilterArgs a)
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
				ZH_CHK(src + ((ar % a.ring_rows) * cols + c), 1);
				v = src[(ar % a.ring_rows) * cols + c];
			}
			else {
				long long r = ar - mid + j;
				r = r < a.clamp_lo ? a.clamp_lo : (r > a.clamp_hi ? a.clamp_hi : r);
				if (a.causal_self && r > ar)
					r = ar;
				ZH_CHK(src + ((r % a.ring_rows) * cols + col), 1);
				v = src[(r % a.ring_rows) * cols + col];
			}
			v = pre_of(v, a.sse_pre);
			acc = (j == 0) ? v : acc + v;
		}
		ZH_CHK(dst + i, 1);
		dst[i] = post_of(acc, flen, a.sse_post, a.post_factor);
	}
}

} // namespace

int launch_box(const FilterArgs& a, hipStream_t stream)
{
	if (a.n_out_rows <= 0 || a.cols <= 0 || a.n_streams <= 0)
		return ZEN_HIP_OK;
	if (a.len < 1 || !(a.len & 1))
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "box: mask length %d must be odd and positive", a.len);
	if (a.direction == ZEN_HIP_FREQUENCY && a.len <= MAX_LEN) {
		const int segs = (a.cols + FREQ_OUTS - 1) / FREQ_OUTS;
		const long long blocks = (long long)a.n_out_rows * segs;
		if (blocks <= 0x7fffffffLL) {
			hipLaunchKernelGGL(box_freq_kernel, dim3((unsigned)blocks, (unsigned)a.n_streams), dim3(256), 0, stream, a, segs);
			ZH_HIP(hipGetLastError());
			return ZEN_HIP_OK;
		}
	}
	if (a.direction == ZEN_HIP_FREQUENCY && a.len > MAX_LEN && sizeof(float) * (size_t)(FREQ_OUTS + a.len - 1) <= 150 * 1024) {
		const int segs = (a.cols + FREQ_OUTS - 1) / FREQ_OUTS;
		const long long blocks = (long long)a.n_out_rows * segs;
		const size_t lds = sizeof(float) * (size_t)(FREQ_OUTS + a.len - 1);
		if (blocks <= 0x7fffffffLL) {
			if (lds > 64 * 1024)
				ZH_HIP(hipFuncSetAttribute((const void*)box_freq_long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
			hipLaunchKernelGGL(box_freq_long_kernel, dim3((unsigned)blocks, (unsigned)a.n_streams), dim3(256), lds, stream, a, segs);
			ZH_HIP(hipGetLastError());
			return ZEN_HIP_OK;
		}
	}
	if (a.direction != ZEN_HIP_FREQUENCY && a.len - 1 + 16 <= TIME_TILE_ROWS) {
		int rpb = TIME_TILE_ROWS - (a.len - 1);
		if (rpb > 128)
			rpb = 128;
		const int col_blocks = (a.cols + TIME_COLS - 1) / TIME_COLS;
		// enough workgroups to fill the chip when the matrix is small
		while (rpb > 16 && (long long)col_blocks * ((a.n_out_rows + rpb - 1) / rpb) * a.n_streams < 1024)
			rpb >>= 1;
		const long long row_blocks = (a.n_out_rows + rpb - 1) / rpb;
		if (row_blocks <= 65535 && a.n_streams <= 65535) {
			hipLaunchKernelGGL(box_time_kernel, dim3((unsigned)col_blocks, (unsigned)row_blocks, (unsigned)a.n_streams),
			                   dim3(256), 0, stream, a, rpb);
			ZH_HIP(hipGetLastError());
			return ZEN_HIP_OK;
		}
	}
	const long long n = (long long)a.n_out_rows * a.cols;
	long long blocks = (n + 255) / 256;
	if (blocks > 8192)
		blocks = 8192;
	dim3 grid((unsigned)blocks, (unsigned)a.n_streams);
	if (a.direction == ZEN_HIP_FREQUENCY)
		hipLaunchKernelGGL(box_direct_kernel<0>, grid, dim3(256), 0, stream, a);
	else
		hipLaunchKernelGGL(box_direct_kernel<1>, grid, dim3(256), 0, stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace zen_hip_impl
