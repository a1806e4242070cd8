// median.hip -- exact 1-D sliding median along either axis of a (rows x cols) float matrix, replicate
// border, for gfx950.  Semantics: MedianFilterCPU (libzen/mfilt.h:270-342), the reference's parity
// target; replaces the NPP calls of MedianFilterGPU::filter (libzen/mfilt.h:233-267).
//
// Kernel in this file: the general wave-cooperative sliding window (any odd length up to 255, both
// directions).  One 64-lane wavefront keeps the current window as a SORTED array spread over its lanes
// (position p lives in register p/64, lane p%64) and slides it one sample at a time: the leaving and the
// entering sample are located with two wave-wide compares + ballots (popcount = rank, the array is
// sorted so each ballot is a prefix mask), the elements between the two ranks move one lane up or down
// (wave-wide DPP shift + a select under a scalar lane mask),
// and the middle position is the output.  No data-dependent memory traffic: the line being filtered is
// staged once into LDS with coalesced loads, results go back through LDS so that stores are coalesced
// in both directions.  Values are ordered through the usual monotone float->int key, so the result is
// the bit pattern of an input sample (exact order statistic, as IPP's).
#include "common.h"
#include "filters.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

constexpr int KEY_MAX = 0x7fffffff;

__device__ __forceinline__ int f2key(float f)
{
	int b = __float_as_int(f);
	return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

__device__ __forceinline__ long long clampll(long long v, long long lo, long long hi)
{
	return v < lo ? lo : (v > hi ? hi : v);
}

// Sorted window of up to 64*R keys held by one wavefront: position p lives in register p/64, lane p%64.
// One slide step = two rank computations (v_cmp + s_bcnt1 per register), then per register one
// wave-wide DPP shift (wave_shl:1 / wave_shr:1, lane i <- lane i+1 / i-1, verified on gfx950 by
// tools/probe_dpp.hip), one range test and two selects.
template <int R>
struct SortedWindow {
	int s[R];

	__device__ __forceinline__ void init()
	{
#pragma unroll
		for (int r = 0; r < R; ++r)
			s[r] = KEY_MAX;
	}

	// remove one instance of key vo (must be present), insert key vi; vo, vi wave-uniform
	__device__ __forceinline__ void replace(int vo, int vi, int lane)
	{
		vo = __builtin_amdgcn_readfirstlane(vo);
		vi = __builtin_amdgcn_readfirstlane(vi);
		if (vi == vo)
			return;
		int p_out = 0, c_in = 0; // #keys < vo (= position of the first vo), #keys < vi
#pragma unroll
		for (int r = 0; r < R; ++r) {
			p_out += __popcll(__ballot(s[r] < vo));
			c_in += __popcll(__ballot(s[r] < vi));
		}
		if (vi > vo) { // positions [p_out, c_in-1) take their upper neighbour, vi lands at c_in-1
			const int lo = p_out, hi = c_in - 1;
#pragma unroll
			for (int r = 0; r < R; ++r) {
				const int carry = (r + 1 < R) ? __builtin_amdgcn_readlane(s[r + 1], 0) : 0;
				const int nb = __builtin_amdgcn_update_dpp(carry, s[r], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
				const int p = r * 64 + lane;
				const int t = ((unsigned)(p - lo) < (unsigned)(hi - lo)) ? nb : s[r];
				s[r] = (p == hi) ? vi : t;
			}
		}
		else { // vi lands at c_in, positions (c_in, p_out] take their lower neighbour
			const int lo = c_in, hi = p_out;
#pragma unroll
			for (int r = R - 1; r >= 0; --r) {
				const int carry = (r > 0) ? __builtin_amdgcn_readlane(s[r - 1], 63) : 0;
				const int nb = __builtin_amdgcn_update_dpp(carry, s[r], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
				const int p = r * 64 + lane;
				const int t = ((unsigned)(p - lo - 1) < (unsigned)(hi - lo)) ? nb : s[r];
				s[r] = (p == lo) ? vi : t;
			}
		}
	}

	__device__ __forceinline__ int at(int pos) const // pos wave-uniform; result wave-uniform
	{
		int v = 0;
#pragma unroll
		for (int r = 0; r < R; ++r)
			if ((pos >> 6) == r)
				v = __builtin_amdgcn_readlane(s[r], pos & 63);
		return v;
	}
};

constexpr int SEG_F = 256; // outputs per line segment, frequency direction (one wave each)
constexpr int NL_F = 4;    // line segments per block
constexpr int SEG_T = 64;  // output rows per block, time direction
constexpr int NL_T = 64;   // columns per block, time direction (16 per wave)

// DIR 0: along cols (frequency), DIR 1: along rows (time)
template <int R, int DIR>
__global__ __launch_bounds__(256) void median_wave_kernel(FilterArgs a)
{
	extern __shared__ int smem[];
	constexpr int SEG = DIR == 0 ? SEG_F : SEG_T;
	constexpr int NL = DIR == 0 ? NL_F : NL_T;
	const int w = a.len, mid = w >> 1;
	const int span = SEG + 2 * mid;
	const int pin = span | 1; // odd pitches: column-wise LDS access stays conflict free
	constexpr int pout = SEG + 1;
	int* tin = smem;
	int* tout = smem + NL * pin;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: the per-line control flow is wave-uniform
	const float* __restrict__ src = a.src + (long long)blockIdx.z * a.src_stream_stride;
	float* __restrict__ dst = a.dst + (long long)blockIdx.z * a.dst_stream_stride;
	const int cols = a.cols;

	int segs_per_row = 0;
	long long total_lines = 0, line0 = 0;
	int c0 = 0, t0 = 0;
	if (DIR == 0) {
		segs_per_row = (cols + SEG - 1) / SEG;
		total_lines = (long long)a.n_out_rows * segs_per_row;
		line0 = (long long)blockIdx.x * NL;
		for (int e = tid; e < NL * span; e += 256) {
			const int line = e / span, u = e - line * span;
			const long long L = line0 + line;
			if (L < total_lines) {
				const long long row = L / segs_per_row;
				const int seg = (int)(L - row * segs_per_row);
				int col = seg * SEG - mid + u;
				col = col < 0 ? 0 : (col > cols - 1 ? cols - 1 : col);
				const long long srow = (a.first_row + row) % a.ring_rows;
				ZH_CHK(src + (srow * cols + col), 1);
				tin[line * pin + u] = f2key(src[srow * cols + col]);
			}
		}
	}
	else {
		c0 = blockIdx.x * NL;
		t0 = blockIdx.y * SEG;
		for (int e = tid; e < NL * span; e += 256) {
			const int line = e & (NL - 1), u = e / NL;
			const int col = c0 + line;
			if (col < cols) {
				long long ar = clampll(a.first_row + t0 - mid + u, a.clamp_lo, a.clamp_hi);
				ZH_CHK(src + ((ar % a.ring_rows) * cols + col), 1);
				tin[line * pin + u] = f2key(src[(ar % a.ring_rows) * cols + col]);
			}
		}
	}
	__syncthreads();

	constexpr int LINES_PER_WAVE = NL / 4;
	for (int li = 0; li < LINES_PER_WAVE; ++li) {
		const int line = wave * LINES_PER_WAVE + li;
		int n_out;
		if (DIR == 0) {
			const long long L = line0 + line;
			if (L >= total_lines)
				break;
			const int seg = (int)(L % segs_per_row);
			n_out = min(SEG, cols - seg * SEG);
		}
		else {
			if (c0 + line >= cols)
				break;
			n_out = min(SEG, a.n_out_rows - t0);
		}
		const int* X = tin + line * pin;
		int* O = tout + line * pout;
		SortedWindow<R> sw;
		sw.init();
		for (int u = 0; u < w; ++u)
			sw.replace(KEY_MAX, X[u], lane);
		int m = sw.at(mid);
		if (lane == 0)
			O[0] = m;
		for (int o = 1; o < n_out; ++o) {
			sw.replace(X[o - 1], X[o + w - 1], lane);
			m = sw.at(mid);
			if (lane == 0)
				O[o] = m;
		}
	}
	__syncthreads();

	if (DIR == 0) {
		for (int e = tid; e < NL * SEG; e += 256) {
			const int line = e / SEG, o = e - line * SEG;
			const long long L = line0 + line;
			if (L < total_lines) {
				const long long row = L / segs_per_row;
				const int seg = (int)(L - row * segs_per_row);
				const int col = seg * SEG + o;
				if (col < cols) {
					ZH_CHK(dst + (row * cols + col), 1);
					dst[row * cols + col] = key2f(tout[line * pout + o]);
				}
			}
		}
	}
	else {
		for (int e = tid; e < NL * SEG; e += 256) {
			const int line = e & (NL - 1), o = e / NL;
			const int col = c0 + line, row = t0 + o;
			if (col < cols && row < a.n_out_rows) {
				ZH_CHK(dst + ((long long)row * cols + col), 1);
				dst[(long long)row * cols + col] = key2f(tout[line * pout + o]);
			}
		}
	}
}

// len == 1: the median of one tap is the tap (e.g. time direction at hop >= 2048)
__global__ __launch_bounds__(256) void copy_rows_kernel(FilterArgs a)
{
	const float* __restrict__ src = a.src + (long long)blockIdx.y * a.src_stream_stride;
	float* __restrict__ dst = a.dst + (long long)blockIdx.y * a.dst_stream_stride;
	const long long n = (long long)a.n_out_rows * a.cols;
	for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
	     i += (long long)gridDim.x * blockDim.x) {
		const long long row = i / a.cols;
		const int col = (int)(i - row * a.cols);
		ZH_CHK(dst + i, 1);
		ZH_CHK(src + (((a.first_row + row) % a.ring_rows) * a.cols + col), 1);
		dst[i] = src[((a.first_row + row) % a.ring_rows) * a.cols + col];
	}
}

template <int R, int DIR>
int launch_wave(const FilterArgs& a, hipStream_t stream)
{
	constexpr int SEG = DIR == 0 ? SEG_F : SEG_T;
	constexpr int NL = DIR == 0 ? NL_F : NL_T;
	const int span = SEG + 2 * (a.len >> 1);
	const size_t lds = sizeof(int) * ((size_t)NL * (span | 1) + (size_t)NL * (SEG + 1));
	auto kern = median_wave_kernel<R, DIR>;
	if (lds > 64 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	dim3 grid;
	if (DIR == 0) {
		const long long lines = (long long)a.n_out_rows * ((a.cols + SEG - 1) / SEG);
		grid = dim3((unsigned)((lines + NL - 1) / NL), 1, (unsigned)a.n_streams);
	}
	else {
		grid = dim3((unsigned)((a.cols + NL - 1) / NL), (unsigned)((a.n_out_rows + SEG - 1) / SEG),
		            (unsigned)a.n_streams);
	}
	hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int DIR>
int launch_dir(const FilterArgs& a, hipStream_t stream)
{
	if (a.len <= 64)
		return launch_wave<1, DIR>(a, stream);
	if (a.len <= 128)
		return launch_wave<2, DIR>(a, stream);
	if (a.len <= 192)
		return launch_wave<3, DIR>(a, stream);
	return launch_wave<4, DIR>(a, stream);
}

} // namespace

int launch_median(const FilterArgs& a, hipStream_t stream, int* bits_done)
{
	if (bits_done)
		*bits_done = 0;
	if (a.n_out_rows <= 0 || a.cols <= 0 || a.n_streams <= 0)
		return ZEN_HIP_OK;
	if (a.len < 1 || !(a.len & 1))
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "median: mask length %d must be odd and positive", a.len);
	if (a.len > 255)
		ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "median: mask length %d > 255 not supported", a.len);
	if (a.len == 1) {
		const long long n = (long long)a.n_out_rows * a.cols;
		unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
		hipLaunchKernelGGL(copy_rows_kernel, dim3(blocks, (unsigned)a.n_streams), dim3(256), 0, stream, a);
		ZH_HIP(hipGetLastError());
		return ZEN_HIP_OK;
	}
	if (!a.force_general && !g_opt_median_general) { // sorting-network fast path (median_net.hip) for masks <= 63 taps
		bool handled = false;
		ZH_TRY(launch_median47_dpp(a, stream, &handled)); // headline shape: 47 taps on whole 4096-bin rows
		if (handled)
			return ZEN_HIP_OK;
		ZH_TRY(launch_median_net(a, stream, &handled, bits_done));
		if (handled)
			return ZEN_HIP_OK;
		ZH_TRY(launch_median_big(a, stream, &handled, bits_done)); // block-merge kernel for the long frequency masks
		if (handled)
			return ZEN_HIP_OK;
	}
	if (a.hermitian || a.pitch)
		ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "median: half rows (hermitian / pitch) are not implemented for mask %d on %d columns",
		        a.len, a.cols); // the engine asks filter_supports_hermitian() first
	if (a.direction == ZEN_HIP_FREQUENCY)
		return launch_dir<0>(a, stream);
	return launch_dir<1>(a, stream);
}

} // namespace zen_hip_impl
