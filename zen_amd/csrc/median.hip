// median.hip -- exact 1-D sliding median along either axis of a (rows x cols) float matrix, replicate
// border, for gfx950.  Semantics: MedianFilterCPU (libzen/mfilt.h:270-342), the reference's parity
// target; replaces the NPP calls of MedianFilterGPU::filter (libzen/mfilt.h:233-267).
//
// Kernels in this file: the general wave-cooperative sliding windows -- ANY odd length up to the filtered dimension, both
// directions (mfilt.h:296-305 accepts any filter_len <= the dimension; l_perc = 500 Hz of bins, hps.h:229, is 257 taps at
// fs 32 kHz / hop 4096, the CLI's default hop_h, zen/offline.h:19-32).  median_wave_kernel (masks <= 2047 taps): one 64-lane
// wavefront keeps the current window as a SORTED array spread over its lanes
// (position p lives in register p/64, lane p%64) and slides it one sample at a time: the leaving and the
// entering sample are located with two wave-wide compares + ballots (popcount = rank, the array is
// sorted so each ballot is a prefix mask), the elements between the two ranks move one lane up or down
// (wave-wide DPP shift + a select under a scalar lane mask),
// and the middle position is the output.  No data-dependent memory traffic: the line being filtered is
// staged once into LDS with coalesced loads, results go back through LDS so that stores are coalesced
// in both directions.  Values are ordered through the usual monotone float->int key, so the result is
// the bit pattern of an input sample (exact order statistic, as IPP's).
// median_long_kernel (longer masks, and time masks > 255 taps): the same sliding sorted window kept in MEMORY -- the
// wavefront's LDS, or a scratch buffer in device memory where one window exceeds the LDS -- searched 64 probes at a time and
// shifted 64 keys at a time; the first window of a line segment is sorted by a bitonic network the wavefront runs on it.
#include "common.h"
#include "filters.h"
#include "memguard.h"

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

// Per block: NL lines (frequency: 4 line segments, one per wave; time: 64 columns, 16 per wave) of SEG outputs each.  Every
// segment starts by inserting its first window sample by sample, so masks beyond 255 taps (R > 4) take segments four times
// as long, and in the time direction a quarter of the columns (the tile is (SEG + len - 1) x NL keys of LDS).
template <int R, int DIR>
struct WaveGeo {
	static constexpr int SEG = DIR == 0 ? (R > 4 ? 1024 : 256) : (R > 4 ? 128 : 64);
	static constexpr int NL = DIR == 0 ? 4 : (R > 4 ? 16 : 64);
};

// DIR 0: along cols (frequency), DIR 1: along rows (time)
template <int R, int DIR>
__global__ __launch_bounds__(256) void median_wave_kernel(FilterArgs a)
{
	extern __shared__ int smem[];
	constexpr int SEG = WaveGeo<R, DIR>::SEG;
	constexpr int NL = WaveGeo<R, DIR>::NL;
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

// ---- masks beyond the register window: the sorted window in memory -----------------------------------------------------
// One 64-lane workgroup (a wavefront) per line segment.  `win` is the wavefront's own: LDS, or -- GLOBAL -- a slice of a
// scratch buffer in device memory, read and written past the L1 (agent scope) so that a lane sees what another lane of the
// wavefront stored an instruction earlier.  A wavefront's LDS / memory instructions execute in program order, all 64 lanes of
// one before any of the next: every loop below reads its 64 keys in one instruction before it stores them in the next.
template <bool GLOBAL>
__device__ __forceinline__ int wld(const int* win, int i)
{
	if (GLOBAL)
		return __hip_atomic_load(win + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return win[i];
}
template <bool GLOBAL>
__device__ __forceinline__ void wst(int* win, int i, int v)
{
	if (GLOBAL)
		__hip_atomic_store(win + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else
		win[i] = v;
}
// between a step that stores keys and a step in which other lanes load them: nothing for LDS (in order); device memory: the
// stores are waited for and written through before the loads are issued
template <bool GLOBAL>
__device__ __forceinline__ void wsync()
{
	if (GLOBAL)
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
}

// ascending sort of win[0..n) by one wavefront: the bitonic network in its all-ascending form (a merge of 2^m starts by
// comparing element i with its mirror image in the block, then half-cleaners), in which a comparator never moves a larger
// key down -- so the network of the next power of two sorts n keys if every comparator that reaches past n is left out
// (the missing keys are +infinity and would not move).
template <bool GLOBAL>
__device__ void wave_sort(int* win, int n, int lane)
{
	int P = 1;
	while (P < n)
		P <<= 1;
	for (int k = 2; k <= P; k <<= 1) {
		const int hk = k >> 1;
		wsync<GLOBAL>();
		for (int i = lane; i < (P >> 1); i += 64) {
			const int blk = i / hk, off = i - blk * hk;
			const int ia = blk * k + off, ib = blk * k + k - 1 - off;
			if (ib < n) {
				const int x = wld<GLOBAL>(win, ia), y = wld<GLOBAL>(win, ib);
				if (x > y) {
					wst<GLOBAL>(win, ia, y);
					wst<GLOBAL>(win, ib, x);
				}
			}
		}
		for (int j = k >> 2; j >= 1; j >>= 1) {
			wsync<GLOBAL>();
			for (int i = lane; i < (P >> 1); i += 64) {
				const int q = i / j;
				const int ia = q * 2 * j + (i - q * j), ib = ia + j;
				if (ib < n) {
					const int x = wld<GLOBAL>(win, ia), y = wld<GLOBAL>(win, ib);
					if (x > y) {
						wst<GLOBAL>(win, ia, y);
						wst<GLOBAL>(win, ib, x);
					}
				}
			}
		}
	}
}

// number of keys < v in the sorted win[0..w), v wave-uniform: 64 probes per round (the comparisons are a prefix of the lanes)
template <bool GLOBAL>
__device__ __forceinline__ int wave_rank(const int* win, int w, int v, int lane)
{
	int lo = 0, n = w; // the answer is in [lo, lo + n]
	while (n > 0) {
		const int chunk = (n + 63) >> 6;
		const int i = lo + (lane + 1) * chunk - 1;
		const bool below = (i < lo + n) && (wld<GLOBAL>(win, i) < v);
		const int c = __popcll(__ballot(below));
		const int end = lo + n;
		lo += c * chunk;
		n = min(chunk - 1, end - lo); // probe c (if any) is >= v: the answer lies in the chunk - 1 keys in front of it
	}
	return lo;
}

struct LongGeo {
	int seg_len;          // outputs per segment
	int segs_per_line;
	long long n_lines;    // frequency: output rows; time: columns
	int line_len;         // outputs per line (frequency: cols, time: n_out_rows)
	int* scratch;         // GLOBAL: windows of `scratch_pitch` keys, one per workgroup of the launch
	long long scratch_pitch;
};

template <int DIR, bool GLOBAL>
__global__ __launch_bounds__(64) void median_long_kernel(FilterArgs a, LongGeo g)
{
	extern __shared__ int smem[];
	const int lane = threadIdx.x;
	const long long wg = (long long)blockIdx.x + (long long)blockIdx.y * gridDim.x;
	if (wg >= g.n_lines * g.segs_per_line)
		return;
	int* win = GLOBAL ? g.scratch + (wg + (long long)blockIdx.z * g.n_lines * g.segs_per_line) * g.scratch_pitch : smem;
	const long long line = wg / g.segs_per_line;
	const int seg = (int)(wg - line * g.segs_per_line);
	const int w = a.len, mid = w >> 1, cols = a.cols;
	const float* __restrict__ src = a.src + (long long)blockIdx.z * a.src_stream_stride;
	float* __restrict__ dst = a.dst + (long long)blockIdx.z * a.dst_stream_stride;
	const int o0 = seg * g.seg_len, o1 = min(o0 + g.seg_len, g.line_len);
	// key of the line's sample at position u (any integer: replicate border), and where output o goes
	const long long srow = DIR == 0 ? ((a.first_row + line) % a.ring_rows) * (long long)cols : 0;
	auto key_at = [&](long long u) -> int {
		if (DIR == 0) {
			const int c = (int)(u < 0 ? 0 : (u > cols - 1 ? cols - 1 : u));
			ZH_CHK(src + (srow + c), 1);
			return f2key(src[srow + c]);
		}
		const long long r = clampll(a.first_row + u, a.clamp_lo, a.clamp_hi);
		ZH_CHK(src + ((r % a.ring_rows) * cols + line), 1);
		return f2key(src[(r % a.ring_rows) * cols + line]);
	};
	auto out_ptr = [&](int o) -> float* { return DIR == 0 ? dst + (line * cols + o) : dst + ((long long)o * cols + line); };

	for (int i = lane; i < w; i += 64) // the first window of the segment: taps o0 - mid .. o0 + mid
		wst<GLOBAL>(win, i, key_at((long long)o0 - mid + i));
	wave_sort<GLOBAL>(win, w, lane);

	int res = 0, pre_out = 0, pre_in = 0; // lane l: the result of output ob + l; the samples leaving / entering at output ob + 1 + l
	for (int ob = o0; ob < o1; ob += 64) {
		if (ob + lane + 1 < o1) {
			pre_out = key_at((long long)ob + lane - mid);
			pre_in = key_at((long long)ob + lane + 1 + mid);
		}
		const int nb = min(64, o1 - ob);
		for (int k = 0; k < nb; ++k) {
			wsync<GLOBAL>();
			const int m = wld<GLOBAL>(win, mid);
			if (lane == k)
				res = m;
			if (ob + k + 1 >= o1)
				break;
			const int vo = __builtin_amdgcn_readlane(pre_out, k), vi = __builtin_amdgcn_readlane(pre_in, k);
			if (vi == vo)
				continue;
			const int p_out = wave_rank<GLOBAL>(win, w, vo, lane); // position of the first vo
			const int c_in = wave_rank<GLOBAL>(win, w, vi, lane);  // keys < vi
			if (vi > vo) { // positions [p_out, c_in - 1) take their upper neighbour, vi lands at c_in - 1
				const int hi = c_in - 1;
				for (int base = p_out; base < hi; base += 64) {
					const int p = base + lane;
					const int t = p < hi ? wld<GLOBAL>(win, p + 1) : 0;
					if (p < hi)
						wst<GLOBAL>(win, p, t);
				}
				if (lane == 0)
					wst<GLOBAL>(win, hi, vi);
			}
			else { // vi lands at c_in, positions (c_in, p_out] take their lower neighbour
				for (int top = p_out; top > c_in; top -= 64) {
					const int p = top - lane;
					const int t = p > c_in ? wld<GLOBAL>(win, p - 1) : 0;
					if (p > c_in)
						wst<GLOBAL>(win, p, t);
				}
				if (lane == 0)
					wst<GLOBAL>(win, c_in, vi);
			}
		}
		if (lane < nb) {
			ZH_CHK(out_ptr(ob + lane), 1);
			*out_ptr(ob + lane) = key2f(res);
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
	constexpr int SEG = WaveGeo<R, DIR>::SEG;
	constexpr int NL = WaveGeo<R, DIR>::NL;
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

constexpr size_t LONG_LDS_MAX = 150 * 1024; // one window per 64-lane workgroup (the CU has 160 KB)

template <int DIR>
int launch_long(const FilterArgs& a, hipStream_t stream)
{
	LongGeo g;
	memset(&g, 0, sizeof(g));
	g.n_lines = DIR == 0 ? a.n_out_rows : a.cols;
	g.line_len = DIR == 0 ? a.cols : a.n_out_rows;
	// a segment starts with a sort of its first window (about as much work as 50-200 slides): segments of at least two
	// windows' length, whole lines where there are enough lines to fill the device
	int seg = a.len * 2 > 2048 ? a.len * 2 : 2048;
	if (g.n_lines >= 2048 || seg > g.line_len)
		seg = g.line_len;
	g.seg_len = seg;
	g.segs_per_line = (g.line_len + seg - 1) / seg;
	const long long wgs = g.n_lines * g.segs_per_line;
	const unsigned gx = (unsigned)(wgs < 32768 ? wgs : 32768), gy = (unsigned)((wgs + gx - 1) / gx);
	if (gy > 65535 || a.n_streams > 65535)
		ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "median: %lld lines x %d streams exceed the launch grid", wgs, a.n_streams);
	const dim3 grid(gx, gy, (unsigned)a.n_streams);
	const size_t lds = sizeof(int) * (size_t)a.len;
	if (lds <= LONG_LDS_MAX) {
		auto kern = median_long_kernel<DIR, false>;
		if (lds > 64 * 1024)
			ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		hipLaunchKernelGGL(kern, grid, dim3(64), lds, stream, a, g);
		ZH_HIP(hipGetLastError());
		return ZEN_HIP_OK;
	}
	// a window longer than the LDS: scratch in device memory, given back when the launch has finished (a mask of > 38 400
	// taps: the drop-in wrapper on a matrix of that many columns or rows -- no configuration of the engine comes near)
	g.scratch_pitch = ((long long)a.len + 63) & ~63LL;
	ZH_HIP(zh_malloc((void**)&g.scratch, sizeof(int) * (size_t)g.scratch_pitch * (size_t)wgs * (size_t)a.n_streams));
	LongGeo gs = g; // (the streams of a launch: blockIdx.z has its own slices)
	hipLaunchKernelGGL((median_long_kernel<DIR, true>), grid, dim3(64), 0, stream, a, gs);
	const hipError_t le = hipGetLastError();
	const hipError_t se = hipStreamSynchronize(stream);
	(void)zh_free(g.scratch);
	ZH_HIP(le);
	ZH_HIP(se);
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
	if (a.len <= 256)
		return launch_wave<4, DIR>(a, stream);
	if (a.len <= 384)
		return launch_wave<6, DIR>(a, stream);
	if (a.len <= 512)
		return launch_wave<8, DIR>(a, stream);
	if (a.len <= 768)
		return launch_wave<12, DIR>(a, stream);
	if (a.len <= 1024)
		return launch_wave<16, DIR>(a, stream);
	if (a.len <= 1536)
		return launch_wave<24, DIR>(a, stream);
	if (a.len <= 2048)
		return launch_wave<32, DIR>(a, stream);
	return launch_long<DIR>(a, stream);
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
