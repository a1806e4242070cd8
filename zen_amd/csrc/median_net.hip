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
#include "median_net.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

using znet::f2key;
using znet::key2f;

__device__ __forceinline__ int lds_off(int g) { return g + 4 * (g >> 6); }
constexpr int lds_words(int n) { return n + 4 * ((n + 63) / 64) + 4; }

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
template <int W>
__global__ __launch_bounds__(256) void median_net_freq_kernel(FilterArgs a, RowMap rm, int segs_per_row, int vec_ok)
{
	constexpr int T = znet::outputs_per_thread(W), mid = W / 2;
	constexpr int NV = (W + T - 1 + 3) / 4, NE = NV * 4;
	constexpr int OUTS = 256 * T;
	constexpr int SPAN = 255 * T + NE; // last word any thread reads, + 1
	__shared__ __attribute__((aligned(16))) int tile[lds_words(SPAN)];
	__shared__ __attribute__((aligned(16))) int otile[lds_words(OUTS)];

	const int tid = threadIdx.x;
	const int row = blockIdx.x / segs_per_row, seg = blockIdx.x - row * segs_per_row;
	const int cols = a.cols, col0 = seg * OUTS;
	const float* __restrict__ srow =
	    a.src + (long long)blockIdx.y * a.src_stream_stride + (long long)map_row(rm, row) * cols;
	float* __restrict__ drow = a.dst + (long long)blockIdx.y * a.dst_stream_stride + (long long)row * cols;

	const int g_lo = col0 - mid; // column of tile word 0
	if (vec_ok) {
		// interior: aligned float4 loads, each word dropped at its tile position
		int v0 = g_lo < 0 ? 0 : (g_lo & ~3);
		int v1 = col0 + OUTS + mid;
		v1 = v1 > cols ? cols : v1;
		for (int vc = v0 + 4 * tid; vc < v1; vc += 4 * 256) {
			const float4 x = *reinterpret_cast<const float4*>(srow + vc);
			const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const int g = vc + j - g_lo;
				if (g >= 0 && g < SPAN)
					tile[lds_off(g)] = f2key(xs[j]);
			}
		}
		// replicate border (only the first / last segment of a row has any)
		if (g_lo < 0) {
			const int k0 = f2key(srow[0]);
			for (int g = tid; g < -g_lo; g += 256)
				tile[lds_off(g)] = k0;
		}
		if (col0 + OUTS + mid > cols) {
			const int k1 = f2key(srow[cols - 1]);
			for (int g = cols - g_lo + tid; g < OUTS + 2 * mid; g += 256)
				tile[lds_off(g)] = k1;
		}
	}
	else {
		for (int g = tid; g < OUTS + 2 * mid; g += 256) {
			int c = g_lo + g;
			c = c < 0 ? 0 : (c > cols - 1 ? cols - 1 : c);
			tile[lds_off(g)] = f2key(srow[c]);
		}
	}
	__syncthreads();

	int e[NE], out[T];
#pragma unroll
	for (int v = 0; v < NV; ++v) {
		const int4 x = *reinterpret_cast<const int4*>(&tile[lds_off(tid * T + 4 * v)]);
		e[4 * v] = x.x;
		e[4 * v + 1] = x.y;
		e[4 * v + 2] = x.z;
		e[4 * v + 3] = x.w;
	}
	znet::medians<W, T, NE>(e, out);
#pragma unroll
	for (int v = 0; v < T / 4; ++v)
		*reinterpret_cast<int4*>(&otile[lds_off(tid * T + 4 * v)]) =
		    make_int4(out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]);
	__syncthreads();

	if (vec_ok) {
		for (int g = 4 * tid; g < OUTS; g += 4 * 256) {
			const int c = col0 + g;
			if (c < cols) { // cols % 4 == 0: whole vector inside
				const int4 k = *reinterpret_cast<const int4*>(&otile[lds_off(g)]);
				*reinterpret_cast<float4*>(drow + c) = make_float4(key2f(k.x), key2f(k.y), key2f(k.z), key2f(k.w));
			}
		}
	}
	else {
		for (int g = tid; g < OUTS; g += 256) {
			const int c = col0 + g;
			if (c < cols)
				drow[c] = key2f(otile[lds_off(g)]);
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
	if constexpr (VC == 1) {
		k[0] = f2key(*p);
	}
	else if constexpr (VC == 2) {
		const float2 x = *reinterpret_cast<const float2*>(p);
		k[0] = f2key(x.x);
		k[1] = f2key(x.y);
	}
	else {
		const float4 x = *reinterpret_cast<const float4*>(p);
		k[0] = f2key(x.x);
		k[1] = f2key(x.y);
		k[2] = f2key(x.z);
		k[3] = f2key(x.w);
	}
}

template <int VC>
__device__ __forceinline__ void store_keys(float* p, const int (&k)[VC])
{
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

template <int W, int VC>
__global__ __launch_bounds__(256) void median_net_time_kernel(FilterArgs a, RowMap rm, int rows_per_block)
{
	constexpr int T = znet::outputs_per_thread(W), mid = W / 2, NE = W + T - 1;
	const int cols = a.cols;
	const int c = (blockIdx.x * 256 + threadIdx.x) * VC;
	if (c >= cols)
		return;
	const float* __restrict__ src = a.src + (long long)blockIdx.z * a.src_stream_stride + c;
	float* __restrict__ dst = a.dst + (long long)blockIdx.z * a.dst_stream_stride + c;
	const int r0 = blockIdx.y * rows_per_block;
	int rend = r0 + rows_per_block;
	rend = rend > a.n_out_rows ? a.n_out_rows : rend;

	int e[VC][NE];
#pragma unroll
	for (int q = 0; q < W - 1; ++q) { // taps r0-mid .. r0+mid-1
		int k[VC];
		load_keys<VC>(src + (long long)map_row(rm, r0 - mid + q) * cols, k);
#pragma unroll
		for (int v = 0; v < VC; ++v)
			e[v][q] = k[v];
	}
	for (int rr = r0; rr < rend; rr += T) {
#pragma unroll
		for (int i = 0; i < T; ++i) { // T new rows: rr+mid .. rr+mid+T-1
			int k[VC];
			load_keys<VC>(src + (long long)map_row(rm, rr + mid + i) * cols, k);
#pragma unroll
			for (int v = 0; v < VC; ++v)
				e[v][W - 1 + i] = k[v];
		}
		int out[VC][T];
#pragma unroll
		for (int v = 0; v < VC; ++v)
			znet::medians<W, T, NE>(e[v], out[v]);
#pragma unroll
		for (int g = 0; g < T; ++g) {
			if (rr + g < rend) {
				int k[VC];
#pragma unroll
				for (int v = 0; v < VC; ++v)
					k[v] = out[v][g];
				store_keys<VC>(dst + (long long)(rr + g) * cols, k);
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
int launch_freq(const FilterArgs& a, const RowMap& rm, hipStream_t stream)
{
	constexpr int T = znet::outputs_per_thread(W);
	const int segs = (a.cols + 256 * T - 1) / (256 * T);
	const int vec_ok = (a.cols % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.src) & 15) == 0)
	                   && ((reinterpret_cast<uintptr_t>(a.dst) & 15) == 0) && (a.src_stream_stride % 4 == 0)
	                   && (a.dst_stream_stride % 4 == 0);
	dim3 grid((unsigned)((long long)a.n_out_rows * segs), (unsigned)a.n_streams);
	hipLaunchKernelGGL(median_net_freq_kernel<W>, grid, dim3(256), 0, stream, a, rm, segs, vec_ok);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int W, int VC>
int launch_time_vc(const FilterArgs& a, const RowMap& rm, hipStream_t stream)
{
	constexpr int T = znet::outputs_per_thread(W);
	const int col_blocks = (a.cols + 256 * VC - 1) / (256 * VC);
	// enough row blocks to fill the chip (>= ~2048 workgroups), but long enough to amortise the W-1 halo
	int rpb = 256;
	while (rpb > 4 * T && rpb > 32 && (long long)col_blocks * ((a.n_out_rows + rpb - 1) / rpb) * a.n_streams < 2048)
		rpb >>= 1;
	rpb = (rpb + T - 1) / T * T;
	dim3 grid((unsigned)col_blocks, (unsigned)((a.n_out_rows + rpb - 1) / rpb), (unsigned)a.n_streams);
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
int launch_median_net(const FilterArgs& a, hipStream_t stream, bool* handled)
{
	*handled = false;
	RowMap rm;
	if (a.len < 3 || a.len > 63 || !prepare(a, &rm))
		return ZEN_HIP_OK;
	const bool freq = a.direction == ZEN_HIP_FREQUENCY;
	if (freq && a.len < 7)
		return ZEN_HIP_OK; // T < 4: the 16-byte LDS path does not apply; tiny masks go the general way
	*handled = true;
	switch (a.len) {
#define X(W) \
	case W: return freq ? launch_freq<W>(a, rm, stream) : launch_time<W>(a, rm, stream);
		ZH_NET_WIDTHS(X)
#undef X
	default: *handled = false; return ZEN_HIP_OK;
	}
}

} // namespace zen_hip_impl
