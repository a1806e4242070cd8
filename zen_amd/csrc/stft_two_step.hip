// stft_two_step.hip -- analysis and synthesis of BLOCKS of frames at the long transforms (nfft 8192 / 16384: pass 1 of
// the default offline configuration, hop 2048 / 4096) as two-step transforms with many small workgroups.
//
// Why.  stft_kernel<14> / istft_kernel<14> keep one frame's 139 KB image in LDS: one 1024-thread workgroup per CU,
// so a frame's loads (64-160 KB at the CU's share of HBM bandwidth: 6-7 us), its butterflies (~10 us) and its stores
// never overlap -- 16 us per transform and CU, a third of the roof.  Here the transform is cut in two steps exactly as
// rt_wide.hip cuts it for single hops (and fft_big.hip for nfft 32768), but for throughput:
//
//   N = M*J, M = 128.   step A: J independent M-point transforms of the decimated sequences x[j + n*J] (stages 1..7,
//                               twiddle tw[(k*M/2^s)*J]): 8 threads per sequence, 32 sequences per 256-thread workgroup.
//                       step B: per column kappa < M a J-point transform of Y_7[j][kappa] (stages 8..log2 N, twiddle
//                               tw[(q*J/2^t)*M + kappa*J/2^t]): 4 or 8 threads per column.
//   Every butterfly is the one the one-piece kernels and the oracle evaluate (tests/test_two_step_fft.py).
//
// A workgroup needs 35 KB of LDS and ~100 VGPRs: four to five per CU, whose loads, butterflies and stores overlap.  The
// values the steps exchange (8*N bytes per frame) go through a scratch buffer that the caller sizes to a few hundred
// frames: written by step A and read back by step B microseconds later, it lives in the L2 / Infinity Cache and costs
// no HBM traffic to speak of.  Two launches per sub-batch of frames (and per output in the synthesis).
#include "common.h"
#include "fft_dev.h"
#include "masks.h"
#include "stft.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

using zfft::Plan;

constexpr int WGT = 256;

template <int LOG2N>
struct Geo2 {
	static constexpr int N = 1 << LOG2N;
	static constexpr int LOG2M = 7, M = 1 << LOG2M;
	static constexpr int LOG2J = LOG2N - LOG2M, J = 1 << LOG2J;
	using PA = Plan<LOG2M>;
	using PB = Plan<LOG2J>;
	static constexpr int FA = WGT / PA::TF, FB = WGT / PB::TF; // sequences / columns per workgroup
	static constexpr int GA = J / FA, GB = M / FB;             // workgroups per frame in either step
	static_assert(PA::TF <= 64 && PB::TF <= 64, "frames are synchronised inside their wavefront");
	static_assert(J % FA == 0 && M % FB == 0, "whole workgroups");
	static constexpr size_t LDS_A = sizeof(float2) * FA * PA::LDS_FLOAT2, LDS_B = sizeof(float2) * FB * PB::LDS_FLOAT2;
};

template <int LOG2J>
struct StrideTw { // step A: index idx of the M-point transform's table = entry idx*J of the N-point table
	const float2* __restrict__ p;
	__device__ __forceinline__ float2 operator()(int, int idx) const { return p[idx << LOG2J]; }
};
template <int LOG2M, int LOG2J>
struct TwistTw { // step B: stage t, plain J-point index idx -> entry idx*M + kappa*J/2^t
	const float2* __restrict__ p;
	int kappa;
	__device__ __forceinline__ float2 operator()(int t, int idx) const { return p[(idx << LOG2M) + (kappa << (LOG2J - t))]; }
};
template <int LOG2J>
struct XchOut { // Y_7[j][k] -> T[k][j]: step B reads a column contiguously
	float2* T;
	int j;
	__device__ __forceinline__ void operator()(int k, float2 X, bool, int) const { T[(k << LOG2J) + j] = X; }
};
template <int LOG2J>
struct XchIn {
	const float2* T;
	int kappa;
	__device__ __forceinline__ float2 operator()(int jj, int) const { return T[(kappa << LOG2J) + jj]; }
};

// lanes of step A: neighbouring lanes hold neighbouring sequences j (same position inside the sequence), so the strided
// elements x[j + n*J] they ask for are neighbours in memory; a sequence still lives inside one wavefront
template <int LOG2N>
__device__ __forceinline__ void lanes_a(int t, int g, int& tf, int& f, int& j)
{
	using G = Geo2<LOG2N>;
	constexpr int FW = 64 / G::PA::TF; // sequences per wavefront
	tf = (t & 63) / FW;
	f = (t >> 6) * FW + (t & 63) % FW;
	j = g * G::FA + f;
}

// ------------------------------------------------------------------------------------------------ analysis
template <int LOG2J>
struct FwdAIn { // x[j + n*J] of the windowed, zero-padded frame (only n < M/2 is asked for: ZU)
	const float* prev;
	const float* cur;
	const float* window;
	int hop, j;
	__device__ __forceinline__ float2 operator()(int n, int) const
	{
		const int idx = j + (n << LOG2J);
		const float x = idx < hop ? prev[idx] : cur[idx - hop];
		return make_float2(x * window[idx], 0.0f); // window_functor hps.h:24-33
	}
};
template <int LOG2M>
struct FwdBOut { // X[kappa + M*q]: what StftOut of stft.hip does with it
	float2* S;
	float* mag;
	int kappa, n;
	bool full;
	__device__ __forceinline__ void operator()(int q, float2 X, bool, int) const
	{
		const int k = kappa + (q << LOG2M);
		if (k <= (n >> 1)) {
			S[k] = X;
			const float m = zfft::cabs_exact(X.x, X.y); // complex_abs_functor hps.h:82-89
			mag[k] = m;
			if (full && k != 0 && k != (n >> 1))
				mag[n - k] = m; // |S[n-k]| == |S[k]| bit for bit
		}
	}
};

template <int LOG2N>
__global__ __launch_bounds__(WGT) void stft2_a_kernel(StftArgs a, int item0, int n_items)
{
	using G = Geo2<LOG2N>;
	using PA = typename G::PA;
	extern __shared__ float2 lds[];
	const int t = threadIdx.x, hop = a.hop;
	if ((int)blockIdx.y == n_items) { // housekeeping (the extra block of stft_kernel), once per call: input tail and carries
		if (item0 != 0)
			return;
		for (int s = blockIdx.x; s < a.n_streams; s += gridDim.x) {
			const float* last = a.in + (long long)s * a.in_stride + (long long)(a.n_frames - 1) * hop;
			for (int i = t; i < hop; i += WGT)
				a.tail_next[(long long)s * hop + i] = last[i];
			if (a.prev_frames > 0) {
				for (int o = 0; o < 3; ++o) {
					if (!a.carry[o])
						continue;
					const float* y = a.Y[o] + (long long)s * a.y_stream_stride + (long long)(a.prev_frames - 1) * (2 * hop) + hop;
					for (int i = t; i < hop; i += WGT)
						a.carry[o][(long long)s * hop + i] = y[i];
				}
			}
		}
		return;
	}
	const int item = item0 + blockIdx.y, s = item / a.n_frames, f = item - s * a.n_frames;
	int tf, fl, j;
	lanes_a<LOG2N>(t, blockIdx.x, tf, fl, j);
	const float* in_s = a.in + (long long)s * a.in_stride;
	zfft::TwRegs<G::LOG2M> twr;
	twr.fill_with(tf, StrideTw<G::LOG2J>{a.tw});
	FwdAIn<G::LOG2J> in{f == 0 ? a.tail_prev + (long long)s * hop : in_s + (long long)(f - 1) * hop, in_s + (long long)f * hop,
	                    a.window, hop, j};
	XchOut<G::LOG2J> out{a.xch + (long long)blockIdx.y * G::N, j};
	zfft::PassRunner<G::LOG2M, 0, false, true, false, FwdAIn<G::LOG2J>, XchOut<G::LOG2J>, false, zfft::TwRegs<G::LOG2M>>::run(
	    tf, lds + fl * PA::LDS_FLOAT2, twr, in, out, true);
}

template <int LOG2N>
__global__ __launch_bounds__(WGT) void stft2_b_kernel(StftArgs a, int item0)
{
	using G = Geo2<LOG2N>;
	using PB = typename G::PB;
	extern __shared__ float2 lds[];
	const int t = threadIdx.x;
	const int item = item0 + blockIdx.y, s = item / a.n_frames, f = item - s * a.n_frames;
	const int fl = t / PB::TF, tf = t % PB::TF, kappa = blockIdx.x * G::FB + fl;
	const long long row = ((a.row0 + f) % a.ring_rows) + (long long)s * a.ring_rows;
	zfft::TwRegs<G::LOG2J, false> twr; // (not the J-point transform's own table: no trivial-twiddle shortcuts)
	twr.fill_with(tf, TwistTw<G::LOG2M, G::LOG2J>{a.tw, kappa});
	XchIn<G::LOG2J> in{a.xch + (long long)blockIdx.y * G::N, kappa};
	FwdBOut<G::LOG2M> out{a.S + row * a.s_stride, a.mag + row * G::N, kappa, G::N, f >= a.mag_full_from};
	zfft::PassRunner<G::LOG2J, 0, false, false, false, XchIn<G::LOG2J>, FwdBOut<G::LOG2M>, false, zfft::TwRegs<G::LOG2J, false>>::run(
	    tf, lds + fl * PB::LDS_FLOAT2, twr, in, out, true);
}

// ------------------------------------------------------------------------------------------------ synthesis
template <int LOG2J>
struct InvAIn { // (S * mask)[j + n*J] (IstftIn of istft.hip): the upper half of the spectrum is the conjugate mirror image
	const float2* S;
	const float* H;
	const float* P;
	MaskCfg cfg;
	HardThr thr;
	int which, n, p_mid, j;
	__device__ __forceinline__ float2 operator()(int nn, int) const
	{
		const int idx = j + (nn << LOG2J);
		const bool mirror = idx > (n >> 1);
		const int lo = mirror ? n - idx : idx;
		float2 z = S[lo];
		if (mirror)
			z.y = -z.y;
		const int pi = (mirror && idx >= n - p_mid) ? idx : lo;
		const float m = mask_value_thr(which, H[lo], P[pi], cfg, thr);
		return make_float2(z.x * m, z.y * m); // apply_mask_functor hps.h:58-66
	}
};
template <int LOG2M>
struct InvBOut { // x[kappa + M*q], q < J/2 (HALF_OUT): the nwin real outputs that are used
	float* Y;
	float cola;
	int kappa;
	__device__ __forceinline__ void operator()(int q, float2 x, bool, int) const
	{
		Y[kappa + (q << LOG2M)] = x.x * cola; // the product of overlap_add_functor hps.h:68-80; the sum is in finalize
	}
};

template <int LOG2N>
__global__ __launch_bounds__(WGT) void istft2_a_kernel(IstftArgs a, int oi, int item0)
{
	using G = Geo2<LOG2N>;
	using PA = typename G::PA;
	extern __shared__ float2 lds[];
	const int t = threadIdx.x;
	const int item = item0 + blockIdx.y, s = item / a.n_frames, f = item - s * a.n_frames;
	int tf, fl, j;
	lanes_a<LOG2N>(t, blockIdx.x, tf, fl, j);
	const long long ring_row = ((a.crow0 + f) % a.ring_rows) + (long long)s * a.ring_rows;
	zfft::TwRegs<G::LOG2M> twr;
	twr.fill_with(tf, StrideTw<G::LOG2J>{a.tw});
	InvAIn<G::LOG2J> in{a.S + ring_row * a.s_stride,
	                    a.h_is_ring ? a.H + ring_row * G::N : a.H + (long long)s * a.h_stream_stride + (long long)f * G::N,
	                    a.P + (long long)s * a.p_stream_stride + (long long)f * G::N,
	                    MaskCfg{a.beta, a.beta_h, a.soft, a.power, a.sse, a.out_h, a.out_p},
	                    HardThr{a.thr_p, a.thr_h, a.thr_p_inc, a.thr_h_inc},
	                    a.out_id[oi], G::N, a.p_mid, j};
	XchOut<G::LOG2J> out{a.xch + (long long)blockIdx.y * G::N, j};
	zfft::PassRunner<G::LOG2M, 0, true, false, false, InvAIn<G::LOG2J>, XchOut<G::LOG2J>, false, zfft::TwRegs<G::LOG2M>>::run(
	    tf, lds + fl * PA::LDS_FLOAT2, twr, in, out, true);
}

template <int LOG2N>
__global__ __launch_bounds__(WGT) void istft2_b_kernel(IstftArgs a, int oi, int item0)
{
	using G = Geo2<LOG2N>;
	using PB = typename G::PB;
	extern __shared__ float2 lds[];
	const int t = threadIdx.x;
	const int item = item0 + blockIdx.y, s = item / a.n_frames, f = item - s * a.n_frames;
	const int fl = t / PB::TF, tf = t % PB::TF, kappa = blockIdx.x * G::FB + fl;
	zfft::TwRegs<G::LOG2J, false> twr;
	twr.fill_with(tf, TwistTw<G::LOG2M, G::LOG2J>{a.tw, kappa});
	XchIn<G::LOG2J> in{a.xch + (long long)blockIdx.y * G::N, kappa};
	InvBOut<G::LOG2M> out{a.Y[oi] + (long long)s * a.y_stream_stride + (long long)f * (G::N / 2), a.cola, kappa};
	zfft::PassRunner<G::LOG2J, 0, true, false, true, XchIn<G::LOG2J>, InvBOut<G::LOG2M>, false, zfft::TwRegs<G::LOG2J, false>>::run(
	    tf, lds + fl * PB::LDS_FLOAT2, twr, in, out, true);
}

template <int LOG2N>
int launch_stft2_t(const StftArgs& a, hipStream_t stream)
{
	using G = Geo2<LOG2N>;
	const long long total = (long long)a.n_streams * a.n_frames;
	for (long long i0 = 0; i0 < total; i0 += a.xch_frames) {
		const int n = (int)(total - i0 < a.xch_frames ? total - i0 : a.xch_frames);
		hipLaunchKernelGGL(stft2_a_kernel<LOG2N>, dim3(G::GA, (unsigned)n + (i0 == 0 ? 1 : 0)), dim3(WGT), G::LDS_A, stream, a, (int)i0, n);
		hipLaunchKernelGGL(stft2_b_kernel<LOG2N>, dim3(G::GB, (unsigned)n), dim3(WGT), G::LDS_B, stream, a, (int)i0);
	}
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int LOG2N>
int launch_istft2_t(const IstftArgs& a, hipStream_t stream)
{
	using G = Geo2<LOG2N>;
	const long long total = (long long)a.n_streams * a.n_frames;
	for (long long i0 = 0; i0 < total; i0 += a.xch_frames) {
		const int n = (int)(total - i0 < a.xch_frames ? total - i0 : a.xch_frames);
		for (int oi = 0; oi < a.n_out; ++oi) {
			hipLaunchKernelGGL(istft2_a_kernel<LOG2N>, dim3(G::GA, (unsigned)n), dim3(WGT), G::LDS_A, stream, a, oi, (int)i0);
			hipLaunchKernelGGL(istft2_b_kernel<LOG2N>, dim3(G::GB, (unsigned)n), dim3(WGT), G::LDS_B, stream, a, oi, (int)i0);
		}
	}
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace

bool stft_two_step_available(int log2n) { return log2n == 13 || log2n == 14; }

int launch_stft_two_step(int log2n, const StftArgs& a, hipStream_t stream)
{
	if (a.n_frames <= 0)
		return ZEN_HIP_OK;
	if (!a.xch || a.xch_frames <= 0)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "two-step analysis without its exchange buffer");
	switch (log2n) {
	case 13: return launch_stft2_t<13>(a, stream);
	case 14: return launch_stft2_t<14>(a, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "two-step analysis: nfft = 2^%d", log2n);
	}
}

int launch_istft_two_step(int log2n, const IstftArgs& a, hipStream_t stream)
{
	if (a.n_frames <= 0 || a.n_out <= 0)
		return ZEN_HIP_OK;
	if (!a.xch || a.xch_frames <= 0)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "two-step synthesis without its exchange buffer");
	switch (log2n) {
	case 13: return launch_istft2_t<13>(a, stream);
	case 14: return launch_istft2_t<14>(a, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "two-step synthesis: nfft = 2^%d", log2n);
	}
}

} // namespace zen_hip_impl
