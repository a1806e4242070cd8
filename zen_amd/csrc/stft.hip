// stft.hip -- analysis / synthesis kernels of the HPSS engine for gfx950, built on fft_dev.h.
//
//   stft_kernel   : hps.cu:452-472 + :492 (new row only): frame assembly from the hop stream, sqrt-Hann
//                   window, zero-padding (pruned: the upper half is never loaded), forward FFT, spectrum
//                   and magnitude rows written into the ring.  Replaces thrust::copy x2, transform, fill,
//                   cufftExecC2C, the STFT rotation (2 overlapping copies) and the whole-matrix abs.
//   istft_kernel  : hps.cu:498-579 per consumed row: mask from (H, P), complex*real, inverse FFT
//                   (only the nwin real outputs that are used are computed), *COLA.
//   finalize_kernel: overlap-add of neighbouring frames + copy-out (hps.cu:435-449, :526-528, :341-363).
//   fft_kernel    : the plain FFTC2CWrapperGPU transform (fftw.h:35-43).
#include "common.h"
#include "fft_dev.h"
#include "masks.h"
#include "stft.h"

#include <cfloat>

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

using zfft::Plan;

// ------------------------------------------------------------------------------------------------
template <int LOG2N>
struct StftIn {
	const float* prev; // hop samples before the frame's second half
	const float* cur;  // hop samples
	const float* window;
	int hop;
	__device__ __forceinline__ float2 operator()(int idx, int) const
	{
		const float x = idx < hop ? prev[idx] : cur[idx - hop];
		return make_float2(x * window[idx], 0.0f); // window_functor hps.h:24-33
	}
};

struct StftOut {
	float2* S;  // n/2 + 1 bins are kept: the frame is real, so the spectrum is exactly Hermitian
	            // (radix-2 DAG + a twiddle table with tw[n/2-j] == -conj(tw[j]) bit for bit) and the
	            // synthesis kernel rebuilds S[n-k] = conj(S[k]).  Halves the spectrum traffic.
	float* mag; // all n bins: the frequency median runs over the full spectrum
	int n;
	__device__ __forceinline__ void operator()(int idx, float2 X, bool lower, int) const
	{
		if (lower || idx == (n >> 1))
			S[idx] = X;
		mag[idx] = zfft::cabs_exact(X.x, X.y); // complex_abs_functor hps.h:82-89
	}
};

template <int LOG2N>
__global__ __launch_bounds__(Plan<LOG2N>::THREADS) void stft_kernel(StftArgs a)
{
	using PL = Plan<LOG2N>;
	extern __shared__ float2 lds[];
	const int tid = threadIdx.x, s = blockIdx.y, hop = a.hop;
	if (blockIdx.x == gridDim.x - 1) { // housekeeping block
		const float* last = a.in + (long long)s * a.in_stride + (long long)(a.n_frames - 1) * hop;
		for (int i = tid; i < hop; i += PL::THREADS)
			a.tail_next[(long long)s * hop + i] = last[i];
		if (a.prev_frames > 0) {
			for (int o = 0; o < 3; ++o) {
				if (!a.carry[o])
					continue;
				const float* y = a.Y[o] + (long long)s * a.y_stream_stride
				                 + (long long)(a.prev_frames - 1) * (2 * hop) + hop;
				for (int i = tid; i < hop; i += PL::THREADS)
					a.carry[o][(long long)s * hop + i] = y[i];
			}
		}
		return;
	}
	const int slot = tid / PL::TF, tf = tid - slot * PL::TF;
	const int f = blockIdx.x * PL::FRAMES_PER_BLOCK + slot;
	const bool active = f < a.n_frames;
	const float* in_s = a.in + (long long)s * a.in_stride;
	StftIn<LOG2N> in;
	in.prev = (f == 0) ? a.tail_prev + (long long)s * hop : in_s + (long long)(f - 1) * hop;
	in.cur = in_s + (long long)f * hop;
	in.window = a.window;
	in.hop = hop;
	const long long row = ((a.row0 + f) % a.ring_rows) + (long long)s * a.ring_rows;
	StftOut out;
	out.S = a.S + row * a.s_stride;
	out.mag = a.mag + row * PL::N;
	out.n = PL::N;
	zfft::fft_frame<LOG2N, false, true, false>(tf, lds + slot * PL::LDS_FLOAT2, a.tw, in, out, active);
}

// ------------------------------------------------------------------------------------------------
struct IstftIn {
	const float2* S;
	const float* H;
	const float* P;
	MaskCfg cfg;
	int which;
	int n;
	__device__ __forceinline__ float2 operator()(int idx, int) const
	{
		const bool mirror = idx > (n >> 1); // upper half: S[n-k] = conj(S[k])
		float2 z = S[mirror ? n - idx : idx];
		if (mirror)
			z.y = -z.y;
		const float m = mask_value(which, H[idx], P[idx], cfg);
		return make_float2(z.x * m, z.y * m); // apply_mask_functor hps.h:58-66
	}
};

struct IstftOut {
	float* Y;
	float cola;
	__device__ __forceinline__ void operator()(int idx, float2 x, bool, int) const
	{
		Y[idx] = x.x * cola; // the product of overlap_add_functor hps.h:68-80; the sum is in finalize
	}
};

template <int LOG2N>
__global__ __launch_bounds__(Plan<LOG2N>::THREADS) void istft_kernel(IstftArgs a)
{
	using PL = Plan<LOG2N>;
	extern __shared__ float2 lds[];
	const int tid = threadIdx.x, s = blockIdx.z, oi = blockIdx.y;
	const int slot = tid / PL::TF, tf = tid - slot * PL::TF;
	const int f = blockIdx.x * PL::FRAMES_PER_BLOCK + slot;
	const bool active = f < a.n_frames;
	const long long ring_row = ((a.crow0 + f) % a.ring_rows) + (long long)s * a.ring_rows;
	IstftIn in;
	in.S = a.S + ring_row * a.s_stride;
	in.n = PL::N;
	in.H = a.h_is_ring ? a.H + ring_row * PL::N : a.H + (long long)s * a.h_stream_stride + (long long)f * PL::N;
	in.P = a.P + (long long)s * a.p_stream_stride + (long long)f * PL::N;
	in.cfg = MaskCfg{a.beta, a.beta_h, a.soft, a.power, a.sse, a.out_h, a.out_p};
	in.which = a.out_id[oi];
	IstftOut out;
	out.Y = a.Y[oi] + (long long)s * a.y_stream_stride + (long long)f * (PL::N / 2);
	out.cola = a.cola;
	zfft::fft_frame<LOG2N, true, false, true>(tf, lds + slot * PL::LDS_FLOAT2, a.tw, in, out, active);
}

// Hard masks with more than one output (HPRIOffline pass 1: H, P and R of every frame): one workgroup
// synthesises all outputs of its frame.  The two binary masks of a bin are compared once, while the first
// output loads S, H and P, and kept as two bits per bin in one register; the other outputs re-read only S
// (from L2) instead of S, H and P, and repeat no division.  Same values as mask_value().
struct IstftHardIn {
	const float2* S;
	const float* H;
	const float* P;
	MaskCfg cfg;
	unsigned* bits; // bit 2*slot: percussive mask, bit 2*slot + 1: harmonic mask
	int which;
	int first;
	int n;
	__device__ __forceinline__ float2 operator()(int idx, int slot) const
	{
		const bool mirror = idx > (n >> 1);
		float2 z = S[mirror ? n - idx : idx];
		if (mirror)
			z.y = -z.y;
		if (first) {
			const float h = H[idx], p = P[idx];
			const unsigned pm = cfg.out_p || which == 0 ? (unsigned)(pmask_value(h, p, cfg) != 0.0f) : 0u;
			const unsigned hm = cfg.out_h || which == 1 ? (unsigned)(hmask_value(h, p, cfg) != 0.0f) : 0u;
			*bits |= (pm | (hm << 1)) << (2 * slot);
		}
		const float pm = (float)((*bits >> (2 * slot)) & 1u), hm = (float)((*bits >> (2 * slot + 1)) & 1u);
		const float m = which == 0 ? pm : (which == 1 ? hm : 1 - (hm + pm)); // residual_mask_functor hps.h:35-43
		return make_float2(z.x * m, z.y * m);
	}
};

template <int LOG2N>
__global__ __launch_bounds__(Plan<LOG2N>::THREADS) void istft_hard_multi_kernel(IstftArgs a)
{
	using PL = Plan<LOG2N>;
	extern __shared__ float2 lds[];
	const int tid = threadIdx.x, s = blockIdx.z;
	const int slot = tid / PL::TF, tf = tid - slot * PL::TF;
	const int f = blockIdx.x * PL::FRAMES_PER_BLOCK + slot;
	const bool active = f < a.n_frames;
	const long long ring_row = ((a.crow0 + f) % a.ring_rows) + (long long)s * a.ring_rows;
	unsigned bits = 0;
	IstftHardIn in;
	in.S = a.S + ring_row * a.s_stride;
	in.n = PL::N;
	in.H = a.h_is_ring ? a.H + ring_row * PL::N : a.H + (long long)s * a.h_stream_stride + (long long)f * PL::N;
	in.P = a.P + (long long)s * a.p_stream_stride + (long long)f * PL::N;
	in.cfg = MaskCfg{a.beta, a.beta_h, 0, a.power, 0, a.out_h, a.out_p};
	in.bits = &bits;
	for (int oi = 0; oi < a.n_out; ++oi) {
		in.which = a.out_id[oi];
		in.first = oi == 0;
		IstftOut out;
		out.Y = a.Y[oi] + (long long)s * a.y_stream_stride + (long long)f * (PL::N / 2);
		out.cola = a.cola;
		// The thread index and the table pointer are made opaque per output: otherwise every LDS address and
		// twiddle index of the transform (all functions of tf alone) is hoisted out of this loop and kept
		// in registers (226 VGPRs instead of 90 at nfft 4096, 128 spilled at nfft 16384).
		int tf_o = tf;
		const float2* tw_o = a.tw;
		asm volatile("" : "+v"(tf_o));
		asm volatile("" : "+s"(tw_o));
		zfft::fft_frame<LOG2N, true, false, true>(tf_o, lds + slot * PL::LDS_FLOAT2, tw_o, in, out, active);
		__syncthreads(); // the frame image is reused by the next output
	}
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs a)
{
	const int s = blockIdx.y, hop = a.hop;
	const float* Y = a.Y + (long long)s * a.y_stream_stride;
	const float* carry = a.carry + (long long)s * hop;
	float* out = a.out + (long long)s * a.out_stride;
	const long long n = (long long)a.n_frames * hop;
	for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
	     e += (long long)gridDim.x * blockDim.x) {
		const long long i = e / hop;
		const int k = (int)(e - i * hop);
		const float prev = (i == 0) ? carry[k] : Y[(i - 1) * 2 * hop + hop + k];
		out[e] = prev + Y[i * 2 * hop + k];
	}
}

// ------------------------------------------------------------------------------------------------
struct PlainIn {
	const float2* d;
	__device__ __forceinline__ float2 operator()(int idx, int) const { return d[idx]; }
};
struct PlainOut {
	float2* d;
	__device__ __forceinline__ void operator()(int idx, float2 X, bool, int) const { d[idx] = X; }
};

template <int LOG2N, bool INV>
__global__ __launch_bounds__(Plan<LOG2N>::THREADS) void fft_kernel(float2* data, const float2* tw, int batch)
{
	using PL = Plan<LOG2N>;
	extern __shared__ float2 lds[];
	const int tid = threadIdx.x;
	const int slot = tid / PL::TF, tf = tid - slot * PL::TF;
	const long long f = (long long)blockIdx.x * PL::FRAMES_PER_BLOCK + slot;
	const bool active = f < batch;
	PlainIn in{data + f * PL::N};
	PlainOut out{data + f * PL::N};
	zfft::fft_frame<LOG2N, INV, false, false>(tf, lds + slot * PL::LDS_FLOAT2, tw, in, out, active);
}

template <int LOG2N>
constexpr size_t lds_bytes()
{
	return sizeof(float2) * (size_t)Plan<LOG2N>::LDS_FLOAT2 * Plan<LOG2N>::FRAMES_PER_BLOCK;
}

template <class K>
int set_lds(K kern, size_t bytes)
{
	if (bytes > 64 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
	return ZEN_HIP_OK;
}

template <int LOG2N>
int launch_stft_t(const StftArgs& a, hipStream_t stream)
{
	using PL = Plan<LOG2N>;
	auto kern = stft_kernel<LOG2N>;
	ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
	dim3 grid((unsigned)ceil_div(a.n_frames, PL::FRAMES_PER_BLOCK) + 1, (unsigned)a.n_streams);
	hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int LOG2N>
int launch_istft_t(const IstftArgs& a, hipStream_t stream)
{
	using PL = Plan<LOG2N>;
	// all outputs of a frame in one workgroup; not at nfft 8192/16384, where a CU holds one or two frames and
	// three short workgroups per frame schedule better than one long one (measured)
	if (LOG2N <= 12 && a.n_out > 1 && !a.soft && !a.sse && !g_opt_no_istft_multi) {
		auto kern = istft_hard_multi_kernel<LOG2N>;
		ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
		dim3 grid((unsigned)ceil_div(a.n_frames, PL::FRAMES_PER_BLOCK), 1, (unsigned)a.n_streams);
		hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, a);
		ZH_HIP(hipGetLastError());
		return ZEN_HIP_OK;
	}
	auto kern = istft_kernel<LOG2N>;
	ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
	dim3 grid((unsigned)ceil_div(a.n_frames, PL::FRAMES_PER_BLOCK), (unsigned)a.n_out, (unsigned)a.n_streams);
	hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int LOG2N>
int launch_fft_t(float2* data, const float2* tw, size_t batch, int inverse, hipStream_t stream)
{
	using PL = Plan<LOG2N>;
	dim3 grid((unsigned)ceil_div(batch, PL::FRAMES_PER_BLOCK));
	if (inverse) {
		auto kern = fft_kernel<LOG2N, true>;
		ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
		hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, data, tw, (int)batch);
	}
	else {
		auto kern = fft_kernel<LOG2N, false>;
		ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
		hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, data, tw, (int)batch);
	}
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace

#define ZH_DISPATCH_LOG2N(log2n, CALL)                                                     \
	switch (log2n) {                                                                       \
	case 5: return CALL(5);                                                                \
	case 6: return CALL(6);                                                                \
	case 7: return CALL(7);                                                                \
	case 8: return CALL(8);                                                                \
	case 9: return CALL(9);                                                                \
	case 10: return CALL(10);                                                              \
	case 11: return CALL(11);                                                              \
	case 12: return CALL(12);                                                              \
	case 13: return CALL(13);                                                              \
	case 14: return CALL(14);                                                              \
	default:                                                                               \
		ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "nfft = 2^%d outside the supported 32..16384", log2n); \
	}

int launch_stft(int log2n, const StftArgs& a, hipStream_t stream)
{
	if (a.n_frames <= 0)
		return ZEN_HIP_OK;
#define CALL(L) launch_stft_t<L>(a, stream)
	ZH_DISPATCH_LOG2N(log2n, CALL)
#undef CALL
}

int launch_istft(int log2n, const IstftArgs& a, hipStream_t stream)
{
	if (a.n_frames <= 0 || a.n_out <= 0)
		return ZEN_HIP_OK;
#define CALL(L) launch_istft_t<L>(a, stream)
	ZH_DISPATCH_LOG2N(log2n, CALL)
#undef CALL
}

int launch_fft(int log2n, float2* data, const float2* tw, size_t batch, int inverse, hipStream_t stream)
{
	if (batch == 0)
		return ZEN_HIP_OK;
#define CALL(L) launch_fft_t<L>(data, tw, batch, inverse, stream)
	ZH_DISPATCH_LOG2N(log2n, CALL)
#undef CALL
}

int launch_finalize(const FinalizeArgs& a, hipStream_t stream)
{
	if (a.n_frames <= 0)
		return ZEN_HIP_OK;
	const long long n = (long long)a.n_frames * a.hop;
	long long blocks = (n + 255) / 256;
	if (blocks > 4096)
		blocks = 4096;
	hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)blocks, (unsigned)a.n_streams), dim3(256), 0, stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace zen_hip_impl
