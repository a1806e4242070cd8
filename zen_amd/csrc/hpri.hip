// hpri.hip -- HPRIOffline<Backend::GPU> behind the C-ABI: two cascaded HPR passes ("HPR-I") on two streaming
// engines (hpr.hip), everything resident in HBM.
//
//   zen_hip_hpri_* : HPRIOffline<GPU>::process (libzen/hps.cu:128-221): pass 1 (hop_h; H, P, R) ->
//                    P+R shifted by lag_h*hop_h -> pass 2 (hop_p; P).  The reference walks the clip hop by hop
//                    through two IOGPU buffers (hps.cu:146-166,189-204); here each pass is a handful of
//                    block calls on the engine.  Also: a batch of equal-length clips per call, and any time
//                    range of one clip from its own warm-up halo.
#include "common.h"
#include "hpr_engine.h"

#include <cfloat>
#include <climits>
#include <cmath>

using namespace zen_hip_impl;

// =================================================================================================
// HPRIOffline<GPU>
// =================================================================================================
struct zen_hip_hpri {
	zen_hip_hpr* eh = nullptr; // p_impl_h : hop_h, H+P+R, anticausal (hps.cu:38-43)
	zen_hip_hpr* ep = nullptr; // p_impl_p : hop_p, P only, anticausal (hps.cu:45-48)
	size_t hop_h, hop_p, n_clips;
	hipStream_t stream = nullptr;
	// scratch, grown on demand
	size_t cap1 = 0, cap2 = 0;
	float *a1 = nullptr, *H1 = nullptr, *P1 = nullptr, *R1 = nullptr, *in2 = nullptr, *P2 = nullptr;
	float *stage_in = nullptr, *stage_out[3] = {nullptr, nullptr, nullptr};
	size_t stage_cap = 0;
};

namespace {

// hps.cu:109-126 hpss_chunk_padder: float ceil of a float quotient, plus `lag` chunks
int chunk_padder(size_t audio_size, size_t hop, size_t lag, size_t* padded)
{
	int n = (int)(ceilf((float)audio_size / (float)hop));
	n += (int)lag;
	*padded = (size_t)n * hop;
	return n;
}

// The three streaming helpers below move four samples per thread: one 16-byte access where the four come from one
// aligned run (VEC: the caller's rows are 16-byte aligned; the engine's own buffers always are), sample by sample
// at the seams and the ends.
template <bool VEC>
__global__ __launch_bounds__(256) void pad_clips_kernel(const float* __restrict__ in, long long in_stride,
                                                        size_t n, float* __restrict__ out, size_t padded)
{
	const float* src = in + (long long)blockIdx.y * in_stride;
	float* dst = out + (size_t)blockIdx.y * padded; // (padded is a multiple of the hop: rows stay aligned)
	for (size_t i = 4 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x); i < padded; i += 4 * (size_t)gridDim.x * blockDim.x) {
		if (VEC && i + 4 <= n) {
			*reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
		}
		else {
			for (size_t k = i; k < i + 4 && k < padded; ++k)
				dst[k] = k < n ? src[k] : 0.0F; // audio.resize(size + pad, 0.0F)  hps.cu:123
		}
	}
}

// hps.cu:153-160 (xp1 + xr1), :171-176 (shift left by lag_h*hop_h in place; the tail keeps its old
// contents) and :186-190 (pass 2 reads `intermediate` up to n2*hop_p, past size() -- SURVEY Q9).
__global__ __launch_bounds__(256) void intermediate_kernel(const float* __restrict__ P1, const float* __restrict__ R1,
                                                           size_t padded1, size_t sh1, float* __restrict__ in2,
                                                           size_t padded2)
{
	const float* p = P1 + (size_t)blockIdx.y * padded1;
	const float* r = R1 + (size_t)blockIdx.y * padded1;
	float* dst = in2 + (size_t)blockIdx.y * padded2;
	for (size_t j = 4 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x); j < padded2; j += 4 * (size_t)gridDim.x * blockDim.x) {
		if (j + 4 <= padded1 - sh1 && j + 4 <= padded2) { // (padded1, padded2 and sh1 are multiples of a hop: aligned)
			const float4 x = *reinterpret_cast<const float4*>(p + j + sh1), y = *reinterpret_cast<const float4*>(r + j + sh1);
			*reinterpret_cast<float4*>(dst + j) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); // sum_vectors_functor hps.h:142-150
		}
		else {
			for (size_t k = j; k < j + 4 && k < padded2; ++k) {
				float v = 0.0F; // beyond the reference's allocation (undefined there)
				if (k < padded1) {
					const size_t q = (k < padded1 - sh1) ? k + sh1 : k;
					v = p[q] + r[q];
				}
				dst[k] = v;
			}
		}
	}
}

// hps.cu:171-178, :209-217 : drop the lag*hop delay, truncate to the clip length
template <bool VEC>
__global__ __launch_bounds__(256) void unshift_kernel(const float* __restrict__ full, size_t padded, size_t sh,
                                                      float* __restrict__ out, long long out_stride, size_t n)
{
	const float* src = full + (size_t)blockIdx.y * padded;
	float* dst = out + (long long)blockIdx.y * out_stride;
	for (size_t j = 4 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x); j < n; j += 4 * (size_t)gridDim.x * blockDim.x) {
		if (VEC && j + 4 <= n && j + 4 <= padded - sh) {
			*reinterpret_cast<float4*>(dst + j) = *reinterpret_cast<const float4*>(src + j + sh);
		}
		else {
			for (size_t k = j; k < j + 4 && k < n; ++k) {
				float v = 0.0F;
				if (k < padded)
					v = (k < padded - sh) ? src[k + sh] : src[k];
				dst[k] = v;
			}
		}
	}
}

// rows of a caller's buffer on which 16-byte accesses are allowed
bool rows_aligned(const void* base, long long stride_floats, size_t rows)
{
	return (reinterpret_cast<uintptr_t>(base) & 15) == 0 && (rows <= 1 || stride_floats % 4 == 0);
}

unsigned grid_for4(size_t n) // four samples per thread
{
	size_t b = (n + 1023) / 1024;
	return (unsigned)(b > 4096 ? 4096 : (b ? b : 1));
}

unsigned grid_for(size_t n)
{
	size_t b = (n + 255) / 256;
	return (unsigned)(b > 4096 ? 4096 : (b ? b : 1));
}

void hpri_free_scratch(zen_hip_hpri* o)
{
	(void)hipFree(o->a1);
	(void)hipFree(o->H1);
	(void)hipFree(o->P1);
	(void)hipFree(o->R1);
	(void)hipFree(o->in2);
	(void)hipFree(o->P2);
	o->a1 = o->H1 = o->P1 = o->R1 = o->in2 = o->P2 = nullptr;
	o->cap1 = o->cap2 = 0;
}

} // namespace

extern "C" {

int zen_hip_hpri_create(float fs, size_t hop_h, size_t hop_p, float beta_h, float beta_p, int nocopybord,
                        size_t n_clips, zen_hip_hpri_t* h)
{
	if (!h || n_clips == 0)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_create: null handle or zero clips");
	if (hop_p == 0 || hop_h % hop_p != 0) // hps.cu:33-36
		ZH_FAIL(ZEN_HIP_E_HOPS_NOT_DIVISIBLE, "hop_h and hop_p should be evenly divisible");
	zen_hip_hpri* o = new zen_hip_hpri;
	o->hop_h = hop_h;
	o->hop_p = hop_p;
	o->n_clips = n_clips;
	int rc = zen_hip_hpr_create(fs, hop_h, beta_h,
	                            ZEN_HIP_OUTPUT_HARMONIC | ZEN_HIP_OUTPUT_PERCUSSIVE | ZEN_HIP_OUTPUT_RESIDUAL,
	                            ZEN_HIP_TIME_ANTICAUSAL, !nocopybord, n_clips, 0, &o->eh);
	if (rc == ZEN_HIP_OK)
		rc = zen_hip_hpr_create(fs, hop_p, beta_p, ZEN_HIP_OUTPUT_PERCUSSIVE, ZEN_HIP_TIME_ANTICAUSAL,
		                        !nocopybord, n_clips, 0, &o->ep);
	if (rc != ZEN_HIP_OK) {
		zen_hip_hpr_destroy(o->eh);
		delete o;
		return rc;
	}
	*h = o;
	return ZEN_HIP_OK;
}

int zen_hip_hpri_destroy(zen_hip_hpri_t h)
{
	if (h) {
		zen_hip_hpr_destroy(h->eh);
		zen_hip_hpr_destroy(h->ep);
		hpri_free_scratch(h);
		(void)hipFree(h->stage_in);
		for (int i = 0; i < 3; ++i)
			(void)hipFree(h->stage_out[i]);
		delete h;
	}
	return ZEN_HIP_OK;
}

int zen_hip_hpri_set_stream(zen_hip_hpri_t h, void* stream)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null handle");
	ZH_TRY(zen_hip_hpr_set_stream(h->eh, stream));
	ZH_TRY(zen_hip_hpr_set_stream(h->ep, stream));
	h->stream = (hipStream_t)stream;
	return ZEN_HIP_OK;
}

int zen_hip_hpri_use_sse_filter(zen_hip_hpri_t h)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null handle");
	zen_hip_hpr_use_sse_filter(h->eh);
	zen_hip_hpr_use_sse_filter(h->ep);
	return ZEN_HIP_OK;
}

int zen_hip_hpri_use_soft_mask(zen_hip_hpri_t h)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null handle");
	zen_hip_hpr_use_soft_mask(h->eh);
	zen_hip_hpr_use_soft_mask(h->ep);
	return ZEN_HIP_OK;
}

int zen_hip_hpri_hop_counts(zen_hip_hpri_t h, size_t n, size_t* n_hops_h, size_t* n_hops_p)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null handle");
	size_t p1, p2;
	const int n1 = chunk_padder(n, h->hop_h, (size_t)h->eh->lag, &p1);
	const int n2 = chunk_padder(n, h->hop_p, (size_t)h->ep->lag, &p2);
	if (n_hops_h)
		*n_hops_h = (size_t)n1;
	if (n_hops_p)
		*n_hops_p = (size_t)n2;
	return ZEN_HIP_OK;
}

int zen_hip_hpri_process_device(zen_hip_hpri_t h, const float* audio_dev, size_t n, size_t stride,
                                float* harm_dev, float* perc_dev, float* resid_dev, size_t out_stride)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process: null handle");
	if (n == 0)
		return ZEN_HIP_OK; // empty clips: nothing to write
	if (!audio_dev)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process: null argument");
	const size_t C = h->n_clips;
	size_t padded1, padded2;
	const int n1 = chunk_padder(n, h->hop_h, (size_t)h->eh->lag, &padded1); // hps.cu:133-134
	const int n2 = chunk_padder(n, h->hop_p, (size_t)h->ep->lag, &padded2); // hps.cu:180-181
	if (n1 <= 0 || n2 <= 0)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process: clip too short");
	if (padded1 > h->cap1 || padded2 > h->cap2) {
		ZH_HIP(hipStreamSynchronize(h->stream));
		hpri_free_scratch(h);
		ZH_HIP(hipMalloc((void**)&h->a1, sizeof(float) * C * padded1));
		ZH_HIP(hipMalloc((void**)&h->H1, sizeof(float) * C * padded1));
		ZH_HIP(hipMalloc((void**)&h->P1, sizeof(float) * C * padded1));
		ZH_HIP(hipMalloc((void**)&h->R1, sizeof(float) * C * padded1));
		ZH_HIP(hipMalloc((void**)&h->in2, sizeof(float) * C * padded2));
		ZH_HIP(hipMalloc((void**)&h->P2, sizeof(float) * C * padded2));
		h->cap1 = padded1;
		h->cap2 = padded2;
	}
	ZH_TRY(zen_hip_hpr_reset_buffers(h->eh)); // each process() call is a fresh pair of HPR objects' state
	ZH_TRY(zen_hip_hpr_reset_buffers(h->ep));

	if (rows_aligned(audio_dev, (long long)stride, C))
		hipLaunchKernelGGL(pad_clips_kernel<true>, dim3(grid_for4(padded1), (unsigned)C), dim3(256), 0, h->stream, audio_dev,
		                   (long long)stride, n, h->a1, padded1);
	else
		hipLaunchKernelGGL(pad_clips_kernel<false>, dim3(grid_for4(padded1), (unsigned)C), dim3(256), 0, h->stream, audio_dev,
		                   (long long)stride, n, h->a1, padded1);
	ZH_HIP(hipGetLastError());
	// pass 1: large hop, harmonic + percussive + residual (hps.cu:142-167)
	ZH_TRY(zen_hip_hpr_process(h->eh, h->a1, (size_t)n1, padded1, h->H1, h->P1, h->R1, padded1));
	const size_t sh1 = (size_t)h->eh->lag * h->hop_h;
	hipLaunchKernelGGL(intermediate_kernel, dim3(grid_for4(padded2), (unsigned)C), dim3(256), 0, h->stream, h->P1,
	                   h->R1, padded1, sh1, h->in2, padded2);
	ZH_HIP(hipGetLastError());
	// pass 2: small hop on xp1 + xr1, percussive only (hps.cu:185-205)
	ZH_TRY(zen_hip_hpr_process(h->ep, h->in2, (size_t)n2, padded2, nullptr, h->P2, nullptr, padded2));
	const size_t sh2 = (size_t)h->ep->lag * h->hop_p;
	if (harm_dev) {
		if (rows_aligned(harm_dev, (long long)out_stride, C))
			hipLaunchKernelGGL(unshift_kernel<true>, dim3(grid_for4(n), (unsigned)C), dim3(256), 0, h->stream, h->H1, padded1,
			                   sh1, harm_dev, (long long)out_stride, n);
		else
			hipLaunchKernelGGL(unshift_kernel<false>, dim3(grid_for4(n), (unsigned)C), dim3(256), 0, h->stream, h->H1, padded1,
			                   sh1, harm_dev, (long long)out_stride, n);
		ZH_HIP(hipGetLastError());
	}
	if (perc_dev) {
		if (rows_aligned(perc_dev, (long long)out_stride, C))
			hipLaunchKernelGGL(unshift_kernel<true>, dim3(grid_for4(n), (unsigned)C), dim3(256), 0, h->stream, h->P2, padded2,
			                   sh2, perc_dev, (long long)out_stride, n);
		else
			hipLaunchKernelGGL(unshift_kernel<false>, dim3(grid_for4(n), (unsigned)C), dim3(256), 0, h->stream, h->P2, padded2,
			                   sh2, perc_dev, (long long)out_stride, n);
		ZH_HIP(hipGetLastError());
	}
	if (resid_dev) // pass 2's residual_out is never written: zeros (hps.cu:45-48, :200-204)
		ZH_HIP(hipMemset2DAsync(resid_dev, sizeof(float) * out_stride, 0, sizeof(float) * n, C, h->stream));
	return ZEN_HIP_OK;
}

// ---- time-sharding one long clip ---------------------------------------------------
// Output samples [begin, end) of an n-sample clip, bit-identical to the same range of
// zen_hip_hpri_process.  Both passes are streaming recurrences whose state (input tail, the last W-1
// spectra, the overlap-add carry) is a function of the last <= W+1 hops only, so a shard that starts
// 2W+2 hops early from zero state reaches exactly the serial state before its first kept sample.  The
// shard therefore needs only a halo of input: (2W_p+2)*hop_p + lag_p*hop_p for pass 2 on top of
// (2W_h+2)*hop_h + lag_h*hop_h for pass 1; ranks exchange nothing.
extern "C++" {
namespace {

struct RangePlan {
	size_t padded1, padded2, sh1, sh2;
	size_t q1, k1;   // pass-1 hops [q1, k1) are run
	size_t q2, m1;   // pass-2 hops [q2, m1) are run
	size_t in_begin, in_end; // input samples read (clipped to n; beyond n is zero padding)
};

int plan_range(zen_hip_hpri* h, size_t n, size_t begin, size_t end, RangePlan* p)
{
	if (!(begin < end) || end > n)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri range [%zu, %zu) outside the clip of %zu samples", begin, end, n);
	const size_t hop_h = h->hop_h, hop_p = h->hop_p;
	const int n1 = chunk_padder(n, hop_h, (size_t)h->eh->lag, &p->padded1);
	const int n2 = chunk_padder(n, hop_p, (size_t)h->ep->lag, &p->padded2);
	if (n1 <= 0 || n2 <= 0)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri range: clip too short");
	p->sh1 = (size_t)h->eh->lag * hop_h;
	p->sh2 = (size_t)h->ep->lag * hop_p;
	const size_t warm_p = 2 * h->ep->W + 2, warm_h = 2 * h->eh->W + 2;
	// pass-2 output positions [begin + sh2, end + sh2) (or unshifted near the very end: Q9-style tail)
	size_t m0 = begin / hop_p;                       // conservative: positions j or j + sh2
	p->m1 = ceil_div(end + p->sh2, hop_p);
	if (p->m1 > (size_t)n2)
		p->m1 = (size_t)n2;
	if (m0 > p->m1)
		m0 = p->m1;
	p->q2 = m0 > warm_p ? m0 - warm_p : 0;
	// pass-1 output positions: pass-2 input j in [q2*hop_p, m1*hop_p) reads position j or j + sh1;
	// the harmonic output reads [begin, end + sh1)
	size_t a1 = p->q2 * hop_p < begin ? p->q2 * hop_p : begin;
	size_t b1 = p->m1 * hop_p + p->sh1;
	if (end + p->sh1 > b1)
		b1 = end + p->sh1;
	if (b1 > p->padded1)
		b1 = p->padded1;
	const size_t k0 = a1 / hop_h;
	p->k1 = ceil_div(b1, hop_h);
	if (p->k1 > (size_t)n1)
		p->k1 = (size_t)n1;
	p->q1 = k0 > warm_h ? k0 - warm_h : 0;
	p->in_begin = p->q1 * hop_h;
	p->in_end = p->k1 * hop_h < n ? p->k1 * hop_h : n;
	if (p->in_begin > p->in_end)
		p->in_begin = p->in_end;
	return ZEN_HIP_OK;
}

// a1[i] = audio[off + i] (zero beyond n), i < count
// (the three range kernels move four samples per thread like the whole-clip helpers above; VEC: the launcher found
// source and destination 16-byte aligned for every group of four)
template <bool VEC>
__global__ __launch_bounds__(256) void range_input_kernel(const float* __restrict__ audio, size_t n, size_t off,
                                                          float* __restrict__ dst, size_t count)
{
	for (size_t i = 4 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x); i < count; i += 4 * (size_t)gridDim.x * blockDim.x) {
		if (VEC && i + 4 <= count && off + i + 4 <= n) {
			*reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(audio + off + i);
		}
		else {
			for (size_t k = i; k < i + 4 && k < count; ++k)
				dst[k] = (off + k < n) ? audio[off + k] : 0.0F;
		}
	}
}

// pass-2 input positions j in [j0, j0+count): intermediate'[j] as intermediate_kernel defines it;
// P1/R1 hold pass-1 output positions [base1, ...)
template <bool VEC>
__global__ __launch_bounds__(256) void range_intermediate_kernel(const float* __restrict__ P1, const float* __restrict__ R1,
                                                                 size_t base1, size_t padded1, size_t sh1, size_t j0,
                                                                 float* __restrict__ dst, size_t count)
{
	for (size_t i = 4 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x); i < count; i += 4 * (size_t)gridDim.x * blockDim.x) {
		const size_t j = j0 + i;
		if (VEC && i + 4 <= count && j + 4 <= padded1 - sh1) {
			const float4 x = *reinterpret_cast<const float4*>(P1 + (j + sh1 - base1)), y = *reinterpret_cast<const float4*>(R1 + (j + sh1 - base1));
			*reinterpret_cast<float4*>(dst + i) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
		}
		else {
			for (size_t k = i; k < i + 4 && k < count; ++k) {
				const size_t jk = j0 + k;
				float v = 0.0F;
				if (jk < padded1) {
					const size_t q = (jk < padded1 - sh1) ? jk + sh1 : jk;
					v = P1[q - base1] + R1[q - base1];
				}
				dst[k] = v;
			}
		}
	}
}

// out[i] = full'[begin + i] where full' is `full` with the lag*hop delay removed (unshift_kernel);
// `full` holds positions [base, ...)
template <bool VEC>
__global__ __launch_bounds__(256) void range_unshift_kernel(const float* __restrict__ full, size_t base, size_t padded,
                                                            size_t sh, size_t begin, float* __restrict__ out, size_t count)
{
	for (size_t i = 4 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x); i < count; i += 4 * (size_t)gridDim.x * blockDim.x) {
		const size_t j = begin + i;
		if (VEC && i + 4 <= count && j + 4 <= padded - sh) {
			*reinterpret_cast<float4*>(out + i) = *reinterpret_cast<const float4*>(full + (j + sh - base));
		}
		else {
			for (size_t k = i; k < i + 4 && k < count; ++k) {
				const size_t jk = begin + k;
				float v = 0.0F;
				if (jk < padded) {
					const size_t q = (jk < padded - sh) ? jk + sh : jk;
					v = full[q - base];
				}
				out[k] = v;
			}
		}
	}
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

} // namespace
} // extern "C++"

int zen_hip_hpri_range_halo(zen_hip_hpri_t h, size_t n, size_t begin, size_t end, size_t* in_begin, size_t* in_end)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null handle");
	RangePlan p;
	ZH_TRY(plan_range(h, n, begin, end, &p));
	if (in_begin)
		*in_begin = p.in_begin;
	if (in_end)
		*in_end = p.in_end;
	return ZEN_HIP_OK;
}

int zen_hip_hpri_process_range(zen_hip_hpri_t h, const float* audio_dev, size_t n, size_t begin, size_t end,
                               float* harm_dev, float* perc_dev)
{
	if (!h || !audio_dev)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process_range: null argument");
	if (h->n_clips != 1)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process_range needs a handle created with n_clips == 1");
	RangePlan p;
	ZH_TRY(plan_range(h, n, begin, end, &p));
	const size_t hop_h = h->hop_h, hop_p = h->hop_p;
	const size_t c1 = (p.k1 - p.q1) * hop_h, c2 = (p.m1 - p.q2) * hop_p;
	if (c1 > h->cap1 || c2 > h->cap2) {
		ZH_HIP(hipStreamSynchronize(h->stream));
		hpri_free_scratch(h);
		ZH_HIP(hipMalloc((void**)&h->a1, sizeof(float) * c1));
		ZH_HIP(hipMalloc((void**)&h->H1, sizeof(float) * c1));
		ZH_HIP(hipMalloc((void**)&h->P1, sizeof(float) * c1));
		ZH_HIP(hipMalloc((void**)&h->R1, sizeof(float) * c1));
		ZH_HIP(hipMalloc((void**)&h->in2, sizeof(float) * (c2 ? c2 : 1)));
		ZH_HIP(hipMalloc((void**)&h->P2, sizeof(float) * (c2 ? c2 : 1)));
		h->cap1 = c1;
		h->cap2 = c2;
	}
	ZH_TRY(zen_hip_hpr_reset_buffers(h->eh));
	ZH_TRY(zen_hip_hpr_reset_buffers(h->ep));
	const size_t base1 = p.q1 * hop_h, base2 = p.q2 * hop_p;
	if (aligned16(audio_dev + base1))
		hipLaunchKernelGGL(range_input_kernel<true>, dim3(grid_for4(c1)), dim3(256), 0, h->stream, audio_dev, n, base1, h->a1, c1);
	else
		hipLaunchKernelGGL(range_input_kernel<false>, dim3(grid_for4(c1)), dim3(256), 0, h->stream, audio_dev, n, base1, h->a1, c1);
	ZH_HIP(hipGetLastError());
	ZH_TRY(zen_hip_hpr_process(h->eh, h->a1, p.k1 - p.q1, c1, h->H1, h->P1, h->R1, c1));
	if (c2) {
		if ((base2 + p.sh1 - base1) % 4 == 0)
			hipLaunchKernelGGL(range_intermediate_kernel<true>, dim3(grid_for4(c2)), dim3(256), 0, h->stream, h->P1, h->R1, base1,
			                   p.padded1, p.sh1, base2, h->in2, c2);
		else
			hipLaunchKernelGGL(range_intermediate_kernel<false>, dim3(grid_for4(c2)), dim3(256), 0, h->stream, h->P1, h->R1, base1,
			                   p.padded1, p.sh1, base2, h->in2, c2);
		ZH_HIP(hipGetLastError());
		ZH_TRY(zen_hip_hpr_process(h->ep, h->in2, p.m1 - p.q2, c2, nullptr, h->P2, nullptr, c2));
	}
	const size_t cnt = end - begin;
	if (harm_dev) {
		if (aligned16(harm_dev) && (begin + p.sh1 - base1) % 4 == 0)
			hipLaunchKernelGGL(range_unshift_kernel<true>, dim3(grid_for4(cnt)), dim3(256), 0, h->stream, h->H1, base1, p.padded1,
			                   p.sh1, begin, harm_dev, cnt);
		else
			hipLaunchKernelGGL(range_unshift_kernel<false>, dim3(grid_for4(cnt)), dim3(256), 0, h->stream, h->H1, base1, p.padded1,
			                   p.sh1, begin, harm_dev, cnt);
		ZH_HIP(hipGetLastError());
	}
	if (perc_dev) {
		if (aligned16(perc_dev) && (begin + p.sh2 - base2) % 4 == 0)
			hipLaunchKernelGGL(range_unshift_kernel<true>, dim3(grid_for4(cnt)), dim3(256), 0, h->stream, h->P2, base2, p.padded2,
			                   p.sh2, begin, perc_dev, cnt);
		else
			hipLaunchKernelGGL(range_unshift_kernel<false>, dim3(grid_for4(cnt)), dim3(256), 0, h->stream, h->P2, base2, p.padded2,
			                   p.sh2, begin, perc_dev, cnt);
		ZH_HIP(hipGetLastError());
	}
	return ZEN_HIP_OK;
}

// profiling hooks for bench.py: the two passes are zen_hip_hpr engines with HIP events around every launch
int zen_hip_hpri_profile(zen_hip_hpri_t h, int enable)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null handle");
	ZH_TRY(zen_hip_hpr_profile(h->eh, enable));
	return zen_hip_hpr_profile(h->ep, enable);
}

int zen_hip_hpri_profile_get_all(zen_hip_hpri_t h, int pass, double ms[6], unsigned long long launches[6])
{
	if (!h || (pass != 1 && pass != 2))
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_profile_get_all: pass must be 1 or 2");
	return zen_hip_hpr_profile_get_all(pass == 1 ? h->eh : h->ep, ms, launches);
}

int zen_hip_hpri_process(zen_hip_hpri_t h, const float* audio_host, size_t n, float* harm_host,
                         float* perc_host, float* resid_host)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process: null handle");
	if (n == 0)
		return ZEN_HIP_OK; // the reference returns three empty vectors (hps.cu:128-221 on an empty input)
	if (!audio_host)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process: null argument");
	if (h->n_clips != 1)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process(host) needs a handle created with n_clips == 1");
	if (n > h->stage_cap) {
		ZH_HIP(hipStreamSynchronize(h->stream));
		(void)hipFree(h->stage_in);
		h->stage_in = nullptr; // a failed allocation below must not leave freed pointers behind a stale capacity
		for (int i = 0; i < 3; ++i) {
			(void)hipFree(h->stage_out[i]);
			h->stage_out[i] = nullptr;
		}
		h->stage_cap = 0;
		ZH_HIP(hipMalloc((void**)&h->stage_in, sizeof(float) * n));
		for (int i = 0; i < 3; ++i)
			ZH_HIP(hipMalloc((void**)&h->stage_out[i], sizeof(float) * n));
		h->stage_cap = n;
	}
	ZH_HIP(hipMemcpyAsync(h->stage_in, audio_host, sizeof(float) * n, hipMemcpyHostToDevice, h->stream));
	ZH_TRY(zen_hip_hpri_process_device(h, h->stage_in, n, n, harm_host ? h->stage_out[0] : nullptr,
	                                   perc_host ? h->stage_out[1] : nullptr,
	                                   resid_host ? h->stage_out[2] : nullptr, n));
	float* hosts[3] = {harm_host, perc_host, resid_host};
	for (int i = 0; i < 3; ++i)
		if (hosts[i])
			ZH_HIP(hipMemcpyAsync(hosts[i], h->stage_out[i], sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
	ZH_HIP(hipStreamSynchronize(h->stream));
	return ZEN_HIP_OK;
}

} // extern "C"
