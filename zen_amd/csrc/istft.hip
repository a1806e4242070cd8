// istft.hip -- synthesis kernels of the HPSS engine for gfx950, built on fft_dev.h.
//
//   istft_kernel            : hps.cu:498-579 per consumed row: mask from (H, P), complex*real, inverse FFT
//                             (only the nwin real outputs that are used are computed), *COLA.
//   istft_hard_multi_kernel : the same for hard masks with several outputs, one workgroup per frame.
// Its own translation unit because it is compiled with the max-ILP scheduling strategy (build.py), which
// helps these kernels (-4 %) and hurts the analysis kernel of stft.hip.
#include "common.h"
#include "fft_dev.h"
#include "fft_launch.h"
#include "masks.h"
#include "stft.h"

#include <cfloat>

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

using zfft::Plan;

// ------------------------------------------------------------------------------------------------
// |S| is exactly Hermitian (fft_dev.h), so H (|S| or its time median / box mean) is mirror symmetric, and so is P
// except for the p_mid bins next to either end of the row, where the replicate border differs.
// The upper halves of the H and P rows are therefore never read: they cost no HBM traffic.
// HARD: hard masks decided by exact comparison (masks.h hard_mask_exact; both thresholds valid: the launcher checks) --
// the default of every configuration but the soft-mask / SSE ones.  Both comparisons are made for every bin and the
// output's mask is formed from them with wave-uniform coefficients: m = s0 + s1 * (gh*hm + gp*pm), all operands 0 or
// +-1, so every operation is exact and the value is mask_value()'s: (gh, gp, s0, s1) = (0, 1, 0, 1) percussive,
// (1, 0, 0, 1) harmonic, (out_h, out_p, 1, -1) residual = 1 - (hm + pm) (hps.h:35-43).  No branch, and none of the
// generic build's divide / soft / SSE variants, which it carries as wave-uniform branches around every element.
struct HardSel {
	float gh, gp, s0, s1;
};
__device__ __forceinline__ HardSel hard_sel(int which, const MaskCfg& c)
{
	if (which == 0)
		return HardSel{0.0f, 1.0f, 0.0f, 1.0f};
	if (which == 1)
		return HardSel{1.0f, 0.0f, 0.0f, 1.0f};
	return HardSel{c.out_h ? 1.0f : 0.0f, c.out_p ? 1.0f : 0.0f, 1.0f, -1.0f};
}
__device__ __forceinline__ float hard_mask_sel(float h, float p, const HardThr& t, const HardSel& k)
{
	const float pm = hard_mask_exact(p, h + FLT_EPSILON, t.p); // hps.cu:501-505
	const float hm = hard_mask_exact(h, p + FLT_EPSILON, t.h); // hps.cu:535-540
	return k.s0 + k.s1 * (k.gh * hm + k.gp * pm);
}

// MODE 0: any mask (generic); 1: hard masks by comparison, any output; 2: the same for the percussive output alone
// (pass 2 of the offline configuration, the realtime default): one comparison per bin.
template <int MODE>
struct IstftIn {
	const float2* S;
	const float* H;
	const float* P;
	MaskCfg cfg;
	HardThr thr;
	HardSel sel;
	int which;
	int n;
	int p_mid;
	__device__ __forceinline__ float2 operator()(int idx, int) const
	{
		const bool mirror = idx > (n >> 1); // upper half: S[n-k] = conj(S[k])
		const int lo = mirror ? n - idx : idx;
		float2 z = S[lo];
		if (mirror)
			z.y = -z.y;
		const int pi = (mirror && idx >= n - p_mid) ? idx : lo;
		float m;
		if constexpr (MODE == 2)
			m = hard_mask_exact(P[pi], H[lo] + FLT_EPSILON, thr.p); // hps.cu:501-505
		else if constexpr (MODE == 1)
			m = hard_mask_sel(H[lo], P[pi], thr, sel);
		else
			m = mask_value_thr(which, H[lo], P[pi], cfg, thr);
		return make_float2(z.x * m, z.y * m); // apply_mask_functor hps.h:58-66
	}
};

struct IstftOut {
	float* Y;
	float cola;
	float* ready;       // single-frame calls: the finished hop = carry + first half of this frame
	const float* cv;    // the thread's four carry samples (saved by the housekeeping block of the analysis kernel of
	int hop;            // the same call), idx = tf + slot*TF, slot < 4: loaded before the transform -- a load here
	                    // would queue behind the stores
	__device__ __forceinline__ void operator()(int idx, float2 x, bool, int slot) const
	{
		const float y = x.x * cola; // the product of overlap_add_functor hps.h:68-80; the sum is in finalize ...
		Y[idx] = y;
		if (ready && idx < hop) // ... or here, when the call is a single hop (hps.cu:341-363 hands out [0, hop))
			ready[idx] = cv[slot & 3] + y;
	}
};

// the four carry samples of thread tf (hop == 4*TF)
template <int TF>
__device__ __forceinline__ void load_carry(const float* carry, int tf, bool wanted, float (&cv)[4])
{
#pragma unroll
	for (int i = 0; i < 4; ++i)
		cv[i] = wanted ? carry[tf + i * TF] : 0.0f;
}

template <int LOG2N, int MODE>
__global__ __launch_bounds__(Plan<LOG2N>::THREADS) void istft_kernel(IstftArgs a)
{
	using PL = Plan<LOG2N>;
	extern __shared__ float2 lds[];
	const int tid = threadIdx.x, s = blockIdx.z, oi = blockIdx.y;
	const int slot = tid / PL::TF, tf = tid - slot * PL::TF;
	const int f = blockIdx.x * PL::FRAMES_PER_BLOCK + slot;
	const bool active = f < a.n_frames;
	const long long ring_row = ((a.crow0 + f) % a.ring_rows) + (long long)s * a.ring_rows;
	IstftIn<MODE> in;
	in.S = a.S + ring_row * a.s_stride;
	in.n = PL::N;
	in.H = a.h_is_ring ? a.H + ring_row * PL::N : a.H + (long long)s * a.h_stream_stride + (long long)f * PL::N;
	in.P = a.P + (long long)s * a.p_stream_stride + (long long)f * PL::N;
	in.cfg = MaskCfg{a.beta, a.beta_h, a.soft, a.power, a.sse, a.out_h, a.out_p};
	in.thr = HardThr{a.thr_p, a.thr_h, a.thr_p_inc, a.thr_h_inc};
	in.which = a.out_id[oi];
	in.sel = hard_sel(in.which, in.cfg);
	in.p_mid = a.p_mid;
	IstftOut out;
	out.Y = a.Y[oi] + (long long)s * a.y_stream_stride + (long long)f * (PL::N / 2);
	out.cola = a.cola;
	out.ready = (a.n_frames == 1 && a.ready[oi]) ? a.ready[oi] + (long long)s * a.hop : nullptr;
	float cv[4];
	load_carry<PL::TF>(a.carry[oi] + (long long)s * a.hop, tf, out.ready != nullptr, cv);
	out.cv = cv;
	out.hop = a.hop;
	zfft::fft_frame<LOG2N, true, false, true>(tf, lds + slot * PL::LDS_FLOAT2, a.tw, in, out, active);
	if (a.n_frames == 1 && a.publish_seq && a.ready[oi]) { // see rt_fused.hip publish_ready (a.n_frames == 1: one frame per block)
		__threadfence_system();
		__syncthreads();
		if (tid == 0)
			__hip_atomic_store(reinterpret_cast<unsigned*>(a.ready[oi] + (long long)s * a.hop + a.hop), a.seq, __ATOMIC_RELEASE,
			                   __HIP_MEMORY_SCOPE_SYSTEM);
	}
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
	HardThr thr;
	unsigned* bits; // bit 2*slot: percussive mask, bit 2*slot + 1: harmonic mask
	bool cmp_only;  // both hard-mask thresholds are valid (the usual case)
	int which;
	int first;
	int n;
	int p_mid;
	__device__ __forceinline__ float2 operator()(int idx, int slot) const
	{
		const bool mirror = idx > (n >> 1);
		const int lo = mirror ? n - idx : idx;
		float2 z = S[lo];
		if (mirror)
			z.y = -z.y;
		if (first) {
			const float h = H[lo], p = P[(mirror && idx >= n - p_mid) ? idx : lo]; // see IstftIn
			unsigned pm, hm;
			if (cmp_only) { // wave-uniform; both thresholds valid: no divide behind it
				pm = cfg.out_p || which == 0 ? (unsigned)(hard_mask_exact(p, h + FLT_EPSILON, thr.p) != 0.0f) : 0u;
				hm = cfg.out_h || which == 1 ? (unsigned)(hard_mask_exact(h, p + FLT_EPSILON, thr.h) != 0.0f) : 0u;
			}
			else {
				pm = cfg.out_p || which == 0 ? (unsigned)(pmask_thr(h, p, cfg, thr) != 0.0f) : 0u;
				hm = cfg.out_h || which == 1 ? (unsigned)(hmask_thr(h, p, cfg, thr) != 0.0f) : 0u;
			}
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
	in.thr = HardThr{a.thr_p, a.thr_h, a.thr_p_inc, a.thr_h_inc};
	in.bits = &bits;
	in.cmp_only = a.thr_p != 0.0 && a.thr_h != 0.0;
	in.p_mid = a.p_mid;
	for (int oi = 0; oi < a.n_out; ++oi) {
		in.which = a.out_id[oi];
		in.first = oi == 0;
		IstftOut out;
		out.Y = a.Y[oi] + (long long)s * a.y_stream_stride + (long long)f * (PL::N / 2);
		out.cola = a.cola;
		out.ready = (a.n_frames == 1 && a.ready[oi]) ? a.ready[oi] + (long long)s * a.hop : nullptr;
		float cv[4];
		load_carry<PL::TF>(a.carry[oi] + (long long)s * a.hop, tf, out.ready != nullptr, cv);
		out.cv = cv;
		out.hop = a.hop;
		// The thread index and the table pointer are made opaque per output: otherwise every LDS address and
		// twiddle index of the transform (all functions of tf alone) is hoisted out of this loop and kept
		// in registers (226 VGPRs instead of 90 at nfft 4096, 128 spilled at nfft 16384).
		int tf_o = tf;
		const float2* tw_o = a.tw;
		asm volatile("" : "+v"(tf_o));
		asm volatile("" : "+s"(tw_o));
		zfft::fft_frame<LOG2N, true, false, true>(tf_o, lds + slot * PL::LDS_FLOAT2, tw_o, in, out, active);
		if (a.n_frames == 1 && a.publish_seq && a.ready[oi]) {
			__threadfence_system();
			__syncthreads();
			if (tid == 0)
				__hip_atomic_store(reinterpret_cast<unsigned*>(a.ready[oi] + (long long)s * a.hop + a.hop), a.seq,
				                   __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		__syncthreads(); // the frame image is reused by the next output
	}
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
		dim3 grid((unsigned)ceil_div((size_t)a.n_frames, (size_t)PL::FRAMES_PER_BLOCK), 1, (unsigned)a.n_streams);
		hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, a);
		ZH_HIP(hipGetLastError());
		return ZEN_HIP_OK;
	}
	dim3 grid((unsigned)ceil_div((size_t)a.n_frames, (size_t)PL::FRAMES_PER_BLOCK), (unsigned)a.n_out, (unsigned)a.n_streams);
	if (!a.soft && !a.sse && a.thr_p != 0.0 && a.thr_h != 0.0) { // hard masks by comparison: the builds with nothing else in them
		auto kern = (a.n_out == 1 && a.out_id[0] == 0) ? istft_kernel<LOG2N, 2> : istft_kernel<LOG2N, 1>;
		ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
		hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, a);
		ZH_HIP(hipGetLastError());
		return ZEN_HIP_OK;
	}
	auto kern = istft_kernel<LOG2N, 0>;
	ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
	hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}


} // namespace

int launch_istft(int log2n, const IstftArgs& a, hipStream_t stream)
{
	if (a.n_frames <= 0 || a.n_out <= 0)
		return ZEN_HIP_OK;
#define CALL(L) launch_istft_t<L>(a, stream)
	ZH_DISPATCH_LOG2N(log2n, CALL)
#undef CALL
}

} // namespace zen_hip_impl
