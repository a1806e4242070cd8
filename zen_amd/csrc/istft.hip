// istft.hip -- synthesis kernels of the HPSS engine for gfx950, built on fft_dev.h.
//
//   istft_kernel            : hps.cu:498-579 per consumed row: mask from (H, P), complex*real, inverse FFT
//                             (only the nwin real outputs that are used are computed), *COLA.
//   istft_hard_multi_kernel : the same for hard masks with several outputs, one workgroup per frame.
// Its own translation unit because it is compiled with the max-ILP scheduling strategy (build.py), which
// helps these kernels (-4 %); since the loads of a frame are in flight together it helps stft.hip's as well.
#include "common.h"
#include "bounds.h"
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

// The thread's word of mask bits (two per bin: percussive, harmonic) as two-bit signed codes of the output's mask value:
// 00 -> 0, 01 -> 1, 11 -> -1 (the residual mask 1 - (hm + pm) where both masks are set, hps.h:35-43).  Wave-uniform, once
// per thread; per bin the synthesis then needs one signed bit-field extract and one conversion.
__device__ __forceinline__ unsigned mask_code(unsigned w, int which, const MaskCfg& c)
{
	const unsigned E = 0x55555555u;
	const unsigned pm = w & E, hm = (w >> 1) & E;
	if (which == 0)
		return pm;
	if (which == 1)
		return hm;
	const unsigned a = c.out_p ? pm : 0u, b = c.out_h ? hm : 0u; // hps.cu:562-567: a mask that is not computed counts as 0
	return (~(a ^ b) & E) | ((a & b) << 1);                      // 1 - (a + b): 0 + 0 -> 01, one set -> 00, both -> 11
}

// MODE 0: any mask (generic); 1: hard masks by comparison, any output; 2: the same for the percussive output alone
// (the realtime default): one comparison per bin; 3: hard masks from the bits of launch_mask_bits (blocks of frames);
// 5: soft masks (harmonic / percussive output); 7: soft masks read from the rows the median kernel left (IstftArgs::mask_rows).
// V: complex values per thread of the transform (fft_dev.h Plan::V).  A word of IstftArgs::bits_t holds the masks of bins
// w + s*nfft/16, s < 16; a thread of a V = 32 plan owns bins tf + s'*nfft/32, s' < 32: its words are tf and tf + nfft/32,
// slot s' sits in word s' & 1 at position s' >> 1.
template <int MODE, int V = 16>
struct IstftIn {
	const float2* S;
	const float* H;
	const float* P;
	unsigned bw[V / 16]; // MODE 3: the thread's word(s) of mask bits (IstftArgs::bits_t)
	MaskCfg cfg;
	HardThr thr;
	HardSel sel;
	int which;
	int n;
	int p_mid;
	// Split input (fft_dev.h has_split_input): the first pass issues load() for all sixteen elements, then finish().
	// Written as one operator() the loads of an element sat behind the mask arithmetic of the one before -- wave-uniform
	// branches (soft / SSE / which output), a division -- and every element cost its own trip to memory.
	struct Raw {
		float2 z;
		float h, p;
	};
	// MODE 3: the thread's word of mask bits was loaded in the kernel's prologue; its first use is here, behind the loads
	// of the spectrum -- turned into the output's codes in the prologue it cost a trip to memory of its own in front of them
	__device__ __forceinline__ void prepare()
	{
		if constexpr (MODE == 3) {
#pragma unroll
			for (int i = 0; i < V / 16; ++i)
				bw[i] = mask_code(bw[i], which, cfg);
		}
	}
	__device__ __forceinline__ Raw load(int idx, int) const
	{
		const bool mirror = idx > (n >> 1); // upper half: S[n-k] = conj(S[k])
		const int lo = mirror ? n - idx : idx;
		Raw r;
		ZH_CHK(S + lo, 1);
		r.z = S[lo];
		r.h = r.p = 0.0f;
		if constexpr (MODE == 7) { // the output's soft mask, computed by the median kernel: one value per bin (H: that row)
			ZH_CHK(H + ((mirror && idx >= n - p_mid) ? idx : lo), 1);
			r.p = H[(mirror && idx >= n - p_mid) ? idx : lo];
		}
		else if constexpr (MODE != 3) { // (MODE 3: the masks are in the thread's word of bits)
			ZH_CHK(H + lo, 1);
			ZH_CHK(P + ((mirror && idx >= n - p_mid) ? idx : lo), 1);
			r.h = H[lo];
			r.p = P[(mirror && idx >= n - p_mid) ? idx : lo];
		}
		return r;
	}
	__device__ __forceinline__ float2 finish(const Raw& r, int idx, int slot) const
	{
		float2 z = r.z;
		if (idx > (n >> 1))
			z.y = -z.y;
		float m;
		if constexpr (MODE == 3) { // the comparisons were made once per bin (launch_mask_bits + _transpose)
			const unsigned w = V == 32 ? bw[slot & (V / 16 - 1)] : bw[0];
			const int pos = V == 32 ? slot >> 1 : slot;
			m = (float)((int)(w << (30 - 2 * pos)) >> 30); // two bits, sign-extended: 00 -> 0, 01 -> 1, 11 -> -1 (mask_code)
		}
		else if constexpr (MODE == 7)
			m = r.p;
		else if constexpr (MODE == 5) { // soft masks alone (soft_mask_functor hps.h:116-129; the residual does not exist with them)
			const float a = which == 0 ? r.p : r.h, b = which == 0 ? r.h : r.p;
			const float xp = powi(a, cfg.power), yp = powi(b, cfg.power);
			m = xp / (xp + yp + FLT_EPSILON);
		}
		else if constexpr (MODE == 2)
			m = hard_mask_exact(r.p, r.h + FLT_EPSILON, thr.p); // hps.cu:501-505
		else if constexpr (MODE == 1)
			m = hard_mask_sel(r.h, r.p, thr, sel);
		else
			m = mask_value_thr(which, r.h, r.p, cfg, thr);
		return make_float2(z.x * m, z.y * m); // apply_mask_functor hps.h:58-66
	}
	__device__ __forceinline__ float2 operator()(int idx, int slot) const { return finish(load(idx, slot), idx, slot); } // (not with MODE 3)
};

template <int V = 16>
struct IstftOutV {
	float* Y;
	float cola;
	float* ready;       // single-frame calls: the finished hop = carry + first half of this frame
	const float* cv;    // the thread's V/4 carry samples (saved by the housekeeping block of the analysis kernel of
	int hop;            // the same call), idx = tf + slot*TF, slot < V/4: loaded before the transform -- a load here
	                    // would queue behind the stores
	__device__ __forceinline__ void operator()(int idx, float2 x, bool, int slot) const
	{
		const float y = x.x * cola; // the product of overlap_add_functor hps.h:68-80; the sum is in finalize ...
		ZH_CHK(Y + idx, 1);
		Y[idx] = y;
		if (ready && idx < hop) { // ... or here, when the call is a single hop (hps.cu:341-363 hands out [0, hop))
			ZH_CHK(ready + idx, 1);
			ready[idx] = cv[slot & (V / 4 - 1)] + y;
		}
	}
};
using IstftOut = IstftOutV<16>;

// the V/4 carry samples of thread tf (hop == (V/4)*TF)
template <int TF, int CVN>
__device__ __forceinline__ void load_carry(const float* carry, int tf, bool wanted, float (&cv)[CVN])
{
#pragma unroll
	for (int i = 0; i < CVN; ++i) {
		if (wanted)
			ZH_CHK(carry + tf + i * TF, 1);
		cv[i] = wanted ? carry[tf + i * TF] : 0.0f;
	}
}

// nfft 512 .. 2048: the twiddle table lives in LDS (4 / 8 KB behind the 34.8 KB of frame images: still four workgroups per CU)
constexpr bool istft_tw_in_lds(int log2n) { return log2n >= 9 && log2n <= 10; }
template <int LOG2N>
constexpr size_t istft_lds_bytes() { return lds_bytes<LOG2N>() + (istft_tw_in_lds(LOG2N) ? sizeof(float2) * (size_t)(Plan<LOG2N>::N / 2) : 0); }

// (four waves per SIMD: with every load of the first pass in flight at once the max-ILP schedule otherwise takes 130 to
// 200 registers -- three or two waves, and at nfft 8192 one workgroup per CU instead of two)
// Exactly four: the LDS image (34.8 KB per 256 threads at every size) allows no more, and a scheduler that believes in
// six or eight waves keeps the registers low by waiting for every load right behind it (the soft-mask build at nfft 1024:
// 76 registers, one trip to memory per element).
// (a 512-thread frame of 32 values per thread, nfft 16384 with ZEN_FFT16K_V = 32: two waves per SIMD, 256 registers)
template <int LOG2N>
constexpr int istft_waves() { return Plan<LOG2N>::V == 32 ? 2 : 4; }

template <int LOG2N, int MODE>
__global__ __launch_bounds__(Plan<LOG2N>::THREADS, istft_waves<LOG2N>())
    __attribute__((amdgpu_waves_per_eu(istft_waves<LOG2N>(), istft_waves<LOG2N>()))) void istft_kernel(IstftArgs a)
{
	using PL = Plan<LOG2N>;
	constexpr bool TW_IN_LDS = istft_tw_in_lds(LOG2N);
	extern __shared__ float2 lds[];
	const int tid = threadIdx.x, s = blockIdx.z;
	const int slot = tid / PL::TF, tf = tid - slot * PL::TF;
	// Which frames and which output.  One output: blockIdx.x counts groups of frames.  Several outputs (grid_map != 0): the
	// workgroups that synthesise the outputs of one group of frames read the same spectrum rows, so they are given ids 8
	// apart -- the same XCD (workgroups go to the XCDs round robin), one after the other -- and all but the first find the
	// rows in that XCD's L2: id = 8*(g*n_out + oi) + x handles frame group 8*g + x, output oi.  (As a grid of frames x
	// outputs the three workgroups of a frame were a whole clip apart: pass 1 of the offline batch fetched every
	// spectrum row three times, 4.1 GB per step where 1.4 are compulsory.)
	int fb = blockIdx.x, oi = blockIdx.y;
	if (a.grid_map) {
		const int b = blockIdx.x, j = b >> 3;
		oi = j % a.n_out;
		fb = ((j / a.n_out) << 3) + (b & 7);
		if (fb * PL::FRAMES_PER_BLOCK >= a.n_frames)
			return; // (the grid is rounded up to whole groups of eight; nobody waits for this workgroup)
	}
	const int f_ = fb * PL::FRAMES_PER_BLOCK + slot;
	const bool active = f_ < a.n_frames;
	const int f = active ? f_ : a.n_frames - 1; // (an inactive slot transforms the last frame again and stores nothing)
	const long long ring_row = ((a.crow0 + f) % a.ring_rows) + (long long)s * a.ring_rows;
	IstftIn<MODE, PL::V> in;
	in.S = a.S + ring_row * a.s_stride;
	in.n = PL::N;
	in.H = a.h_is_ring ? a.H + ring_row * PL::N : a.H + (long long)s * a.h_stream_stride + (long long)f * PL::N;
	in.P = a.P + (long long)s * a.p_stream_stride + (long long)f * PL::N;
	in.cfg = MaskCfg{a.beta, a.beta_h, a.soft, a.power, a.sse, a.out_h, a.out_p};
	in.thr = HardThr{a.thr_p, a.thr_h, a.thr_p_inc, a.thr_h_inc};
	in.which = a.out_id[oi];
	in.sel = hard_sel(in.which, in.cfg);
	in.p_mid = a.p_mid;
#pragma unroll
	for (int i = 0; i < PL::V / 16; ++i) {
		in.bw[i] = 0;
		if constexpr (MODE == 3) { // a row is nfft/16 words (-> codes: IstftIn::prepare)
			ZH_CHK(a.bits_t + ((long long)s * a.bits_t_stream_stride + (long long)f * (PL::N / 16) + tf + i * PL::TF), 1);
			in.bw[i] = a.bits_t[(long long)s * a.bits_t_stream_stride + (long long)f * (PL::N / 16) + tf + i * PL::TF];
		}
	}
	if constexpr (MODE == 7) // the row of this output's soft mask (percussive: where P would be; harmonic: Hm), laid out as P
		in.H = (in.which == 0 ? a.P : a.Hm) + (long long)s * a.p_stream_stride + (long long)f * PL::N;
	IstftOutV<PL::V> out;
	out.Y = a.Y[oi] + (long long)s * a.y_stream_stride + (long long)f * (PL::N / 2);
	out.cola = a.cola;
	out.ready = (a.n_frames == 1 && a.ready[oi]) ? a.ready[oi] + (long long)s * a.hop : nullptr;
	float cv[PL::V / 4];
	load_carry<PL::TF>(a.carry[oi] + (long long)s * a.hop, tf, out.ready != nullptr, cv);
	out.cv = cv;
	out.hop = a.hop;
	if constexpr (TW_IN_LDS) { // the twiddle table (nfft/2 entries) behind the frame images, staged by the workgroup
		float2* twl = lds + PL::FRAMES_PER_BLOCK * PL::LDS_FLOAT2;
		for (int i = tid; i < PL::N / 2; i += PL::THREADS)
			twl[i] = a.tw[i];
		__syncthreads();
		const zfft::TwLds tl{twl};
		zfft::fft_frame<LOG2N, true, false, true>(tf, lds + slot * PL::LDS_FLOAT2, tl, in, out, active);
	}
	else {
		zfft::fft_frame<LOG2N, true, false, true>(tf, lds + slot * PL::LDS_FLOAT2, a.tw, in, out, active);
	}
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
		ZH_CHK(S + lo, 1);
		float2 z = S[lo];
		if (mirror)
			z.y = -z.y;
		if (first) {
			ZH_CHK(H + lo, 1);
			ZH_CHK(P + ((mirror && idx >= n - p_mid) ? idx : lo), 1);
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
	const int f_ = blockIdx.x * PL::FRAMES_PER_BLOCK + slot;
	const bool active = f_ < a.n_frames;
	const int f = active ? f_ : a.n_frames - 1; // (an inactive slot transforms the last frame again and stores nothing)
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
		int tw_off = 0; // (an opaque offset, not an opaque pointer: istft_run_wide_kernel)
		asm volatile("" : "+v"(tf_o));
		asm volatile("" : "+s"(tw_off));
		const float2* tw_o = a.tw + tw_off;
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

// ------------------------------------------------------------------------------------------------
// Both hard masks of every bin the synthesis reads, decided once (hps.cu:501-505, :535-540 through masks.h
// hard_mask_exact) instead of once per output and mirror image inside the synthesis kernels, which then load two
// bits per bin instead of H and P.  One thread per four entries; the four lanes of a word combine their bytes.
__global__ __launch_bounds__(256) void mask_bits_kernel(IstftArgs a, unsigned* bits, int n, int quads_per_row)
{
	const long long gid = (long long)blockIdx.x * 256 + threadIdx.x; // (a whole number of wavefronts per launch)
	const long long row = gid / quads_per_row;
	const int q = (int)(gid - row * quads_per_row);
	const bool live = row < (long long)a.n_frames * a.n_streams;
	unsigned byte = 0;
	int s = 0, f = 0;
	if (live) {
		s = (int)(row / a.n_frames);
		f = (int)(row - (long long)s * a.n_frames);
		const long long ring_row = ((a.crow0 + f) % a.ring_rows) + (long long)s * a.ring_rows;
		const float* H = a.h_is_ring ? a.H + ring_row * n : a.H + (long long)s * a.h_stream_stride + (long long)f * n;
		const float* P = a.P + (long long)s * a.p_stream_stride + (long long)f * n;
		const int e0 = 4 * q, half = n >> 1;
		float h[4], p[4];
		if (e0 + 3 <= half) {
			ZH_CHK(H + e0, 4);
			ZH_CHK(P + e0, 4);
			const float4 hv = *reinterpret_cast<const float4*>(H + e0), pv = *reinterpret_cast<const float4*>(P + e0);
			h[0] = hv.x, h[1] = hv.y, h[2] = hv.z, h[3] = hv.w;
			p[0] = pv.x, p[1] = pv.y, p[2] = pv.z, p[3] = pv.w;
		}
		else {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int e = e0 + i;
				const int idx = e <= half ? e : n - a.p_mid + (e - half - 1); // the bin; past the last entry: >= n
				const bool ok = idx < n;
				if (ok) {
					ZH_CHK(H + (idx <= half ? idx : n - idx), 1);
					ZH_CHK(P + idx, 1);
				}
				h[i] = ok ? H[idx <= half ? idx : n - idx] : 0.0f;
				p[i] = ok ? P[idx] : 0.0f; // (past the last entry: 0 against 0, no bit set; nobody reads those)
			}
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const unsigned pm = a.need_pm && hard_mask_exact(p[i], h[i] + FLT_EPSILON, a.thr_p) != 0.0f ? 1u : 0u;
			const unsigned hm = a.need_hm && hard_mask_exact(h[i], p[i] + FLT_EPSILON, a.thr_h) != 0.0f ? 1u : 0u;
			byte |= (pm | (hm << 1)) << (2 * i);
		}
	}
	unsigned w = byte << (8 * (threadIdx.x & 3));
	w |= __shfl_xor(w, 1);
	w |= __shfl_xor(w, 2);
	if (live && (threadIdx.x & 3) == 0) {
		ZH_CHK(bits + ((long long)s * a.bits_stream_stride + (long long)f * a.bits_row_words + (q >> 2)), 1);
		bits[(long long)s * a.bits_stream_stride + (long long)f * a.bits_row_words + (q >> 2)] = w;
	}
}

// natural order -> the synthesis threads' order (IstftArgs::bits_t); the mirrored half and the tail are expanded here.
// One thread per sixteen consecutive output words tf0 .. tf0+15 (tf0 a multiple of 16): slot s < 8 of all of them comes
// from ONE natural word (entries tf0 + s*nfft/16 ..), a mirrored slot from two (entries E0 - t, E0 = (16-s)*nfft/16 - tf0
// a multiple of 16: t = 0 is the first entry of word E0/16, t = 1..15 the last fifteen of the word before it); the last
// p_mid bins of the row (slot 15 of the last words) have entries of their own.  24 loads for 16 words.
__global__ __launch_bounds__(256) void mask_bits_transpose_kernel(IstftArgs a, unsigned* bits_t, int n, int log2tf)
{
	const int log2g = log2tf - 4; // groups of sixteen words per row
	const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
	const long long row = gid >> log2g;
	if (row >= (long long)a.n_frames * a.n_streams)
		return;
	const int g = (int)(gid & ((1 << log2g) - 1)), tf0 = g << 4;
	const int s = (int)(row / a.n_frames), f = (int)(row - (long long)s * a.n_frames);
	const unsigned* nat = a.bits + (long long)s * a.bits_stream_stride + (long long)f * a.bits_row_words;
	const int half = n >> 1, tail0 = n - a.p_mid, wstep = 1 << log2g; // natural words per slot
	unsigned lo[8], m0[8], m1[8];
#pragma unroll
	for (int sl = 0; sl < 8; ++sl) {
		const int w0 = (8 - sl) * wstep - g; // word of entry E0 = (16 - (8 + sl)) * TF - tf0
		ZH_CHK(nat + g + sl * wstep, 1);
		ZH_CHK(nat + w0 - 1, 2);
		lo[sl] = nat[g + sl * wstep];
		m0[sl] = nat[w0];
		m1[sl] = nat[w0 - 1];
	}
	unsigned out[16];
#pragma unroll
	for (int t = 0; t < 16; ++t) {
		unsigned r = 0;
#pragma unroll
		for (int sl = 0; sl < 8; ++sl) {
			r |= ((lo[sl] >> (2 * t)) & 3u) << (2 * sl);
			const unsigned mv = t == 0 ? m0[sl] : (m1[sl] >> (2 * (16 - t)));
			r |= (mv & 3u) << (2 * (8 + sl));
		}
		out[t] = r;
	}
	if (tf0 + 15 + (15 << log2tf) >= tail0) { // the row's last p_mid bins: their own entries
#pragma unroll
		for (int t = 0; t < 16; ++t) {
			const int idx = tf0 + t + (15 << log2tf);
			if (idx >= tail0) {
				const int e = half + 1 + idx - tail0;
				ZH_CHK(nat + (e >> 4), 1);
				out[t] = (out[t] & 0x3fffffffu) | (((nat[e >> 4] >> (2 * (e & 15))) & 3u) << 30);
			}
		}
	}
	uint4* dst = reinterpret_cast<uint4*>(bits_t + (long long)s * a.bits_t_stream_stride + ((long long)f << log2tf) + tf0);
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		ZH_CHK(dst + q, 1);
		dst[q] = make_uint4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
	}
}

// ------------------------------------------------------------------------------------------------
// Synthesis in runs (IstftRunArgs, stft.h): the small-hop pass of the offline driver (nfft <= 1024: a frame is one wavefront
// or less), hard masks as bits, one output.  A frame here costs more to move than to transform: its Y row is 2 KB written by
// the synthesis and 2 KB read back by the overlap-add launch for 1 KB of finished samples.  A wavefront therefore walks a
// RUN of consecutive frames of one stream: the second half of a frame and the first half of its successor belong to the
// same thread (idx = tf + slot*TF), so the overlap-add (hps.cu:435-449, :526-528) is four register adds per thread and the
// finished hop goes straight to where the offline driver wants it (finalize_spec_kernel's destination arithmetic): no Y
// rows, no overlap-add launch.  A run starts from the frame before it, synthesised once more for its second half alone
// (1/run of redundant work; a call's first run takes the previous call's carry instead).  What made the first attempt at
// this no faster than the two launches: gfx9 counts loads and stores in ONE in-order counter,
// so the stores of frame f sat in front of the loads of frame f+1 that the wavefront then waited for.  Here the next
// frame's spectrum row and its word of mask bits are fetched into registers BEFORE the current frame is transformed (32
// registers, which the nfft-1024 build has and the nfft-16384 one of pass 1 has not).
// (Transforming the frame again from its samples instead of loading the 4 KB row -- so that the analysis launch need not
// store the spectrum at all -- was measured too: 0.78 ms against 0.49 for this kernel, for 0.10 saved in the analysis.)
struct IstftRunIn {
	const float2* S; // the frame's spectrum in registers: bin tf + slot*TF (upper half already conjugated)
	unsigned w;      // mask_code of the thread's word of IstftArgs::bits_t
	__device__ __forceinline__ float2 operator()(int, int slot) const
	{
		const float m = (float)((int)(w << (30 - 2 * slot)) >> 30); // IstftIn<3>::finish
		const float2 z = S[slot];
		return make_float2(z.x * m, z.y * m); // apply_mask_functor hps.h:58-66
	}
};
struct IstftRunOut { // idx = tf + slot*TF, slot < 8 (HALF_OUT): the frame's nwin outputs
	float* y;
	float cola;
	__device__ __forceinline__ void operator()(int, float2 x, bool, int slot) const { y[slot & 7] = x.x * cola; } // hps.h:68-80
};

template <int LOG2N>
__global__ __launch_bounds__(Plan<LOG2N>::THREADS, 4) __attribute__((amdgpu_waves_per_eu(4, 4))) void istft_run_kernel(IstftRunArgs a)
{
	using PL = Plan<LOG2N>;
	constexpr int TF = PL::TF, N = PL::N, HOP = N / 4;
	static_assert(TF <= 64 && PL::V == 16, "a frame lives inside one wavefront: its passes synchronise there (frame_sync)");
	extern __shared__ float2 lds[];
	const int tid = threadIdx.x, s = blockIdx.y;
	const int slot = tid / TF, tf = tid - slot * TF;
	float2* twl = lds + PL::FRAMES_PER_BLOCK * PL::LDS_FLOAT2; // the twiddle table behind the frame images (istft_kernel)
	for (int i = tid; i < N / 2; i += PL::THREADS)
		twl[i] = a.tw[i];
	__syncthreads();
	const zfft::TwLds tl{twl};
	float2* img = lds + slot * PL::LDS_FLOAT2;
	const int i0 = (blockIdx.x * PL::FRAMES_PER_BLOCK + slot) * a.run; // the run's first frame (>= n_frames: nothing is stored)
	const float2* S_s = a.S + (long long)s * a.ring_rows * a.s_stride;
	const unsigned* bits_s = a.bits_t + (long long)s * a.bits_t_stream_stride;
	const IstftRunGroup& G = a.g[0]; // (one group of one output: launch_istft_run checks)
	float* out_s = G.out + (long long)s * G.out_stride;
	const MaskCfg cfg{0.0f, 0.0f, 0, 0, 0, a.out_h, a.out_p};
	// frame i of the call (one outside it: some frame of it, the result is not used): its sixteen bins tf + slot*TF -- the lower
	// half as stored, the upper half from the mirror image (conjugated where it is used) -- and its word of mask bits
	auto load_row = [&](int i, float2 (&z)[16]) {
		const int r = i < 0 ? 0 : (i < a.n_frames ? i : a.n_frames - 1);
		const float2* row = S_s + ((a.crow0 + r) % a.ring_rows) * a.s_stride;
#pragma unroll
		for (int sl = 0; sl < 16; ++sl) {
			const int idx = tf + sl * TF;
			ZH_CHK(row + (idx > N / 2 ? N - idx : idx), 1);
			z[sl] = row[idx > N / 2 ? N - idx : idx];
		}
	};
	auto load_bits = [&](int i) {
		const int r = i < 0 ? 0 : (i < a.n_frames ? i : a.n_frames - 1);
		ZH_CHK(bits_s + ((long long)r * (N / 16) + tf), 1);
		return bits_s[(long long)r * (N / 16) + tf];
	};
	float cprev[4], carry[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
	for (int j = 0; j < 4; ++j) { // (used by the call's first run only)
		ZH_CHK(G.carry_prev[0] + ((long long)s * HOP + tf + j * TF), 1);
		cprev[j] = G.carry_prev[0][(long long)s * HOP + tf + j * TF];
	}
	float2 S[16];
	load_row(i0 - 1, S);
	unsigned bw = load_bits(i0 - 1);
	for (int i = i0 - 1; i < i0 + a.run; ++i) {
		float2 Sn[16];
		load_row(i + 1, Sn); // the next frame's row and masks: in flight under this frame's transform
		const unsigned bn = load_bits(i + 1);
#pragma unroll
		for (int sl = 0; sl < 16; ++sl)
			if (tf + sl * TF > N / 2) // S[n-k] = conj(S[k]) (IstftIn)
				S[sl].y = -S[sl].y;
		float y[8];
		{
			IstftRunIn in{S, mask_code(bw, G.which[0], cfg)};
			IstftRunOut out{y, a.cola};
			// (the thread index is made opaque per transform: otherwise every LDS address and table index of the transform, all
			// functions of tf alone, is hoisted out of the loop and kept in registers -- see istft_hard_multi_kernel)
			int tf_o = tf;
			asm volatile("" : "+v"(tf_o));
			zfft::fft_frame<LOG2N, true, false, true>(tf_o, img, tl, in, out, true);
		}
		if (i >= i0 && i < a.n_frames) { // the finished hop of frame i (finalize_spec_kernel)
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const float v = carry[j] + y[j];
				const long long p0 = a.pos0 + (long long)i * HOP + tf + j * TF;
				const long long d = p0 - G.shift;
				if (d >= 0 && d < G.len) {
					ZH_CHK(out_s + d, 1);
					out_s[d] = v;
				}
				if (p0 >= G.dup_from) {
					const long long d2 = p0 - G.dup_shift;
					if (d2 >= 0 && d2 < G.dup_len) {
						ZH_CHK(out_s + d2, 1);
						out_s[d2] = v;
					}
				}
			}
		}
#pragma unroll
		for (int j = 0; j < 4; ++j)
			carry[j] = (i < 0) ? cprev[j] : y[4 + j]; // (i < 0: the call's first run, the frame before the call)
		if (i == a.n_frames - 1) {
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				ZH_CHK(G.carry_next[0] + ((long long)s * HOP + tf + j * TF), 1);
				G.carry_next[0][(long long)s * HOP + tf + j * TF] = carry[j];
			}
		}
#pragma unroll
		for (int sl = 0; sl < 16; ++sl)
			S[sl] = Sn[sl];
		bw = bn;
	}
}

// The same for the transforms a workgroup shares (nfft 2048 .. 16384: the large-hop pass of the offline driver, three
// outputs in two groups -- H, and P + R summed for pass 2).  One workgroup per (run, group, stream).  No registers are left
// for a prefetched row here (istft_kernel<14, 3> sits at 126 of the 128 that four waves per SIMD allow), so the other half
// of the remedy is used: the finished hop of frame i waits in four registers and is stored inside the NEXT frame's
// transform, right behind the loads of its spectrum row (the input functor's prepare(), fft_dev.h) -- the in-order memory
// counter then lets the wavefront wait for those loads without draining the stores first.  A group of two outputs loads
// the row again for its second transform (an L2 hit, with nothing but those old stores in front of it).
struct IstftRunPend { // the finished hop that waits for the next frame's loads, and where it goes
	float v[4];
	float* base; // sample tf + j*tf_step of the hop goes to base[tf + j*tf_step]: its place in the destination, or the sink
	int tf, tf_step;
	// Four stores on EVERY path: behind the loads of the next frame's row the compiler can then count them and wait for the
	// loads alone; with a conditional store there it must assume none and waits for everything (s_waitcnt vmcnt(0): the
	// stores drained after all).  Hence the sink: a turn that has no hop to deliver (the frame before the run), or delivered
	// it by the slow path below, stores its four values where nobody reads them.
	__device__ __forceinline__ void flush() const
	{
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			ZH_CHK(base + tf + j * tf_step, 1);
			base[tf + j * tf_step] = v[j];
		}
	}
};
struct IstftRunWideIn : IstftIn<3, 16> {
	IstftRunPend* pend; // null: nothing to store behind this transform's loads
	__device__ __forceinline__ void prepare()
	{
		if (pend) // (compile-time after inlining: the first transform of a turn)  Before the mask codes: those wait for the word of bits
			pend->flush();
		IstftIn<3, 16>::prepare();
	}
};

template <int LOG2N, int NG>
__global__ __launch_bounds__(Plan<LOG2N>::THREADS, istft_waves<LOG2N>())
    __attribute__((amdgpu_waves_per_eu(istft_waves<LOG2N>(), istft_waves<LOG2N>()))) void istft_run_wide_kernel(IstftRunArgs a)
{
	using PL = Plan<LOG2N>;
	constexpr int TF = PL::TF, N = PL::N, HOP = N / 4;
	static_assert(PL::V == 16, "one word of mask bits per thread");
	extern __shared__ float2 lds[];
	// (grid: runs x streams x groups -- the groups of two outputs, twice the work per workgroup, come first in IstftRunArgs::g
	// and are therefore dispatched first: the shorter workgroups fill the launch's tail)
	const int tid = threadIdx.x, s = blockIdx.y;
	const int slot = tid / TF, tf = tid - slot * TF;
	float2* img = lds + slot * PL::LDS_FLOAT2;
	const IstftRunGroup& G = a.g[blockIdx.z];
	const int i0 = (blockIdx.x * PL::FRAMES_PER_BLOCK + slot) * a.run; // the run's first frame (>= n_frames: nothing is stored)
	const float2* S_s = a.S + (long long)s * a.ring_rows * a.s_stride;
	const unsigned* bits_s = a.bits_t + (long long)s * a.bits_t_stream_stride;
	const MaskCfg cfg{0.0f, 0.0f, 0, 0, 0, a.out_h, a.out_p};
	float carry[NG][4];
#pragma unroll
	for (int k = 0; k < NG; ++k)
#pragma unroll
		for (int j = 0; j < 4; ++j)
			carry[k][j] = 0.0f;
	float* out_s = G.out + (long long)s * G.out_stride;
	float* sink_s = a.sink + (long long)s * HOP;
	IstftRunPend pend;
	pend.base = sink_s;
	pend.tf = tf;
	pend.tf_step = TF;
#pragma unroll
	for (int j = 0; j < 4; ++j)
		pend.v[j] = 0.0f;
	// every slot of every workgroup walks run + 1 frames (the transforms of nfft > 1024 meet at workgroup barriers): i = i0 - 1 is
	// the frame before the run, synthesised for its second half alone -- or, for the call's first run, nothing: the previous
	// call's carry takes its place
	for (int i = i0 - 1; i < i0 + a.run; ++i) {
		const int r = i < 0 ? 0 : (i < a.n_frames ? i : a.n_frames - 1);
		const float2* row = S_s + ((a.crow0 + r) % a.ring_rows) * a.s_stride;
		ZH_CHK(bits_s + ((long long)r * (N / 16) + tf), 1);
		const unsigned bw = bits_s[(long long)r * (N / 16) + tf];
		float v[4];
#pragma unroll
		for (int k = 0; k < NG; ++k) {
			if (k >= G.n_out) // (workgroup-uniform: a group of one output in a call that also has a group of two)
				break;
			IstftRunWideIn in;
			in.S = row;
			in.H = in.P = nullptr;
			in.bw[0] = bw;
			in.cfg = cfg;
			in.thr = HardThr{0.0, 0.0, 0, 0};
			in.sel = HardSel{0.0f, 0.0f, 0.0f, 0.0f};
			in.which = G.which[k];
			in.n = N;
			in.p_mid = 0;
			in.pend = k == 0 ? &pend : nullptr;
			float y[8];
			IstftRunOut out{y, a.cola};
			int tf_o = tf; // (opaque per transform: see istft_hard_multi_kernel)
			int tw_off = 0; // (the table pointer through an opaque OFFSET: made opaque itself it loses its address space, and every
			asm volatile("" : "+v"(tf_o)); // twiddle becomes a flat load waited for on the spot, stage by stage)
			asm volatile("" : "+s"(tw_off));
			const float2* tw_o = a.tw + tw_off;
			const zfft::TwGlobalPre twp{tw_o};
			zfft::fft_frame<LOG2N, true, false, true>(tf_o, img, twp, in, out, true);
			if (i < 0) { // (wave-uniform) the call's first run
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					ZH_CHK(G.carry_prev[k] + ((long long)s * HOP + tf + j * TF), 1);
					y[4 + j] = G.carry_prev[k][(long long)s * HOP + tf + j * TF];
				}
			}
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const float f = carry[k][j] + y[j];
				v[j] = k == 0 ? f : v[j] + f; // sum_vectors_functor hps.h:142-150 on the two finished hops
				carry[k][j] = y[4 + j];
			}
		}
		// (the hop before it went out inside this turn's first transform)  Where this one goes (finalize_spec_kernel's
		// destination arithmetic): a hop that lies inside the destination as a whole -- all of them but a few at either end of
		// a clip -- waits for the next turn; one that is cut by the destination's ends, or has a second copy to leave
		// (FinalizeArgs::dup_*: the last hops of a clip), is stored here and now, sample by sample.
		pend.base = sink_s;
#pragma unroll
		for (int j = 0; j < 4; ++j)
			pend.v[j] = v[j];
		if (i >= i0 && i < a.n_frames) {
			const long long h0 = a.pos0 + (long long)i * HOP; // position of the hop's first sample (uniform)
			const long long d0 = h0 - G.shift;
			if (d0 >= 0 && d0 + HOP <= G.len && h0 + HOP <= G.dup_from) {
				pend.base = out_s + d0;
			}
			else {
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const long long p0 = h0 + tf + j * TF;
					const long long d = p0 - G.shift;
					if (d >= 0 && d < G.len) {
						ZH_CHK(out_s + d, 1);
						out_s[d] = v[j];
					}
					if (p0 >= G.dup_from) {
						const long long d2 = p0 - G.dup_shift;
						if (d2 >= 0 && d2 < G.dup_len) {
							ZH_CHK(out_s + d2, 1);
							out_s[d2] = v[j];
						}
					}
				}
			}
		}
		if (i == a.n_frames - 1) {
#pragma unroll
			for (int k = 0; k < NG; ++k)
#pragma unroll
				for (int j = 0; j < 4; ++j)
					if (k < G.n_out) {
						ZH_CHK(G.carry_next[k] + ((long long)s * HOP + tf + j * TF), 1);
						G.carry_next[k][(long long)s * HOP + tf + j * TF] = carry[k][j];
					}
		}
	}
	pend.flush();
}

template <int LOG2N>
int launch_istft_run_t(const IstftRunArgs& a, hipStream_t stream)
{
	using PL = Plan<LOG2N>;
	if constexpr (PL::TF <= 64) {
		auto kern = istft_run_kernel<LOG2N>;
		const size_t lds = lds_bytes<LOG2N>() + sizeof(float2) * (size_t)(PL::N / 2);
		ZH_TRY(set_lds(kern, lds));
		const size_t runs = ceil_div((size_t)a.n_frames, (size_t)a.run);
		dim3 grid((unsigned)ceil_div(runs, (size_t)PL::FRAMES_PER_BLOCK), (unsigned)a.n_streams);
		hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds, stream, a);
		ZH_HIP(hipGetLastError());
		return ZEN_HIP_OK;
	}
	else if constexpr (PL::V != 16) {
		ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "istft runs: nfft = 2^%d with %d values per thread", LOG2N, PL::V);
	}
	else {
		int ng = 1; // (a call with groups of one and of two outputs -- pass 1: {H}, {P, R} -- runs the build for two: a group of one
		for (int g = 0; g < a.n_groups; ++g) // leaves the second output's turn out, a workgroup-uniform branch)
			ng = a.g[g].n_out > ng ? a.g[g].n_out : ng;
		auto kern = ng == 2 ? istft_run_wide_kernel<LOG2N, 2> : istft_run_wide_kernel<LOG2N, 1>;
		ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
		const size_t runs = ceil_div((size_t)a.n_frames, (size_t)a.run);
		dim3 grid((unsigned)ceil_div(runs, (size_t)PL::FRAMES_PER_BLOCK), (unsigned)a.n_streams, (unsigned)a.n_groups);
		hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, a);
		ZH_HIP(hipGetLastError());
		return ZEN_HIP_OK;
	}
}

template <int LOG2N>
int launch_istft_t(const IstftArgs& a, hipStream_t stream)
{
	using PL = Plan<LOG2N>;
	// all outputs of a frame in one workgroup; not at nfft 8192/16384, where a CU holds one or two frames and
	// three short workgroups per frame schedule better than one long one (measured)
	if (LOG2N <= 12 && a.n_out > 1 && !a.soft && !a.sse && !g_opt_no_istft_multi && !a.bits_t) { // (with mask bits H and P may not exist)
		auto kern = istft_hard_multi_kernel<LOG2N>;
		ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
		dim3 grid((unsigned)ceil_div((size_t)a.n_frames, (size_t)PL::FRAMES_PER_BLOCK), 1, (unsigned)a.n_streams);
		hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, a);
		ZH_HIP(hipGetLastError());
		return ZEN_HIP_OK;
	}
	dim3 grid((unsigned)ceil_div((size_t)a.n_frames, (size_t)PL::FRAMES_PER_BLOCK), (unsigned)a.n_out, (unsigned)a.n_streams);
	IstftArgs am = a; // (the kernels below are launched with `am`)
	// the outputs of a frame on one XCD, back to back (istft_kernel).  One stream only: measured on the offline workloads
	// (pass 1, nfft 16384), a single clip gains 4 % (1.32 -> 1.26 ms per 12 922 frames x 2 outputs), a batch of 64 clips
	// LOSES 6 % (2.47 -> 2.62 ms per 20 736 x 3): there the re-reads of a spectrum row were served by the Infinity Cache
	// anyway (the kernel is bound by its workgroup's lock step, not by HBM) and the plain grid schedules better.
	if (a.n_out > 1 && a.n_frames > 1 && a.n_streams == 1 && !g_opt_no_istft_xcd_map) {
		am.grid_map = 1;
		grid = dim3((unsigned)(ceil_div((size_t)grid.x, (size_t)8) * 8 * (size_t)a.n_out), 1, (unsigned)a.n_streams);
	}
	if (!a.soft && !a.sse && a.thr_p != 0.0 && a.thr_h != 0.0) { // hard masks by comparison: the builds with nothing else in them
		auto kern = (a.n_out == 1 && a.out_id[0] == 0) ? istft_kernel<LOG2N, 2> : istft_kernel<LOG2N, 1>;
		if (a.bits_t)
			kern = istft_kernel<LOG2N, 3>;
		ZH_TRY(set_lds(kern, istft_lds_bytes<LOG2N>()));
		hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), istft_lds_bytes<LOG2N>(), stream, am);
		ZH_HIP(hipGetLastError());
		return ZEN_HIP_OK;
	}
	auto kern = istft_kernel<LOG2N, 0>;
	if (a.mask_rows) { // soft masks computed by the median kernel
		kern = istft_kernel<LOG2N, 7>;
	}
	else if (LOG2N <= 13 && a.soft && !a.sse) { // the soft-mask build: nothing but that mask in it (at nfft 16384 it spills 60 bytes
		bool hp = true;                    // where the generic build spills none: 1.91 against 1.74 ms on the offline-long bench)
		for (int i = 0; i < a.n_out; ++i)
			hp = hp && a.out_id[i] < 2;
		if (hp)
			kern = istft_kernel<LOG2N, 5>;
	}
	ZH_TRY(set_lds(kern, istft_lds_bytes<LOG2N>()));
	hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), istft_lds_bytes<LOG2N>(), stream, am);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}


} // namespace

int launch_mask_bits(int nfft, const IstftArgs& a, unsigned* bits, hipStream_t stream)
{
	if (a.n_frames <= 0)
		return ZEN_HIP_OK;
	const int quads = a.bits_row_words * 4;
	const long long threads = (long long)a.n_frames * a.n_streams * quads;
	hipLaunchKernelGGL(mask_bits_kernel, dim3((unsigned)ceil_div((size_t)threads, (size_t)256)), dim3(256), 0, stream, a, bits, nfft,
	                   quads);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

int launch_mask_bits_transpose(int nfft, const IstftArgs& a, unsigned* bits_t, hipStream_t stream)
{
	if (a.n_frames <= 0)
		return ZEN_HIP_OK;
	int log2tf = 0;
	while ((16 << log2tf) < nfft)
		++log2tf;
	if (!mask_bits_supported(nfft, a.p_mid) || (a.bits_t_stream_stride & 3) != 0) // (the engine asks the same question first)
		ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "mask bits: nfft = %d, p_mid = %d outside the transposed layout", nfft, a.p_mid);
	const long long threads = ((long long)a.n_frames * a.n_streams) << (log2tf - 4);
	hipLaunchKernelGGL(mask_bits_transpose_kernel, dim3((unsigned)ceil_div((size_t)threads, (size_t)256)), dim3(256), 0, stream, a, bits_t,
	                   nfft, log2tf);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

// nfft 256 .. 1024 (hop 64 .. 256; a frame is one wavefront or less): one group of one output, the next frame's row prefetched;
// nfft 2048 .. 16384: up to three groups of one or two outputs (16 values per thread: not the ZEN_FFT16K_V = 32 A/B build)
bool istft_run_available(int log2n, int n_groups, int max_outputs_per_group)
{
	if (log2n >= 8 && log2n <= 10)
		return n_groups == 1 && max_outputs_per_group == 1;
	if (log2n == 14 && zfft::Plan<14>::V != 16)
		return false;
	return log2n >= 11 && log2n <= 14 && n_groups >= 1 && n_groups <= 3 && max_outputs_per_group <= 2;
}

int launch_istft_run(int log2n, const IstftRunArgs& a, hipStream_t stream)
{
	if (a.n_frames <= 0 || a.n_groups <= 0)
		return ZEN_HIP_OK;
	int ng = 0;
	for (int g = 0; g < a.n_groups && g < 3; ++g)
		ng = a.g[g].n_out > ng ? a.g[g].n_out : ng;
	if (a.run < 1 || a.hop != (1 << log2n) / 4 || !istft_run_available(log2n, a.n_groups, ng))
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "istft runs: run = %d, hop = %d, %d groups of up to %d outputs at nfft 2^%d", a.run, a.hop, a.n_groups, ng, log2n);
	switch (log2n) {
	case 8: return launch_istft_run_t<8>(a, stream);
	case 9: return launch_istft_run_t<9>(a, stream);
	case 10: return launch_istft_run_t<10>(a, stream);
	case 11: return launch_istft_run_t<11>(a, stream);
	case 12: return launch_istft_run_t<12>(a, stream);
	case 13: return launch_istft_run_t<13>(a, stream);
	case 14: return launch_istft_run_t<14>(a, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "istft runs: nfft = 2^%d", log2n);
	}
}

int launch_istft(int log2n, const IstftArgs& a, hipStream_t stream)
{
	if (a.n_frames <= 0 || a.n_out <= 0)
		return ZEN_HIP_OK;
#define CALL(L) launch_istft_t<L>(a, stream)
	ZH_DISPATCH_LOG2N(log2n, CALL)
#undef CALL
}

} // namespace zen_hip_impl
