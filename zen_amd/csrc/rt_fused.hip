// rt_fused.hip -- the causal realtime path (HPRRealtime<GPU>::process_next_hop, libzen/hps.cu:334-339 ->
// HPR::process_next_hop :429-486 -> apply_median_filter :488-580) in ONE launch, one workgroup per hop.
// The reference issues ~21 launches per hop; the general engine of hpr.hip needs 3 per block of hops (STFT,
// median, iSTFT) with the spectrum, |S| and P going through HBM in between.  In the causal configuration
// the time-direction median is the identity, so a hop depends on nothing but its own frame and
// the whole chain fits in one workgroup:
//
//   window + zero-padded forward FFT         -> spectrum stays in REGISTERS (16 bins per thread, the very
//                                               bins pass 0 of the inverse transform reads: fft_dev.h slot)
//   |S| (double-precision hypot)             -> registers + an LDS row image (+ the rings, for block calls)
//   frequency median of the row (W taps)     -> sorting-network medians from the LDS image (median_net.h)
//   causal time median                       -> identity: H = |S|
//   per output: mask, S*mask, inverse FFT    -> Y row (the two-term overlap-add is done by copy_output)
//
// Used for single-hop calls (latency: 1 launch instead of 3) and for blocks of hops (throughput: the
// spectrum never leaves the chip; only 4*hop input and 8*hop of Y per output touch HBM).  Same arithmetic,
// same carry protocol as the three-kernel path: the two are interchangeable call by call
// (tests/test_gpu_parity.py::test_hpr_blocking_is_invisible, ::test_block_fused_matches_three_kernel_path).
#include "common.h"
#include "fft_dev.h"
#include "masks.h"
#include "median47_core.h"
#include "median_net.h"
#include "rt_fused.h"

#include <type_traits>

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

using zfft::Plan;

// which single hops go to rt_hop_lat.hip (launch_rt_fused, launch_rt_fused_resident): all of them, unless option "no_hop_lat"
inline bool hop_lat_takes(int log2n, const RtFusedArgs& a)
{
	return a.n_frames == 1 && log2n <= 12 && !g_opt_no_hop_lat && !a.diag;
}

// LDS row image of the median stage: T-word chunks spaced T+PAD apart (see median_net.hip RowImage)
template <int T>
struct RtImage {
	static constexpr int PAD = T >= 8 ? 4 : 0;
	static constexpr int STRIDE = T + PAD;
	static constexpr int LOG2T = T == 16 ? 4 : (T == 8 ? 3 : 2);
	static __device__ __forceinline__ int addr(int g) { return (g >> LOG2T) * STRIDE + (g & (T - 1)); }
	static constexpr int caddr(int g) { return (g / T) * STRIDE + (g % T); }
	static constexpr int words(int n) { return ((n + T - 1) / T) * STRIDE; }
};

struct Regs {
	float2 S[16]; // the frame's spectrum, bins tf + slot*TF
	float mag[16]; // |S| of the same bins: loaded back from the LDS image after the median stage
};

template <int LOG2N>
struct FwdIn {
	const float* prev;
	const float* cur;
	const float* window;
	float* tail; // block calls, the call's last frame: receives its new hop (the next call's `prev`); else null
	int hop;
	__device__ __forceinline__ float2 operator()(int idx, int) const
	{
		ZH_CHK(idx < hop ? prev + idx : cur + (idx - hop), 1);
		ZH_CHK(window + idx, 1);
		const float x = idx < hop ? prev[idx] : cur[idx - hop];
		if (tail && idx >= hop) { // (a store between two loads: the second waits for the first.  One workgroup of a
			ZH_CHK(tail + (idx - hop), 1);
			tail[idx - hop] = x; // block call can afford that; the single-hop call copies the tail below instead)
		}
		return make_float2(x * window[idx], 0.0f); // window_functor hps.h:24-33
	}
};

// the same with the eight samples and window values of the thread already in registers (nfft 4096: the first pass asks
// thread tf for idx = slot * 256 + tf, slot < 8): the block build that finishes hops itself issues these loads before it
// waits for its publication words, so that the two round trips of the hand-off run under this one
struct FwdInPre {
	float x[8], w[8];
	__device__ __forceinline__ float2 operator()(int, int slot) const { return make_float2(x[slot] * w[slot], 0.0f); }
};

template <int T, int TFN>
struct FwdOut { // TFN: threads per frame (idx = tf + slot * TFN)
	Regs* r;
	float2* S;   // ring row (bins 0..n/2)
	float* mag;  // ring row
	int* img;    // LDS image of the magnitudes, word 0 = column -MID_AL
	int n, mid_al;
	__device__ __forceinline__ void operator()(int idx, float2 X, bool lower, int slot) const
	{
		r->S[slot] = X;
		// The spectrum of a real frame is exactly Hermitian (fft_dev.h), so |S[n-k]| == |S[k]| bit for bit:
		// the owner of bin k <= n/2 computes the double-precision hypot once and stores it for both bins.
		if (lower || idx == (n >> 1)) {
			const float m = zfft::cabs_exact(X.x, X.y); // complex_abs_functor hps.h:82-89
			const int key = __float_as_int(m);          // |S| >= +0: the bits are the ordering key
			const int mir = (idx == 0 || idx == (n >> 1)) ? idx : n - idx;
			// idx = tf + slot * TF with TF a multiple of the image's chunk length: the chunk arithmetic once per thread (tf's
			// share), the slot's share a compile-time offset -- for the mirror image counted downwards
			static_assert(TFN % T == 0, "whole chunks per slot");
			constexpr int CH = (TFN / T) * RtImage<T>::STRIDE;
			const int tfl = idx - slot * TFN;
			const int a0 = RtImage<T>::addr(tfl + mid_al), a1 = RtImage<T>::addr(n - tfl + mid_al);
			img[a0 + slot * CH] = key;
			img[(idx == 0 || idx == (n >> 1)) ? a0 + slot * CH : a1 - slot * CH] = key;
			if (S) { // the spectrum ring is only kept for single-hop calls
				ZH_CHK(S + idx, 1);
				S[idx] = X;
			}
			if (mag) { // the magnitude ring: single-hop calls and the last stft_width-1 frames of a block
				ZH_CHK(mag + idx, 1);
				ZH_CHK(mag + mir, 1);
				mag[idx] = m;
				mag[mir] = m;
			}
		}
	}
};

// HALF: only bins 0..N/2 and the last MID bins of the P row were filtered.  |S| is exactly Hermitian, so for
// MID < k < N/2 the window of bin N-k is the mirror image of the window of bin k and P[N-k] == P[k] bit for
// bit; the MID bins next to either end see different replicate borders (|S[0]| against |S[N-1]| = |S[1]|,
// SURVEY Q7) and are filtered on both sides.
template <int N, int TF, int MID, bool HALF>
struct InvIn {
	const Regs* r;
	const float* P; // LDS, natural bin order
	MaskCfg cfg;
	int which;
	__device__ __forceinline__ float2 operator()(int idx, int slot) const
	{
		int pi = idx; // idx = tf + slot*TF; `slot` is a constant after unrolling, so the first two tests fold
		if (HALF) {
			const int lo = slot * TF, hi = lo + TF - 1;
			if (hi <= N / 2)
				pi = idx;
			else if (lo > N / 2 && hi < N - MID)
				pi = N - idx;
			else
				pi = (idx > N / 2 && idx < N - MID) ? N - idx : idx;
		}
		const float2 z = r->S[slot];
		const float m = mask_value(which, r->mag[slot], P[pi], cfg);
		return make_float2(z.x * m, z.y * m); // apply_mask_functor hps.h:58-66
	}
};

// LEAN build (47 taps, 4096 bins, one output -- the realtime default): |S| is read back from the magnitude
// image and P from a compact half row, both INSIDE the frame image, so a workgroup needs 34 KB of LDS (four
// per CU) and no |S| registers.  pc: P of bins 0..2048 at pc[bin], of bins 3584..4095 at pc[2052 + bin - 3584].
constexpr int LEAN_PC_WORD = 5184;   // first word of the compact P row (behind the 259 * 20-word magnitude image)
constexpr int LEAN_PC_TAIL = 2052;   // pc index of bin 3584
constexpr int LEAN_EDGE_WORD = LEAN_PC_WORD + LEAN_PC_TAIL + 512 + 4; // wave-edge records of median47_core.h
// HARDP: the percussive output with a hard mask decided by exact comparison (masks.h hard_mask_exact) -- the realtime
// default; nothing else is compiled in: the generic build carries the soft-mask / divide variants of every element's
// mask as wave-uniform branches, a few thousand instructions of cold code threaded through the hot path.
template <int N, int TF, int MID, bool HARDP>
struct InvInLean {
	const Regs* r;
	const int* img; // magnitude image (word = bin + 24, 16-word chunks 20 apart)
	const float* pc;
	MaskCfg cfg;
	int which;
	double thr;     // != 0: hard percussive mask by exact comparison (masks.h hard_mask_exact)
	bool thr_inclusive;
	__device__ __forceinline__ float2 operator()(int idx, int slot) const
	{
		const int lo = slot * TF, hi = lo + TF - 1;
		int pi;
		if (hi <= N / 2)
			pi = idx;
		else if (lo > N / 2 && hi < N - MID)
			pi = N - idx;
		else
			pi = (idx > N / 2 && idx < N - MID) ? N - idx : (idx > N / 2 ? idx - (N - 512) + LEAN_PC_TAIL : idx);
		const int g = idx + 24;
		const int g0 = g - slot * TF; // (tf + 24: the chunk arithmetic once per thread, the slot's share an immediate)
		const float mag = __int_as_float(img[(g0 >> 4) * 20 + (g0 & 15) + slot * (TF / 16) * 20]);
		const float2 z = r->S[slot];
		float m;
		if constexpr (HARDP)
			m = hard_mask_exact(pc[pi], mag + FLT_EPSILON, thr, thr_inclusive);
		else // thr is wave-uniform: one branch, no divide on the taken side
			m = thr != 0.0 ? hard_mask_exact(pc[pi], mag + FLT_EPSILON, thr, thr_inclusive) : mask_value(which, mag, pc[pi], cfg);
		return make_float2(z.x * m, z.y * m); // apply_mask_functor hps.h:58-66
	}
};

// LEAN build with several outputs (hard masks): the two binary masks of the thread's 16 bins are compared ONCE, between
// the median stage and the first inverse transform, and kept as two bits per bin (as istft_hard_multi_kernel keeps
// them); every output's transform then reads nothing but the spectrum registers.  Same values as mask_value().
template <int N, int TF, int MID, bool HARDP>
__device__ __forceinline__ unsigned lean_mask_bits(const int* img, const float* pc, int tf, const MaskCfg& cfg, double thr_p,
                                                   double thr_h)
{
	unsigned bits = 0;
#pragma unroll
	for (int slot = 0; slot < 16; ++slot) {
		const int idx = tf + slot * TF, lo = slot * TF, hi = lo + TF - 1;
		int pi; // where P of bin idx lives in the compact half row (InvInLean)
		if (hi <= N / 2)
			pi = idx;
		else if (lo > N / 2 && hi < N - MID)
			pi = N - idx;
		else
			pi = (idx > N / 2 && idx < N - MID) ? N - idx : (idx > N / 2 ? idx - (N - 512) + LEAN_PC_TAIL : idx);
		const int g = idx + 24;
		const int g0 = tf + 24; // (InvInLean: the chunk arithmetic once per thread)
		const float h = __int_as_float(img[(g0 >> 4) * 20 + (g0 & 15) + slot * (TF / 16) * 20]), p = pc[pi]; // H = |S|: causal, SURVEY Q1
		(void)g;
		unsigned pm, hm;
		if constexpr (HARDP) { // both thresholds are there (the launcher checked): comparisons only, no divide variants compiled in
			pm = cfg.out_p ? (unsigned)(hard_mask_exact(p, h + FLT_EPSILON, thr_p) != 0.0f) : 0u; // hps.cu:501-505
			hm = cfg.out_h ? (unsigned)(hard_mask_exact(h, p + FLT_EPSILON, thr_h) != 0.0f) : 0u; // hps.cu:535-540
		}
		else {
			const HardThr t{thr_p, thr_h, 0, 0};
			pm = cfg.out_p ? (unsigned)(pmask_thr(h, p, cfg, t) != 0.0f) : 0u;
			hm = cfg.out_h ? (unsigned)(hmask_thr(h, p, cfg, t) != 0.0f) : 0u;
		}
		bits |= (pm | (hm << 1)) << (2 * slot);
	}
	return bits;
}
struct InvInBits {
	const Regs* r;
	unsigned bits;
	int which;
	__device__ __forceinline__ float2 operator()(int, int slot) const
	{
		const float pm = (float)((bits >> (2 * slot)) & 1u), hm = (float)((bits >> (2 * slot + 1)) & 1u);
		const float m = which == 0 ? pm : (which == 1 ? hm : 1 - (hm + pm)); // residual_mask_functor hps.h:35-43
		const float2 z = r->S[slot];
		return make_float2(z.x * m, z.y * m); // apply_mask_functor hps.h:58-66
	}
};

struct InvOut { // block builds
	float* Y;
	float cola;
	float* ready;       // single-hop calls: the finished hop = carry + first half of this frame (hps.cu:526-528 + :341-363)
	const float* carry; // second half of the previous frame, saved by this workgroup's housekeeping
	int hop;
	__device__ __forceinline__ void operator()(int idx, float2 x, bool, int) const
	{
		const float y = x.x * cola;
		ZH_CHK(Y + idx, 1);
		Y[idx] = y;
		if (ready && idx < hop) {
			ZH_CHK(ready + idx, 1);
			ZH_CHK(carry + idx, 1);
			ready[idx] = carry[idx] + y;
		}
	}
};
__device__ __forceinline__ unsigned xcc_id_of_cu()
{
	unsigned v;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
	return v & 15u;
}
template <bool KEEP>
struct InvOutRegT { // the single-hop build
	float* Y;
	float cola;
	float* ready;
	float cv[4];        // the thread's four carry samples (second half of the previous frame), idx = tf + slot*TF,
	int hop;            // slot < 4: in registers since the housekeeping -- a load here would queue behind the stores
	float* next;        // KEEP (resident kernel): receives the frame's second half, slots 4..7: the next hop's carries
	__device__ __forceinline__ void operator()(int idx, float2 x, bool, int slot) const
	{
		const float y = x.x * cola;
		ZH_CHK(Y + idx, 1);
		Y[idx] = y;
		// the finished hop goes out with system-scope (write-through) stores: publish_ready<true> then needs no write-back
		// of the L2 (a plain store to mapped host memory may stay in the L2 until one)
		if (ready && idx < hop) {
			ZH_CHK(ready + idx, 1);
			__hip_atomic_store(ready + idx, cv[slot & 3] + y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		if constexpr (KEEP) {
			if (slot >= 4)
				next[slot & 3] = y;
		}
	}
};
using InvOutReg = InvOutRegT<false>;

// the forward transform's input with the previous hop in registers (HopVar::keep_prev) and the new hop's samples handed
// back for the next one
struct FwdInKeep {
	const float* prev; // four registers: idx = slot*TF + tf, slot < 4
	const float* cur;
	const float* window;
	float* next;
	int hop;
	__device__ __forceinline__ float2 operator()(int idx, int slot) const
	{
		float x;
		if (slot < 4) {
			x = prev[slot & 3];
		}
		else {
			ZH_CHK(cur + (idx - hop), 1);
			x = cur[idx - hop];
			next[slot & 3] = x;
		}
		ZH_CHK(window + idx, 1);
		return make_float2(x * window[idx], 0.0f); // window_functor hps.h:24-33
	}
};

// Single-hop calls whose `ready` buffer is mapped host memory: the host does not wait for the launch to retire, it
// polls the word behind the hop.  Every thread makes its stores visible system-wide, the workgroup meets, one
// thread publishes the sequence number.
// LIGHT (the single-hop builds, per launch and resident): the samples of the hop were stored with system-scope
// (write-through) stores (InvOutRegT): they are on their way to host memory once the memory counter has counted them off, and the word
// follows them on the same path.  The system-scope fences of the plain form write back the XCD's whole L2 -- the Y row,
// the carries, the rings of this hop: device memory the host never reads -- twice per hop (release fence + release
// store); here nothing else is written back, the next hop's acquire fence (resident_next_hop) is what orders this
// workgroup's own reuse of that state.  (Plain stores to mapped host memory DO stay in the L2 until a write-back: with
// them this form hands the host zeros.)
// What the light form leans on beyond the memory model: write-through system-scope stores of gfx950 and the order of
// posted writes on the way to host memory.  Where either does not hold (relaxed-ordering PCIe, another MTYPE for mapped
// host memory) `release` -- RtFusedArgs::publish_seq == 2, option "publish_release" / ZEN_HIP_PUBLISH_RELEASE=1 -- selects
// the plain form at run time (a workgroup-uniform branch); tests/test_gpu_round5.py polls from the host over many
// thousand hops in both forms and compares every sample.
template <bool LIGHT = false>
__device__ __forceinline__ void publish_ready(unsigned* flag, unsigned seq, int tf, bool release = false)
{
	ZH_CHK(flag, 1);
	if (LIGHT && !release) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // (the compiler may not sink a sample store below the barrier)
		__syncthreads();
		if (tf == 0)
			__hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	else {
		__threadfence_system();
		__syncthreads();
		if (tf == 0)
			__hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// SINGLE: exactly one output is enabled (the realtime default, percussive only): the spectrum registers die
// in the first pass of the one inverse transform instead of living through a loop over outputs.
// (the kernel's body as a device function: rt_fused_kernel runs it once per workgroup, the resident kernel of
// rt_resident.hip once per hop it is handed; `bid` is the workgroup's place in the launch, blockIdx.x for the former)
template <int LOG2N, int W, int MINB, bool SINGLE, bool LEAN = false, bool HARDP = false, class HV = HopOfArgs>
__device__ __forceinline__ void rt_fused_body(const RtFusedArgs& a, const unsigned bid, const HV& hv, const int tid)
{
	using PL = Plan<LOG2N>;
	constexpr int N = PL::N, TF = PL::TF;
	constexpr int T = znet::outputs_per_thread(W), mid = W / 2;
	static_assert(T >= 4, "needs the 16-byte LDS path");
	using IM = RtImage<T>;
	constexpr int MID_AL = (mid + 3) & ~3, DELTA = MID_AL - mid;
	constexpr int NV = (DELTA + W + T - 1 + 3) / 4, NE = NV * 4;
	constexpr int NCHUNK = N / T, CPT = 16 / T; // median chunks in the row / per thread
	constexpr int IMG_WORDS = IM::words((NCHUNK - 1) * T + NE);
	static_assert(IMG_WORDS * 4 <= PL::LDS_FLOAT2 * 8, "magnitude image must fit in the FFT image");
	// 47 taps on 4096 bins (hop 1024 at 44.1 kHz, the headline configuration): the block scheme of
	// median47_core.h on the half spectrum
	// (with several outputs the spectrum registers live through a loop of inverse transforms; the block scheme's
	// pieces on top of them spill, 1.19 against 1.13 ms with the generic median stage: one output only)
	constexpr bool BLOCK47 = (W == 47 && LOG2N == 12 && (SINGLE || LEAN));
	static_assert(!LEAN || BLOCK47, "the lean layout is the 47-tap kernel (one output, or several with hard masks)");
	static_assert(!HARDP || LEAN, "the comparison-only mask builds are lean builds");
	static_assert(!LEAN || (LEAN_EDGE_WORD + 256) * 4 <= PL::LDS_FLOAT2 * 8, "lean layout must fit in the frame image");

	extern __shared__ float2 lds[];       // [FFT image | P row]; the magnitude image aliases the FFT image
	int* img = reinterpret_cast<int*>(lds);
	float* Prow = LEAN ? reinterpret_cast<float*>(img + LEAN_PC_WORD) : reinterpret_cast<float*>(lds + PL::LDS_FLOAT2);

	const int tf = tid, hop = a.hop; // (tid: threadIdx.x, made opaque per hop by the resident kernel)
	// XCD-aware order.  Workgroups are dealt to the eight XCDs round-robin (workgroup b runs on XCD b % 8), each with its
	// own L2; consecutive hops share an input hop (frame f = hops f-1, f).  Workgroup b therefore takes item
	// (b % 8) * (total / 8) + b / 8 of the launch: the workgroups of one XCD walk through consecutive hops, and the
	// shared hop is an L2 hit instead of a second fetch from memory by another XCD (measured before the change: 220 MB
	// read per 25 840-hop launch for 106 MB of input).  Any other placement is only slower, not wrong.
	const int total = a.n_streams * a.n_frames, xq = total >> 3, xr = total & 7, xcd = (int)(bid & 7);
	const int item = xcd * xq + (xcd < xr ? xcd : xr) + (bid >> 3);
	const int s = item / a.n_frames, f = item - s * a.n_frames;
	// diagnostic hook (tools/rt_latency.cpp --stamps): phase times of a single-hop call (100 MHz), kept in scalar
	// registers until the end -- a store to the host-mapped stamp buffer in front of a barrier would be waited for
	// there -- and only in the single-hop build.
	// (The block builds keep the direct store of round 1: their register allocation sits at the 168-VGPR limit and
	// any change to their code moves the spill count -- 19 registers as committed, 69 with this hook compiled out.)
	unsigned long long stamps[6] = {0, 0, 0, 0, 0, 0};
	auto stamp = [&](int k) {
		if constexpr (MINB == 1) {
			if (a.stamps)
				stamps[k] = __builtin_amdgcn_s_memrealtime();
		}
		else {
			if (a.stamps && bid == 0 && tf == 0)
				a.stamps[k] = __builtin_amdgcn_s_memrealtime();
		}
	};
	auto flush_stamps = [&]() {
		if constexpr (MINB == 1) {
			if (a.stamps && bid == 0 && tf == 0) {
#pragma unroll
				for (int k = 0; k < 6; ++k)
					a.stamps[k] = stamps[k];
			}
		}
	};
	stamp(0);
	const float* cur = hv.in() + (long long)s * a.in_stride + (long long)f * hop;

	// ---- housekeeping (as the extra block of stft_kernel): overlap-add carries, input tail.  The carry is
	// the second half of the previous call's last Y row; the workgroup that will overwrite that row (or, if
	// this call is shorter, the last one) saves it first.  The tail is the call's last hop: the next call's `prev`.
	// (hop == 4*TF: four elements per thread.)
	float cv1[4] = {0.f, 0.f, 0.f, 0.f}; // single-hop calls with one output: the carries of this hop
	if constexpr (MINB == 1 && HV::KEEP) { // resident kernel: the registers hold what the loads below would fetch
		static_assert(!HV::KEEP || SINGLE, "one output");
		const int w0 = a.out_id[0];
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			cv1[i] = hv.keep_carry[i];
			ZH_CHK(a.carry[w0] + ((long long)s * hop + tf + i * TF), 1);
			a.carry[w0][(long long)s * hop + tf + i * TF] = cv1[i]; // (memory stays what a per-launch hop would leave)
		}
		// the input tail is stored behind the forward transform, from the samples its first pass loads anyway
	}
	else if constexpr (MINB == 1) {
		const bool do_carry = hv.prev_frames() > 0 && f == (hv.prev_frames() - 1 < a.n_frames - 1 ? hv.prev_frames() - 1 : a.n_frames - 1);
		// Single-hop launches: all loads first, then the stores.  A store between two loads makes the second wait
		// for the first (the compiler cannot know they do not alias): stored from inside the transform's input
		// functor, the tail cost such a call eight dependent trips to memory.
		float tv[4], sv[3][4]; // (sv is only ever indexed by the unrolled loop counter: registers)
		if (do_carry) {
#pragma unroll
			for (int o = 0; o < 3; ++o) {
				if (!a.carry[o])
					continue;
				const float* y = a.Y[o] + (long long)s * a.y_stream_stride + (long long)(hv.prev_frames() - 1) * (2 * hop) + hop;
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					ZH_CHK(y + tf + i * TF, 1);
					sv[o][i] = y[tf + i * TF];
				}
			}
		}
		if constexpr (SINGLE) { // the carries the synthesis will add: from the previous call's Y row or the carry buffer
			const int w0 = a.out_id[0];
			const float* y = do_carry ? a.Y[w0] + (long long)s * a.y_stream_stride + (long long)(hv.prev_frames() - 1) * (2 * hop) + hop
			                          : a.carry[w0] + (long long)s * hop;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				ZH_CHK(y + tf + i * TF, 1);
				cv1[i] = y[tf + i * TF];
			}
		}
		if (f == a.n_frames - 1) {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				ZH_CHK(cur + tf + i * TF, 1);
				tv[i] = cur[tf + i * TF];
			}
		}
		if (do_carry) {
#pragma unroll
			for (int o = 0; o < 3; ++o) {
				if (!a.carry[o])
					continue;
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					ZH_CHK(a.carry[o] + ((long long)s * hop + tf + i * TF), 1);
					a.carry[o][(long long)s * hop + tf + i * TF] = sv[o][i];
				}
			}
		}
		if (f == a.n_frames - 1) {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				ZH_CHK(hv.tail_next() + ((long long)s * hop + tf + i * TF), 1);
				hv.tail_next()[(long long)s * hop + tf + i * TF] = tv[i];
			}
		}
	}
	else if (hv.prev_frames() > 0 && f == (hv.prev_frames() - 1 < a.n_frames - 1 ? hv.prev_frames() - 1 : a.n_frames - 1)) {
		for (int o = 0; o < 3; ++o) {
			if (!a.carry[o])
				continue;
			const float* y = a.Y[o] + (long long)s * a.y_stream_stride + (long long)(hv.prev_frames() - 1) * (2 * hop) + hop;
			float v[4];
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				ZH_CHK(y + tf + i * TF, 1);
				v[i] = y[tf + i * TF];
			}
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				ZH_CHK(a.carry[o] + ((long long)s * hop + tf + i * TF), 1);
				a.carry[o][(long long)s * hop + tf + i * TF] = v[i];
			}
		}
	}

	using OutT = std::conditional_t<MINB == 1, InvOutRegT<HV::KEEP>, InvOut>;
	float cnext[4] = {0.f, 0.f, 0.f, 0.f}; // KEEP: the second half of this frame, for the next hop
	// the carries of the output being synthesised: in registers since the housekeeping (one output), or read back
	// from the carry buffer this thread wrote there, before the transform starts (several outputs)
	auto pick_carry = [&](int which, float (&cw)[4]) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			if (!SINGLE)
				ZH_CHK(a.carry[which] + ((long long)s * hop + tf + i * TF), 1);
			cw[i] = SINGLE ? cv1[i] : a.carry[which][(long long)s * hop + tf + i * TF];
		}
	};
	stamp(1);
	// ---- block build that finishes hops itself (RtFusedArgs::out_direct), first part.  Output hop h = second half of
	// frame h-1 + first half of frame h (hps.cu:435-449, :526-528).  A workgroup cannot wait for its neighbour at its end
	// (they run side by side: measured 12 % on the whole kernel), so it finishes the hop of the workgroup DIRECT_BACK
	// items before it in its XCD's run, whose rows were written a generation of workgroups ago: it reads the two
	// publication words here, the eight samples of the two rows per thread right behind them (agent-scope loads: straight
	// from the L2 the workgroups of an XCD share, past the L1), both round trips under the latency of its own input
	// loads, and stores the sums after the forward transform.  A hop whose rows are not both published from this XCD with
	// this call's sequence number -- or that has no workgroup DIRECT_BACK items behind it -- is marked for
	// launch_rt_fused_fixup, which adds it up after the kernel: correct wherever and whenever the workgroups run.
	// What the hand-off leans on, none of it expressed in the memory model (the row stores are plain, the publication word
	// relaxed): on gfx942 / gfx950 (i) a store counted off by s_waitcnt vmcnt(0) has reached the L2 of its XCD, (ii) an
	// agent-scope (sc1) load bypasses the L1 and is served by that L2 when issued from the same XCD, (iii) XCC_ID is bits
	// 3:0 of hwreg 20 (xcc_id_of_cu), which is how a reader knows the writer shared its L2.  The Y-row stores of InvOut
	// must therefore stay plain stores (a nontemporal / streaming store may sit in a write-combining buffer past the wait).
	// Anything that does not hold falls back to the fix-up launch only if it shows up as an unpublished word: keep the
	// 40-repetition test (tests/test_gpu_round3.py test_fused_block_kernel_finishes_hops_itself) in the GPU tier.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "rt_fused.hip: the same-XCD hand-off of the block build is written for gfx942 / gfx950"
#endif
	constexpr bool DIRECT = HARDP && MINB != 1;
	constexpr int DIRECT_BACK = 128;
	unsigned dv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	bool dcomb = false;
	int dh = 0;
	FwdInPre pre;
	bool preloaded = false;
	auto preload_inputs = [&]() {
		static_assert(!DIRECT || (LOG2N == 12 && TF == 256), "slot <-> index map of FwdInPre");
		const float* pv = f > 0 ? cur - hop : hv.tail_prev() + (long long)s * hop;
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			const int idx = m * TF + tf;
			ZH_CHK(m < 4 ? pv + idx : cur + (idx - hop), 1);
			ZH_CHK(a.window + idx, 1);
			pre.x[m] = m < 4 ? pv[idx] : cur[idx - hop];
			pre.w[m] = a.window[idx];
		}
		preloaded = true;
	};
	if constexpr (DIRECT) {
		if (a.direct_on) {
			const int xi = bid >> 3, xcount = xq + (xcd < xr ? 1 : 0);
			if (xi + DIRECT_BACK >= xcount && tf == 0) {
				ZH_CHK(a.blk_need + item, 1);
				a.blk_need[item] = 1u; // nobody comes DIRECT_BACK items after this one
			}
			if (xi >= DIRECT_BACK) {
				dh = item - DIRECT_BACK;
				const int hs = dh / a.n_frames, hf = dh - hs * a.n_frames;
				unsigned v0 = 0, v1 = 0;
				if (hf > 0) {
					ZH_CHK(a.blk_flag + dh - 1, 2);
					v0 = __hip_atomic_load(a.blk_flag + dh - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					v1 = __hip_atomic_load(a.blk_flag + dh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				preload_inputs(); // (issued behind the two words, waited for behind them: see FwdInPre)
				if (hf > 0) {
					const unsigned want = (a.blk_seq << 4) | xcc_id_of_cu();
					dcomb = v0 == want && v1 == want;
				}
				if (dcomb && SINGLE) {
					const int which = a.out_id[0];
					const unsigned* row = reinterpret_cast<const unsigned*>(a.Y[which] + (long long)hs * a.y_stream_stride
					                                                        + (long long)hf * (2 * hop));
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						ZH_CHK(row - hop + tf + i * TF, 1);
						ZH_CHK(row + tf + i * TF, 1);
						dv[i] = __hip_atomic_load(row - hop + tf + i * TF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // frame h-1, second half
						dv[4 + i] = __hip_atomic_load(row + tf + i * TF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // frame h, first half
					}
				}
				else if (dcomb) { // several outputs: all loads in flight, then the sums (no registers to carry them further)
					unsigned dm[3][8];
#pragma unroll
					for (int oi = 0; oi < 3; ++oi) {
						if (oi < a.n_out) {
							const unsigned* row = reinterpret_cast<const unsigned*>(a.Y[a.out_id[oi]] + (long long)hs * a.y_stream_stride
							                                                        + (long long)hf * (2 * hop));
#pragma unroll
							for (int i = 0; i < 4; ++i) {
								ZH_CHK(row - hop + tf + i * TF, 1);
								ZH_CHK(row + tf + i * TF, 1);
								dm[oi][i] = __hip_atomic_load(row - hop + tf + i * TF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
								dm[oi][4 + i] = __hip_atomic_load(row + tf + i * TF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							}
						}
					}
#pragma unroll
					for (int oi = 0; oi < 3; ++oi) {
						if (oi < a.n_out) {
							float* o = a.out_direct[a.out_id[oi]] + (long long)hs * a.out_direct_stride + (long long)hf * hop;
#pragma unroll
							for (int i = 0; i < 4; ++i) {
								ZH_CHK(o + tf + i * TF, 1);
								o[tf + i * TF] = __uint_as_float(dm[oi][i]) + __uint_as_float(dm[oi][4 + i]);
							}
						}
					}
				}
				else {
					ZH_CHK(a.blk_need + dh, 1);
					a.blk_need[dh] = 1u; // (every thread that does not add its four samples says so: the decision is per thread)
				}
			}
		}
	}
	// Single-hop launches (MINB == 1, registers to spare): every twiddle of both transforms is loaded here, next
	// to the input samples, instead of in six dependent round trips later (fft_dev.h TwRegs).
	constexpr bool TWC = (MINB == 1);
	zfft::TwRegs<LOG2N> twr;
	if constexpr (TWC)
		twr.fill(tf, a.tw);
	// ---- analysis: hps.cu:452-472, :492
	Regs r;
	{
		FwdIn<LOG2N> in;
		in.prev = f > 0 ? cur - hop : hv.tail_prev() + (long long)s * hop;
		in.cur = cur;
		in.window = a.window;
		in.tail = (MINB == 1 || f != a.n_frames - 1) ? nullptr : hv.tail_next() + (long long)s * hop;
		in.hop = hop;
		const long long row = ((hv.row0() + f) % a.ring_rows) + (long long)s * a.ring_rows;
		FwdOut<T, TF> out;
		out.r = &r;
		out.S = a.S ? a.S + row * a.s_stride : nullptr;
		// A later use_sse_filter() (allowed at any time, hps.h:289) makes the next hop's causal time box filter
		// read the magnitudes of the stft_width-1 frames before it: a block keeps those rows up to date too.
		out.mag = (a.S || f >= a.n_frames - a.keep_mag_rows) ? a.mag + row * N : nullptr;
		out.img = img;
		out.n = N;
		out.mid_al = MID_AL;
		// the last pass overwrites the FFT image with the magnitude image: every thread has read its
		// inputs of that pass before the barrier inside PassRunner, so the aliasing is safe
		if constexpr (DIRECT) {
			if (!preloaded)
				preload_inputs();
			if (in.tail) { // the call's last frame hands its new hop to the next call
#pragma unroll
				for (int m = 4; m < 8; ++m) {
					ZH_CHK(in.tail + ((m - 4) * TF + tf), 1);
					in.tail[(m - 4) * TF + tf] = pre.x[m];
				}
			}
			zfft::fft_frame<LOG2N, false, true, false>(tf, lds, a.tw, pre, out, true);
		}
		else if constexpr (TWC && HV::KEEP) {
			float nx[4];
			FwdInKeep ink{hv.keep_prev, cur, a.window, nx, hop};
			zfft::fft_frame<LOG2N, false, true, false>(tf, lds, twr, ink, out, true);
#pragma unroll
			for (int i = 0; i < 4; ++i) { // the next call's previous hop: to memory as every single-hop call leaves it, and kept
				ZH_CHK(hv.tail_next() + ((long long)s * hop + tf + i * TF), 1);
				hv.tail_next()[(long long)s * hop + tf + i * TF] = nx[i];
				hv.keep_prev[i] = nx[i];
			}
		}
		else if constexpr (TWC)
			zfft::fft_frame<LOG2N, false, true, false>(tf, lds, twr, in, out, true);
		else
			zfft::fft_frame<LOG2N, false, true, false>(tf, lds, a.tw, in, out, true);
	}
	__syncthreads();
	if constexpr (DIRECT && SINGLE) {
		if (dcomb) { // the finished hop of DIRECT_BACK items ago: previous frame's second half + that frame's first half
			const int hs = dh / a.n_frames, hf = dh - hs * a.n_frames;
			float* o = a.out_direct[a.out_id[0]] + (long long)hs * a.out_direct_stride + (long long)hf * hop;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				ZH_CHK(o + tf + i * TF, 1);
				o[tf + i * TF] = __uint_as_float(dv[i]) + __uint_as_float(dv[4 + i]);
			}
		}
	}
	stamp(2);
	// replicate border of the magnitude row (ippBorderRepl)
	{
		const int v0 = img[IM::addr(MID_AL)], v1 = img[IM::addr(N - 1 + MID_AL)];
		for (int g = tf; g < MID_AL; g += TF)
			img[IM::addr(g)] = v0;
		for (int g = N + MID_AL + tf; g < (NCHUNK - 1) * T + NE; g += TF)
			img[IM::addr(g)] = v1;
	}
	__syncthreads();

	// ---- percussive estimate: frequency-direction median of the new row (hps.cu:496)
	if (a.diag == 1) { // timing diagnostic (zen_hip_set_option "rt_fused_diag"): no median, P = |S|
		for (int k = tf; k < (LEAN ? 2048 : N); k += TF)
			Prow[k] = __int_as_float(img[IM::addr(k + MID_AL)]);
	}
	else if constexpr (BLOCK47) {
		// Waves 0 and 1 filter bins 0..2047 (blocks 0..127).  Wave 2 supplies what is left: lanes 0..31 take
		// blocks 128..159 (bin 2048 is wanted, and blocks 128/129 feed wave 1's last lanes), lanes 32..63
		// blocks 224..255 (the last MID bins).  Wave 3 sits the stage out.  P[N-k] is read as P[k] (InvIn).
		static_assert(IM::STRIDE == zm47::RSTR && MID_AL == 24 && TF == 256, "image layout of median47_core.h");
		int(*edge)[64] = LEAN ? reinterpret_cast<int(*)[64]>(img + LEAN_EDGE_WORD) : reinterpret_cast<int(*)[64]>(Prow + N);
		const int lane = tf & 63, wave = __builtin_amdgcn_readfirstlane(tf >> 6);
		const int blk = wave < 2 ? tf : (lane < 32 ? 128 + lane : 192 + lane);
		zm47::Pieces pc;
		if (wave < 3)
			zm47::m47_sort_and_publish(img, edge, blk, lane, wave, tf == 0, blk == 255, pc);
		__syncthreads();
		if (wave < 3) {
			int out[16];
			zm47::m47_select(img, edge, blk, wave, pc, out);
			if constexpr (LEAN) { // compact half row: bins 0..2051 and 3584..4095
				const int at = blk >= 224 ? LEAN_PC_TAIL + (blk - 224) * 16 : blk * 16;
				if (blk <= 127 || blk >= 224) {
#pragma unroll
					for (int v = 0; v < 4; ++v)
						*reinterpret_cast<int4*>(&Prow[at + 4 * v]) =
						    make_int4(out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]);
				}
				else if (blk == 128) {
					*reinterpret_cast<int4*>(&Prow[at]) = make_int4(out[0], out[1], out[2], out[3]);
				}
			}
			else {
#pragma unroll
				for (int v = 0; v < 4; ++v)
					*reinterpret_cast<int4*>(&Prow[blk * 16 + 4 * v]) =
					    make_int4(out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]);
			}
		}
	}
	else {
#pragma unroll
	for (int ci = 0; ci < CPT; ++ci) {
		const int ch = tf * CPT + ci;
		int ld[NE], e[W + T - 1], out[T];
		const int* mine = &img[ch * IM::STRIDE];
#pragma unroll
		for (int v = 0; v < NV; ++v) {
			const int4 q = *reinterpret_cast<const int4*>(mine + IM::caddr(4 * v));
			ld[4 * v] = q.x;
			ld[4 * v + 1] = q.y;
			ld[4 * v + 2] = q.z;
			ld[4 * v + 3] = q.w;
		}
#pragma unroll
		for (int q = 0; q < W + T - 1; ++q)
			e[q] = ld[q + DELTA];
		znet::medians<W, T, W + T - 1>(e, out);
#pragma unroll
		for (int v = 0; v < T / 4; ++v)
			*reinterpret_cast<int4*>(&Prow[ch * T + 4 * v]) =
			    make_int4(out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]);
	}
	}
	__syncthreads(); // P row complete
	stamp(3);
	if constexpr (LEAN && !SINGLE) { // several outputs, hard masks: two bits per bin, then one transform per output
		if (a.diag == 2)
			return;
		const MaskCfg cfg{a.beta, a.beta_h, 0, a.power, 0, a.out_h, a.out_p};
		const unsigned bits = lean_mask_bits<N, TF, mid, HARDP>(img, Prow, tf, cfg, a.thr, a.thr_h);
		__syncthreads(); // the magnitude image and the P row are dead: the transforms overwrite them
		for (int oi = 0; oi < a.n_out; ++oi) {
			const int which = a.out_id[oi];
			InvInBits in{&r, bits, which};
			OutT out;
			out.Y = a.Y[which] + (long long)s * a.y_stream_stride + (long long)f * (2 * hop);
			out.cola = a.cola;
			out.ready = (a.n_frames == 1 && a.ready[which]) ? a.ready[which] + (long long)s * hop : nullptr;
			if constexpr (MINB == 1) {
				pick_carry(which, out.cv);
			}
			else {
				out.carry = a.carry[which] + (long long)s * hop;
			}
			out.hop = hop;
			// opaque per output (see istft_hard_multi_kernel): otherwise every LDS address and twiddle index of the
			// transform, all functions of tf alone, is hoisted out of this loop and kept in registers
			int tf_o = tf;
			int tw_off = 0; // (an opaque offset, not an opaque pointer: the table keeps its address space, istft.hip)
			asm volatile("" : "+v"(tf_o));
			asm volatile("" : "+s"(tw_off));
			const float2* tw_o = a.tw + tw_off;
			if constexpr (TWC)
				zfft::fft_frame<LOG2N, true, false, true>(tf_o, lds, twr, in, out, true);
			else
				zfft::fft_frame<LOG2N, true, false, true>(tf_o, lds, tw_o, in, out, true);
			if (out.ready && a.publish_seq)
				publish_ready<MINB == 1>(reinterpret_cast<unsigned*>(out.ready + hop), hv.seq(), tf, a.publish_seq == 2);
			__syncthreads(); // the frame image is reused by the next output
		}
		if constexpr (DIRECT) {
			if (a.direct_on) { // every output's row is in the L2 once the memory counter has drained: publish the item
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // no instruction on gfx950; keeps the compiler from moving stores below
				__syncthreads();
				if (tf == 0)
					__hip_atomic_store(a.blk_flag + item, (a.blk_seq << 4) | xcc_id_of_cu(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		return;
	}
	else if constexpr (LEAN) { // |S| and P are read inside the first inverse pass, which then meets at a barrier
		if (a.diag == 2)
			return;
		InvInLean<N, TF, mid, HARDP> in;
		in.r = &r;
		in.img = img;
		in.pc = Prow;
		in.cfg = MaskCfg{a.beta, a.beta_h, a.soft, a.power, 0, a.out_h, a.out_p};
		in.which = a.out_id[0];
		in.thr = (in.which == 0 && !a.soft) ? a.thr : 0.0;
		in.thr_inclusive = a.thr_inclusive != 0;
		if constexpr (DIRECT && SINGLE) {
			InvOut out;
			out.Y = a.Y[in.which] + (long long)s * a.y_stream_stride + (long long)f * (2 * hop);
			out.cola = a.cola;
			out.ready = nullptr;
			out.carry = nullptr;
			out.hop = hop;
			zfft::fft_frame<LOG2N, true, false, true, InvInLean<N, TF, mid, HARDP>, InvOut, true>(tf, lds, a.tw, in, out, true);
			if (a.direct_on) { // second part: once the row is in the L2 (the memory counter has drained), publish it
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // no instruction on gfx950; keeps the compiler from moving stores below
				__syncthreads();
				if (tf == 0)
					__hip_atomic_store(a.blk_flag + item, (a.blk_seq << 4) | xcc_id_of_cu(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			return;
		}
		OutT out;
		out.Y = a.Y[in.which] + (long long)s * a.y_stream_stride + (long long)f * (2 * hop);
		out.cola = a.cola;
		out.ready = (a.n_frames == 1 && a.ready[in.which]) ? a.ready[in.which] + (long long)s * hop : nullptr;
		if constexpr (MINB == 1) {
			pick_carry(in.which, out.cv);
			out.next = cnext;
		}
		else {
			out.carry = a.carry[in.which] + (long long)s * hop;
		}
		out.hop = hop;
		if constexpr (TWC)
			zfft::fft_frame<LOG2N, true, false, true, InvInLean<N, TF, mid, HARDP>, OutT, true>(tf, lds, twr, in, out, true);
		else
			zfft::fft_frame<LOG2N, true, false, true, InvInLean<N, TF, mid, HARDP>, OutT, true>(tf, lds, a.tw, in, out, true);
		if constexpr (HV::KEEP) {
#pragma unroll
			for (int i = 0; i < 4; ++i)
				hv.keep_carry[i] = cnext[i];
		}
		if (out.ready && a.publish_seq)
			publish_ready<MINB == 1>(reinterpret_cast<unsigned*>(out.ready + hop), hv.seq(), tf, a.publish_seq == 2);
		if constexpr (MINB == 1) {
			stamp(4);
			stamp(5);
			flush_stamps();
		}
		return;
	}
	// the thread's 16 magnitudes come back from the image (they were not held in registers across the
	// median stage); after the barrier the image is dead and the FFT image is free again
#pragma unroll
	for (int slot = 0; slot < 16; ++slot)
		r.mag[slot] = __int_as_float(img[IM::addr(tf + slot * TF + MID_AL)]);
	__syncthreads();

	// ---- synthesis per enabled output: hps.cu:498-579 (H = |S| of the same row: causal)
	auto synth = [&](int which) {
		InvIn<N, TF, mid, BLOCK47> in;
		in.r = &r;
		in.P = Prow;
		in.cfg = MaskCfg{a.beta, a.beta_h, a.soft, a.power, 0, a.out_h, a.out_p};
		in.which = which;
		OutT out;
		out.Y = a.Y[which] + (long long)s * a.y_stream_stride + (long long)f * (2 * hop);
		out.cola = a.cola;
		out.ready = (a.n_frames == 1 && a.ready[which]) ? a.ready[which] + (long long)s * hop : nullptr;
		if constexpr (MINB == 1) {
			pick_carry(which, out.cv);
			out.next = cnext;
		}
		else {
			out.carry = a.carry[which] + (long long)s * hop;
		}
		out.hop = hop;
		if constexpr (TWC)
			zfft::fft_frame<LOG2N, true, false, true>(tf, lds, twr, in, out, true);
		else
			zfft::fft_frame<LOG2N, true, false, true>(tf, lds, a.tw, in, out, true);
		if constexpr (HV::KEEP) {
#pragma unroll
			for (int i = 0; i < 4; ++i)
				hv.keep_carry[i] = cnext[i];
		}
		if (out.ready && a.publish_seq)
			publish_ready<MINB == 1>(reinterpret_cast<unsigned*>(out.ready + hop), hv.seq(), tf, a.publish_seq == 2);
	};
	if (a.diag == 2) { // timing diagnostic: no synthesis
		if (tf == 0)
			a.Y[a.out_id[0]][(long long)s * a.y_stream_stride + (long long)f * (2 * hop)] = Prow[tf] + r.mag[3] + r.S[5].x;
		return;
	}
	stamp(4);
	if constexpr (SINGLE) {
		synth(a.out_id[0]);
		stamp(5);
		flush_stamps();
	}
	else {
		for (int oi = 0; oi < a.n_out; ++oi) {
			synth(a.out_id[oi]);
			__syncthreads();
		}
	}
}

template <int LOG2N, int W, int MINB, bool SINGLE, bool LEAN = false, bool HARDP = false>
__global__ __launch_bounds__(Plan<LOG2N>::TF, MINB) void rt_fused_kernel(RtFusedArgs a)
{
	rt_fused_body<LOG2N, W, MINB, SINGLE, LEAN, HARDP>(a, blockIdx.x, HopOfArgs{a}, (int)threadIdx.x);
}

template <int LOG2N, int W, int MINB, bool SINGLE, bool LEAN = false, bool HARDP = false>
int launch_k(const RtFusedArgs& a, hipStream_t stream)
{
	using PL = Plan<LOG2N>;
	const size_t lds = LEAN ? sizeof(float2) * PL::LDS_FLOAT2
	                        : sizeof(float2) * PL::LDS_FLOAT2 + sizeof(float) * PL::N + (W == 47 && LOG2N == 12 && SINGLE ? 1024 : 0);
	auto kern = rt_fused_kernel<LOG2N, W, MINB, SINGLE, LEAN, HARDP>;
	if (lds > 64 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(kern, dim3((unsigned)((long long)a.n_streams * a.n_frames)), dim3(PL::TF), lds, stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

// single hops are latency-bound (all the registers the compiler wants); blocks of hops are compiled for
// BLOCK_MINB workgroups per CU
// The kernels for several outputs are instantiated in a translation unit of their own (rt_fused_multi.hip, which
// includes this file with ZEN_RT_FUSED_MULTI defined): the two sets want different scheduler flags (build.py).
#if defined(ZEN_RT_FUSED_MULTI_LEAN)
// (rt_fused_multi_lean.hip: the several-output lean kernel alone, compiled with the default scheduler -- no scratch
// there, 12 bytes per lane under the max-ILP strategy the other several-output builds want)
#elif defined(ZEN_RT_RESIDENT)
// ---- rt_resident.hip: the single-hop, one-output builds of the body inside a kernel that STAYS on its CU between hops.
// A per-hop call of the reference API costs ~18 us of which the kernel is ~11: the rest is the launch and the completion
// poll.  Here one workgroup is launched once and then handed hop after hop through a mailbox the host writes (fine-grained
// device memory behind the BAR, or pinned host memory): thread 0 polls the sequence word while the other waves sleep at
// a barrier; a new number -> acquire fence (the host wrote the hop's samples before the word), the per-hop arguments are
// derived from the launch's (input-tail buffers alternate, the ring row advances, the previous call was one frame), the
// body runs and publishes the finished hop behind its own sequence word as every single-hop launch does.  The kernel's
// life is bounded: it leaves when `stop` is set, when nothing has arrived for idle_ticks of the 100 MHz clock, or after
// max_hops; its last act is to tell the host how far it got (ResidentOut), and a hop posted in the window between its
// last look at the mailbox and that word is simply picked up by the next launch (hpr.hip resident_*).
template <int LOG2N, int W, bool LEAN>
constexpr size_t resident_lds_bytes() // launch_k's formula for the single-hop, one-output builds
{
	return LEAN ? sizeof(float2) * Plan<LOG2N>::LDS_FLOAT2
	            : sizeof(float2) * Plan<LOG2N>::LDS_FLOAT2 + sizeof(float) * Plan<LOG2N>::N + (W == 47 && LOG2N == 12 ? 1024 : 0);
}

template <int LOG2N, int W, bool LEAN, bool HARDP>
__global__ __launch_bounds__(Plan<LOG2N>::TF, 1) void rt_fused_resident_kernel(RtFusedArgs a0, const ResidentCtl* ctl, ResidentOut* ro,
                                                                               unsigned seq_start, unsigned long long idle_ticks,
                                                                               unsigned max_hops)
{
	extern __shared__ float2 lds_all[]; // (the body's image; the two command words sit behind it: no static LDS in front of the 16-byte accesses)
	unsigned* s_cmd = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds_all) + resident_lds_bytes<LOG2N, W, LEAN>());
	unsigned last = seq_start, k = 0;
	HopVar hv; // run_hop_fused's per-call arguments, for hop k of this launch (a0 itself stays in the kernel-argument segment)
	{ // what the first hop's housekeeping would load (HopVar::KEEP): the carries and the previous hop, once per launch
		constexpr int TF = Plan<LOG2N>::TF;
		const int w0 = a0.out_id[0];
		const float* y0 = a0.prev_frames > 0 ? a0.Y[w0] + (long long)(a0.prev_frames - 1) * (2 * a0.hop) + a0.hop : a0.carry[w0];
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			hv.keep_carry[i] = y0[threadIdx.x + i * TF];
			hv.keep_prev[i] = a0.tail_prev[threadIdx.x + i * TF];
		}
	}
	for (;;) {
		unsigned sq;
		if (!resident_next_hop(ctl, last, idle_ticks, k >= max_hops, s_cmd, &sq))
			break;
		hv.in_ = a0.in; // (the input buffer of the launch: a caller that hands over another pointer gets another launch, hpr.hip)
		hv.seq_ = sq;
		hv.row0_ = a0.row0 + k;
		hv.tail_prev_ = (k & 1u) ? a0.tail_next : a0.tail_prev; // the two input-tail buffers flip with every call
		hv.tail_next_ = (k & 1u) ? const_cast<float*>(a0.tail_prev) : a0.tail_next;
		hv.prev_frames_ = k > 0u ? 1 : a0.prev_frames;
		// (The thread index is NOT made opaque per hop here, unlike in rt_sse_resident_kernel: what the compiler hoists out of
		// the loop -- addresses, and the twiddle and window values the launch never changes -- costs registers the kernel has
		// (one workgroup per CU, spills go to AGPRs) and saves about a microsecond per hop: 12.8-13.9 against 13.9-15.0 us at
		// hop 1024, 9.7-10.4 against 10.9-11.5 at hop 256, same box.)
		rt_fused_body<LOG2N, W, 1, true, LEAN, HARDP, HopVar>(a0, 0u, hv, (int)threadIdx.x);
		__syncthreads(); // (every path of the body ends behind its last LDS access; s_cmd is rewritten next)
		last = sq;
		++k;
	}
	resident_leave(ro, last, k);
}

template <int LOG2N, int W, bool LEAN, bool HARDP>
int launch_res_k(const RtFusedArgs& a, const ResidentCtl* ctl, ResidentOut* ro, unsigned seq_start, unsigned long long idle_ticks,
                 unsigned max_hops, hipStream_t stream)
{
	using PL = Plan<LOG2N>;
	const size_t lds = resident_lds_bytes<LOG2N, W, LEAN>() + 16;
	auto kern = rt_fused_resident_kernel<LOG2N, W, LEAN, HARDP>;
	if (lds > 60 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(kern, dim3(1), dim3(PL::TF), lds, stream, a, ctl, ro, seq_start, idle_ticks, max_hops);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int LOG2N, int W>
int launch_res_t(const RtFusedArgs& a, const ResidentCtl* ctl, ResidentOut* ro, unsigned seq_start, unsigned long long idle_ticks,
                 unsigned max_hops, hipStream_t stream)
{
	if constexpr (LOG2N == 12 && W == 47) { // the builds launch_t picks for a single hop with one output
		const bool hardp = a.out_id[0] == 0 && !a.soft && a.thr != 0.0;
		return hardp ? launch_res_k<LOG2N, W, true, true>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream)
		             : launch_res_k<LOG2N, W, true, false>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	}
	return launch_res_k<LOG2N, W, false, false>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
}
#elif defined(ZEN_RT_FUSED_MULTI)
template <int LOG2N, int W>
int launch_multi_t(const RtFusedArgs& a, hipStream_t stream)
{
	if constexpr (LOG2N == 12 && W == 47) {
		// hard masks, blocks of hops: the lean layout with the masks kept as two bits per bin ("block_fused_minb" = 5:
		// the roomy layout below, 6: this translation unit's build of the lean kernel, for comparison)
		if (a.n_frames > 1 && !a.soft && a.thr != 0.0 && a.thr_h != 0.0 && g_opt_block_fused_minb != 5 && g_opt_block_fused_minb != 1
		    && g_opt_block_fused_minb != 2) {
			if (g_opt_block_fused_minb == 6)
				return launch_k<LOG2N, W, 3, false, true>(a, stream);
			return launch_rt_fused_multi_lean(a, stream);
		}
	}
	if (a.n_frames == 1 || g_opt_block_fused_minb == 1)
		return launch_k<LOG2N, W, 1, false>(a, stream);
	if (g_opt_block_fused_minb == 2)
		return launch_k<LOG2N, W, 2, false>(a, stream);
	return launch_k<LOG2N, W, 3, false>(a, stream);
}
#else
template <int LOG2N, int W>
int launch_t(const RtFusedArgs& a, hipStream_t stream)
{
	if (a.n_out != 1)
		return launch_rt_fused_multi(LOG2N, W, a, stream);
	if constexpr (LOG2N == 12 && W == 47) {
		// the headline configuration with one output: the lean layout (34 KB of LDS, no |S| registers).  Built for
		// three workgroups per CU: a fourth would fit in LDS but not in registers (128 VGPRs spill 60 of them:
		// 0.74 ms against 0.60 ms per 25 840 hops, measured); "block_fused_minb" = 4 selects that build, 5 the
		// roomy layout below, for comparison.
		if (a.n_out == 1 && g_opt_block_fused_minb != 5) {
			// the percussive output with a hard mask by comparison (the realtime default): builds with nothing else in them
			const bool hardp = a.out_id[0] == 0 && !a.soft && a.thr != 0.0;
			if (a.n_frames == 1 || g_opt_block_fused_minb == 1) // (1 on a block: the register-rich build, no scratch -- a diagnostic)
				return hardp ? launch_k<LOG2N, W, 1, true, true, true>(a, stream) : launch_k<LOG2N, W, 1, true, true>(a, stream);
			if (g_opt_block_fused_minb == 4)
				return launch_k<LOG2N, W, 4, true, true>(a, stream);
			return hardp ? launch_k<LOG2N, W, 3, true, true, true>(a, stream) : launch_k<LOG2N, W, 3, true, true>(a, stream);
		}
	}
	if (a.n_frames == 1 || g_opt_block_fused_minb == 1)
		return launch_k<LOG2N, W, 1, true>(a, stream);
	if (g_opt_block_fused_minb == 2)
		return launch_k<LOG2N, W, 2, true>(a, stream);
	return launch_k<LOG2N, W, 3, true>(a, stream);
}
#endif

} // namespace

#if defined(ZEN_RT_FUSED_MULTI_LEAN)
int launch_rt_fused_multi_lean(const RtFusedArgs& a, hipStream_t stream) { return launch_k<12, 47, 3, false, true, true>(a, stream); }
#elif defined(ZEN_RT_RESIDENT)
int launch_rt_fused_resident(int log2n, int freq_len, const RtFusedArgs& a, const ResidentCtl* ctl, ResidentOut* ro, unsigned seq_start,
                             unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream)
{
	if (a.n_out != 1 || a.n_frames != 1 || a.n_streams != 1 || !a.publish_seq)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "resident kernel: one stream, one output, single hops, host-mapped hop buffer");
	if (hop_lat_takes(log2n, a)) // (launch_rt_fused)
		return launch_rt_hop_lat_resident(log2n, freq_len, a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	switch (log2n * 100 + freq_len) {
	case 907: return launch_res_t<9, 7>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 1011: return launch_res_t<10, 11>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 1013: return launch_res_t<10, 13>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 1121: return launch_res_t<11, 21>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 1123: return launch_res_t<11, 23>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 1243: return launch_res_t<12, 43>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 1247: return launch_res_t<12, 47>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "no resident realtime kernel for nfft 2^%d, mask %d", log2n, freq_len);
	}
}
#elif defined(ZEN_RT_FUSED_MULTI)
int launch_rt_fused_multi(int log2n, int freq_len, const RtFusedArgs& a, hipStream_t stream)
{
	switch (log2n * 100 + freq_len) {
	case 907: return launch_multi_t<9, 7>(a, stream);
	case 1011: return launch_multi_t<10, 11>(a, stream);
	case 1013: return launch_multi_t<10, 13>(a, stream);
	case 1121: return launch_multi_t<11, 21>(a, stream);
	case 1123: return launch_multi_t<11, 23>(a, stream);
	case 1243: return launch_multi_t<12, 43>(a, stream);
	case 1247: return launch_multi_t<12, 47>(a, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "no fused realtime kernel for nfft 2^%d, mask %d", log2n, freq_len);
	}
}
#else
opt_t g_opt_block_fused_minb{0}; // 0: default (three workgroups per CU)

namespace {
// the hops a direct-output launch left (RtFusedArgs::blk_need): out = (previous frame's second half, or the carry for the
// first hop of the call) + this frame's first half -- finalize_kernel's sum for single hops
__global__ __launch_bounds__(256) void rt_fused_fixup_kernel(RtFusedArgs a)
{
	// one wavefront per item: all marks are read side by side (one trip to memory for the whole launch), a marked hop is
	// added up by its wavefront alone (the marked hops come in runs -- the end of every XCD's run of items -- so a
	// workgroup that walked through its items one after the other would serialise them)
	const int hop = a.hop, total = a.n_streams * a.n_frames;
	const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (item < total)
		ZH_CHK(a.blk_need + item, 1);
	if (item >= total || !a.blk_need[item])
		return;
	const int s = item / a.n_frames, f = item - s * a.n_frames;
	for (int oi = 0; oi < a.n_out; ++oi) {
		const int which = a.out_id[oi];
		const float* Y = a.Y[which] + (long long)s * a.y_stream_stride + (long long)f * (2 * hop);
		const float* prev = f == 0 ? a.carry[which] + (long long)s * hop : Y - hop;
		float* o = a.out_direct[which] + (long long)s * a.out_direct_stride + (long long)f * hop;
		for (int k = lane; k < hop; k += 64) {
			ZH_CHK(o + k, 1);
			ZH_CHK(prev + k, 1);
			ZH_CHK(Y + k, 1);
			o[k] = prev[k] + Y[k];
		}
	}
	if (lane == 0)
		a.blk_need[item] = 0u;
}
} // namespace

int launch_rt_fused_fixup(const RtFusedArgs& a, hipStream_t stream)
{
	const int total = a.n_streams * a.n_frames;
	hipLaunchKernelGGL(rt_fused_fixup_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

bool rt_fused_direct_out_available(int log2n, int freq_len, const RtFusedArgs& a)
{
	if (log2n != 12 || freq_len != 47 || a.n_frames <= 1 || a.soft || a.thr == 0.0 || g_opt_block_fused_minb != 0)
		return false;
	return a.n_out == 1 ? a.out_id[0] == 0 : a.thr_h != 0.0; // (the conditions of launch_t / launch_multi_t for the HARDP builds)
}

// (transform size, frequency mask) pairs with a fused kernel: hops 128..1024 at 44.1 and 48 kHz
bool rt_fused_available(int log2n, int freq_len)
{
	switch (log2n * 100 + freq_len) {
	case 907: case 1011: case 1013: case 1121: case 1123: case 1243: case 1247: return true;
	default: return false;
	}
}

int launch_rt_fused(int log2n, int freq_len, const RtFusedArgs& a, hipStream_t stream)
{
	// single hops: the frame over all four SIMDs of a CU, two wavefronts per SIMD at the larger sizes (rt_hop_lat.hip)
	if (hop_lat_takes(log2n, a))
		return launch_rt_hop_lat(log2n, freq_len, a, stream);
	switch (log2n * 100 + freq_len) {
	case 907: return launch_t<9, 7>(a, stream);
	case 1011: return launch_t<10, 11>(a, stream);
	case 1013: return launch_t<10, 13>(a, stream);
	case 1121: return launch_t<11, 21>(a, stream);
	case 1123: return launch_t<11, 23>(a, stream);
	case 1243: return launch_t<12, 43>(a, stream);
	case 1247: return launch_t<12, 47>(a, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "no fused realtime kernel for nfft 2^%d, mask %d", log2n, freq_len);
	}
}

#endif // ZEN_RT_FUSED_MULTI

} // namespace zen_hip_impl
