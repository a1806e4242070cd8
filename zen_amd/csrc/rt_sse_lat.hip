// rt_sse_lat.hip -- rt_sse.hip's single hop of the causal SSE path (HPR::process_next_hop, libzen/hps.cu:429-486 ->
// apply_sse_filter :582-652) laid out for the latency of ONE call: the frame spread over all four SIMDs of a CU.
//
// rt_sse_kernel transforms with fft_dev.h's plan: 16 values per thread, so a 2048-point frame (hop 512) is two wavefronts
// and everything else -- the history rows of the time box, the box sums of 16 bins per thread, the masks -- is queued up
// in those two instruction streams one phase after the other (3.5 + 3.6 + 3.8 + 4.5 us, profiles/r05_rt_latency.jsonl:
// 1 600 and 2 850 instructions per lane for the two transforms at four to five cycles each).  Here:
//
//   * the frame's transform is lfft_dev.h's: V = 4 values per thread up to nfft 2048, 8 at nfft 4096 (128 to 512 threads),
//     one barrier per pass (two LDS images);
//   * bin idx = t + slot * TF is the thread's own from the forward transform's last pass to the inverse transform's
//     first: spectrum, 1/|S|^2, both box sums, both estimates and the masks stay in the thread's registers (rt_sse_kernel
//     hands the estimates over through two LDS rows and a barrier).  Only the frequency box looks at the neighbours'
//     1/|S|^2: one LDS row, one barrier.  The replicate border (ippBorderRepl) is a clamped index in the two slots
//     that can reach it, not a halo that wants a barrier of its own;
//   * every load that does not depend on this frame is requested at the top, before the first pass: twiddles, window,
//     the previous hop, the overlap-add carries, and the first eight history rows of the time box -- one trip to
//     memory under the forward transform instead of three in front of it.
//
// The arithmetic is rt_sse.hip's, operation for operation (sums in ascending tap order, the exact short divisions of
// exact_div.h): bit-identical outputs, interchangeable call by call with it and with the four-launch path.  Option
// "no_sse_lat" (zen_hip_set_option) selects rt_sse.hip's kernels.
#include "common.h"
#include "exact_div.h"
#include "lfft_dev.h"
#include "masks.h"
#include "rt_fused.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

// values per thread by transform size: nfft 2048 on 512 threads of 4 values (six passes; on 256 threads of 8, four passes,
// the hop took 16.2 / 11.4 us per launch / resident against 15.1 / 11.0: two wavefronts per SIMD overlap each other's LDS
// and barrier waits), nfft 4096 on 512 threads of 8
#ifndef ZEN_SSE_LAT_V11
#define ZEN_SSE_LAT_V11 2
#endif
#ifndef ZEN_SSE_LAT_V12
#define ZEN_SSE_LAT_V12 3
#endif
template <int LOG2N>
struct LatGeo {
	static constexpr int LOG2V = LOG2N <= 10 ? 2 : (LOG2N == 11 ? ZEN_SSE_LAT_V11 : ZEN_SSE_LAT_V12);
	using PL = zfft::LPlan<LOG2N, LOG2V>;
	static constexpr int N = PL::N, V = PL::V, TF = PL::TF;
	static constexpr int QV = V / 4; // a thread's samples per hop (hop = nfft / 4 = QV * TF)
	static_assert(TF >= 128, "the frequency box's border handling assumes len_f / 2 <= 127 <= TF");
	static constexpr size_t LDS_BYTES = sizeof(float2) * PL::LDS_FLOAT2 + sizeof(float) * N;
};

template <int V>
struct LatSpec {
	float2 S[V];
};

template <int V>
struct LatFwdIn {
	const float* xw; // the windowed samples of slots 0 .. V/2-1 (the rest of the frame is the zero padding)
	__device__ __forceinline__ float2 operator()(int, int slot) const { return make_float2(xw[slot], 0.0f); }
};

template <int V>
struct LatFwdOut {
	LatSpec<V>* r;
	float2* S;  // ring row
	float* mag; // ring row, all n bins
	float* pre; // LDS: pre[k] = (1 / (|S[k]| * |S[k]|)) * 1   (hps.h:91-98, :45-56)
	int n;
	__device__ __forceinline__ void operator()(int idx, float2 X, bool lower, int slot) const
	{
		r->S[slot] = X;
		if (lower || (slot == V / 2 && idx == (n >> 1))) { // |S[n-k]| == |S[k]| bit for bit: one hypot per pair
			const float m = zfft::cabs_exact(X.x, X.y);
			const float sq[1] = {m * m};
			float q[1];
			zdiv::recip_batch<1>(sq, q); // the short exact reciprocal (exact_div.h)
			const float p = q[0] * 1.0F;
			const int mir = (idx == 0 || idx == (n >> 1)) ? idx : n - idx;
			ZH_CHK(S + idx, 1);
			ZH_CHK(mag + idx, 1);
			ZH_CHK(mag + mir, 1);
			S[idx] = X;
			mag[idx] = m;
			mag[mir] = m;
			pre[idx] = p;
			pre[mir] = p;
		}
	}
};

template <int V>
struct LatInvIn {
	const float2* v; // the masked spectrum by slot
	__device__ __forceinline__ float2 operator()(int, int slot) const { return v[slot]; }
};

template <int V>
struct LatInvOut {
	float* Y;
	float cola;
	float* ready;
	const float* cv; // the thread's carry samples by slot (< V/4)
	float* keep;     // the second half of the frame by slot - V/4: the next hop's carries
	__device__ __forceinline__ void operator()(int idx, float2 x, bool, int slot) const
	{
		const float y = x.x * cola;
		ZH_CHK(Y + idx, 1);
		Y[idx] = y;
		if (slot < V / 4) { // idx < hop: hps.cu:526-528 + :341-363; a system-scope (write-through) store: see the publication below
			ZH_CHK(ready + idx, 1);
			__hip_atomic_store(ready + idx, cv[slot < V / 4 ? slot : 0] + y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		else
			keep[slot - V / 4 < V / 4 ? slot - V / 4 : 0] = y;
	}
};

// What a resident launch keeps in registers from one hop to the next (memory is kept up to date all the same: the next
// launch starts from it): the hop it has just windowed, and the second half of the frame it has just synthesised.
template <int QV>
struct LatKeep {
	float prev[QV];
	float carry[QV];
	bool valid;
};

template <int LOG2N, bool RESIDENT, class HV>
__device__ __forceinline__ void rt_sse_lat_body(const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, const unsigned bid,
                                                const HV& hv, const int t, const int ring_slot,
                                                const zfft::LTwRegs<LOG2N, LatGeo<LOG2N>::LOG2V>& twr, const float (&win)[LatGeo<LOG2N>::V / 2],
                                                LatKeep<LatGeo<LOG2N>::QV>& keep)
{
	using GEO = LatGeo<LOG2N>;
	using PL = typename GEO::PL;
	constexpr int N = GEO::N, V = GEO::V, TF = GEO::TF, QV = GEO::QV, LOG2V = GEO::LOG2V;
	extern __shared__ float2 lds[]; // [two FFT images | 1/|S|^2 row]
	float* pre = reinterpret_cast<float*>(lds + PL::LDS_FLOAT2);
	const int hop = a.hop, s = bid;
	const float* cur = hv.in() + (long long)s * a.in_stride;
	// diagnostic (tools/rt_latency.cpp --stamps): phase times of the call, kept in registers until the end
	unsigned long long stamps[5] = {0, 0, 0, 0, 0};
	const unsigned long long clk0 = a.stamps ? __builtin_amdgcn_s_memtime() : 0;
	auto stamp = [&](int k) {
		if (a.stamps)
			stamps[k] = __builtin_amdgcn_s_memrealtime();
	};
	stamp(0);
	const long long ar = hv.row0(); // absolute row of this frame
	const long long ring_base = (long long)s * a.ring_rows;
	const int mid_t = len_t >> 1, mid_f = len_f >> 1;

	// ---- everything this frame does not produce itself, requested at once
	float x[V / 2]; // slots 0 .. QV-1: the previous hop, QV .. 2QV-1: this one
	float cvp[QV], cvh[QV], cvr[QV]; // carries by output (three arrays: a runtime choice between them stays a select)
	auto cv = [&](int o) -> float(&)[QV] { return o == 0 ? cvp : (o == 1 ? cvh : cvr); };
	{
		const bool kept = RESIDENT && keep.valid;
#pragma unroll
		for (int i = 0; i < QV; ++i) {
			ZH_CHK(cur + t + i * TF, 1);
			x[QV + i] = cur[t + i * TF];
		}
		if (kept) {
#pragma unroll
			for (int i = 0; i < QV; ++i)
				x[i] = keep.prev[i];
		}
		else {
			const float* prev = hv.tail_prev() + (long long)s * hop;
#pragma unroll
			for (int i = 0; i < QV; ++i) {
				ZH_CHK(prev + t + i * TF, 1);
				x[i] = prev[t + i * TF];
			}
		}
#pragma unroll
		for (int o = 0; o < 3; ++o) {
			if (!a.carry[o]) {
#pragma unroll
				for (int i = 0; i < QV; ++i)
					cv(o)[i] = 0.0f;
				continue;
			}
			if (kept) {
#pragma unroll
				for (int i = 0; i < QV; ++i)
					cv(o)[i] = keep.carry[i];
				continue;
			}
			// the second half of the previous call's last Y row (hps.cu:526-528), or what an earlier call saved of it
			const float* y = hv.prev_frames() > 0
			                     ? a.Y[o] + (long long)s * a.y_stream_stride + (long long)(hv.prev_frames() - 1) * (2 * hop) + hop
			                     : a.carry[o] + (long long)s * hop;
#pragma unroll
			for (int i = 0; i < QV; ++i) {
				ZH_CHK(y + t + i * TF, 1);
				cv(o)[i] = y[t + i * TF];
			}
		}
	}
	// time box, history part: rows ar-mid_t .. ar-1 (clamped below at row 0, box_time_kernel), to be summed in ascending tap
	// order.  The first eight of them now (they were written by earlier hops: every load is a trip to the L2 / HBM).
	auto hist_row = [&](int j) -> const float* { // tap j < mid_t: ring slot of row max(ar - (mid_t - j), 0)
		const int d = mid_t - (j < mid_t ? j : mid_t - 1);
		int slot = ar - d < 0 ? 0 : ring_slot - d;
		slot = slot < 0 ? slot + (int)a.ring_rows : slot;
		return a.mag + (slot + ring_base) * N;
	};
	float hm[8][V];
#pragma unroll
	for (int jj = 0; jj < 8; ++jj) {
		const float* mrow = hist_row(jj);
		if (mid_t > 0) {
#pragma unroll
			for (int i = 0; i < V; ++i) {
				ZH_CHK(mrow + t + i * TF, 1);
				hm[jj][i] = mrow[t + i * TF];
			}
		}
	}
	stamp(1);

	// ---- window (window_functor hps.h:24-33) and forward transform of the zero-padded frame
	LatSpec<V> r;
	{
		float xw[V / 2];
#pragma unroll
		for (int i = 0; i < V / 2; ++i)
			xw[i] = x[i] * win[i];
		LatFwdIn<V> in{xw};
		const long long row = ring_slot + ring_base;
		LatFwdOut<V> out;
		out.r = &r;
		out.S = a.S + row * a.s_stride;
		out.mag = a.mag + row * N;
		out.pre = pre;
		out.n = N;
		zfft::lfft_frame<LOG2N, LOG2V, false, true, false>(t, lds, twr, in, out);
	}
	// the next call's previous hop, and this call's carries where the next call will look for them if it is not this launch
	// (stores behind the transform: nothing in it waits for them)
#pragma unroll
	for (int i = 0; i < QV; ++i) {
		ZH_CHK(hv.tail_next() + ((long long)s * hop + t + i * TF), 1);
		hv.tail_next()[(long long)s * hop + t + i * TF] = x[QV + i];
		if (RESIDENT)
			keep.prev[i] = x[QV + i];
	}
	if (hv.prev_frames() > 0) {
#pragma unroll
		for (int o = 0; o < 3; ++o) {
			if (!a.carry[o])
				continue;
#pragma unroll
			for (int i = 0; i < QV; ++i) {
				ZH_CHK(a.carry[o] + ((long long)s * hop + t + i * TF), 1);
				a.carry[o][(long long)s * hop + t + i * TF] = cv(o)[i];
			}
		}
	}
	__syncthreads(); // the 1/|S|^2 row is complete
	stamp(2);

	// ---- harmonic / percussive estimates of the thread's own bins (hps.cu:596-604)
	float Hv[V], Pv[V];
	{
		float hist[V];
		for (int j0 = 0; j0 < mid_t; j0 += 8) {
			if (j0 > 0) { // (more than eight history rows: the later ones eight at a time)
#pragma unroll
				for (int jj = 0; jj < 8; ++jj) {
					const float* mrow = hist_row(j0 + jj);
#pragma unroll
					for (int i = 0; i < V; ++i) {
						ZH_CHK(mrow + t + i * TF, 1);
						hm[jj][i] = mrow[t + i * TF];
					}
				}
			}
			// (1 / (|S| |S|)) * 1 by the short exact reciprocal (exact_div.h), 32 values per batch: one range vote each
			constexpr int RB = 32 / V; // rows per batch
#pragma unroll
			for (int jb = 0; jb < 8; jb += RB) {
				float sq[RB * V], q[RB * V];
#pragma unroll
				for (int jj = 0; jj < RB; ++jj)
#pragma unroll
					for (int i = 0; i < V; ++i)
						sq[jj * V + i] = hm[jb + jj][i] * hm[jb + jj][i];
				zdiv::recip_batch<RB * V>(sq, q);
#pragma unroll
				for (int jj = 0; jj < RB; ++jj) {
					if (j0 + jb + jj < mid_t) {
#pragma unroll
						for (int i = 0; i < V; ++i) {
							const float v = q[jj * V + i] * 1.0F;
							hist[i] = (j0 + jb + jj) == 0 ? v : hist[i] + v;
						}
					}
				}
			}
		}
		const float flen_t = (float)len_t, flen_f = (float)len_f;
		float accf[V], acct[V], own[V];
		int base[V];
#pragma unroll
		for (int i = 0; i < V; ++i) {
			const int idx = t + i * TF;
			base[i] = idx - mid_f;
			own[i] = pre[idx];
			// first tap; the replicate border (ippBorderRepl) can only be reached from the first and the last slot (mid_f <= TF)
			accf[i] = pre[i == 0 ? max(base[i], 0) : base[i]];
		}
#pragma unroll
		for (int i = 0; i < V; ++i)
			acct[i] = mid_t == 0 ? own[i] : hist[i] + own[i]; // the history sum, then the frame itself ...
		// frequency box: taps idx-mid_f .. idx+mid_f in ascending order (box_freq_kernel); the reads of four taps in flight
		// together, then their sums (len_f is odd: 1 + 4 q + 0 or 2 taps)
		auto tap_at = [&](int i, int j) -> float {
			return pre[i == 0 ? max(base[i] + j, 0) : (i == V - 1 ? min(base[i] + j, N - 1) : base[i] + j)];
		};
		int j = 1;
		for (; j + 4 <= len_f; j += 4) {
			float tap[4][V];
#pragma unroll
			for (int u = 0; u < 4; ++u)
#pragma unroll
				for (int i = 0; i < V; ++i)
					tap[u][i] = tap_at(i, j + u);
#pragma unroll
			for (int u = 0; u < 4; ++u)
#pragma unroll
				for (int i = 0; i < V; ++i)
					accf[i] = accf[i] + tap[u][i];
		}
		for (; j < len_f; ++j) {
			float tap[V];
#pragma unroll
			for (int i = 0; i < V; ++i)
				tap[i] = tap_at(i, j);
#pragma unroll
			for (int i = 0; i < V; ++i)
				accf[i] = accf[i] + tap[i];
		}
		for (int j = mid_t + 1; j < len_t; ++j) { // ... for the remaining len_t - mid_t taps of the time box
#pragma unroll
			for (int i = 0; i < V; ++i)
				acct[i] = acct[i] + own[i];
		}
		{ // the box means (sum / length) and their reciprocals, IEEE-exact in three instructions each (exact_div.h)
			float rf[V], rt[V], zf[V], zt[V];
			zdiv::div_const_batch<V>(accf, flen_f, 1.0f / flen_f, rf);
			zdiv::div_const_batch<V>(acct, flen_t, 1.0f / flen_t, rt);
			zdiv::recip_batch<V>(rf, zf);
			zdiv::recip_batch<V>(rt, zt);
#pragma unroll
			for (int i = 0; i < V; ++i) {
				Pv[i] = zf[i] * fac_p;
				Hv[i] = zt[i] * fac_h;
			}
		}
	}
	stamp(3);

	// ---- synthesis per computed output (percussive, harmonic; the SSE path has no residual, hps.cu:582-652)
	const MaskCfg cfg{a.beta, a.beta_h, a.soft, a.power, 1, a.out_h, a.out_p};
	for (int oi = 0; oi < a.n_out; ++oi) {
		const int which = a.out_id[oi];
		float* ready = a.ready[which] + (long long)s * hop;
		{
			float m[V]; // (which output: decided once, not once per bin)
			if (which == 0) {
#pragma unroll
				for (int i = 0; i < V; ++i)
					m[i] = pmask_value(Hv[i], Pv[i], cfg);
			}
			else if (which == 1) {
#pragma unroll
				for (int i = 0; i < V; ++i)
					m[i] = hmask_value(Hv[i], Pv[i], cfg);
			}
			else {
#pragma unroll
				for (int i = 0; i < V; ++i)
					m[i] = mask_value(which, Hv[i], Pv[i], cfg);
			}
			float2 v[V];
#pragma unroll
			for (int i = 0; i < V; ++i)
				v[i] = make_float2(r.S[i].x * m[i], r.S[i].y * m[i]); // apply_mask_functor hps.h:58-66
			LatInvIn<V> in{v};
			float cw[QV], kp[QV];
#pragma unroll
			for (int i = 0; i < QV; ++i)
				cw[i] = which == 0 ? cvp[i] : (which == 1 ? cvh[i] : cvr[i]);
			LatInvOut<V> out;
			out.Y = a.Y[which] + (long long)s * a.y_stream_stride;
			out.cola = a.cola;
			out.ready = ready;
			out.cv = cw;
			out.keep = kp;
			zfft::lfft_frame<LOG2N, LOG2V, true, false, true>(t, lds, twr, in, out);
			if (RESIDENT) {
#pragma unroll
				for (int i = 0; i < QV; ++i)
					keep.carry[i] = kp[i];
				keep.valid = true;
			}
		}
		if (a.publish_seq) { // the host polls the word behind the finished hop (rt_fused.hip publish_ready<true>: the samples
			// went out write-through; no write-back of the whole L2 per hop; publish_seq == 2: the release form instead)
			if (a.publish_seq == 2) {
				__threadfence_system();
				__syncthreads();
				if (t == 0)
					__hip_atomic_store(reinterpret_cast<unsigned*>(ready + hop), hv.seq(), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			}
			else {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__syncthreads();
				if (t == 0)
					__hip_atomic_store(reinterpret_cast<unsigned*>(ready + hop), hv.seq(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
		__syncthreads(); // (the next transform's first pass writes the image this one's last pass has read)
	}
	stamp(4);
	if (a.stamps && bid == 0 && t == 0) {
#pragma unroll
		for (int k = 0; k < 5; ++k)
			a.stamps[k] = stamps[k];
		a.stamps[5] = __builtin_amdgcn_s_memtime() - clk0; // shader clocks of the call (stamps 0..4: 100 MHz)
	}
}

template <int LOG2N>
__device__ __forceinline__ void lat_window(const float* __restrict__ window, int t, float (&win)[LatGeo<LOG2N>::V / 2])
{
#pragma unroll
	for (int i = 0; i < LatGeo<LOG2N>::V / 2; ++i) {
		ZH_CHK(window + t + i * LatGeo<LOG2N>::TF, 1);
		win[i] = window[t + i * LatGeo<LOG2N>::TF];
	}
}

template <int LOG2N>
__global__ __launch_bounds__(LatGeo<LOG2N>::TF) void rt_sse_lat_kernel(RtFusedArgs a, int len_t, int len_f, float fac_h, float fac_p)
{
	using GEO = LatGeo<LOG2N>;
	const int t = (int)threadIdx.x;
	zfft::LTwRegs<LOG2N, GEO::LOG2V> twr;
	twr.fill(t, a.tw);
	float win[GEO::V / 2];
	lat_window<LOG2N>(a.window, t, win);
	LatKeep<GEO::QV> keep;
	keep.valid = false;
	// (the ring slot of the frame: one 64-bit remainder per call, scalar, behind the loads above)
	rt_sse_lat_body<LOG2N, false>(a, len_t, len_f, fac_h, fac_p, blockIdx.x, HopOfArgs{a}, t, (int)(a.row0 % a.ring_rows), twr, win, keep);
}

// The same body inside a kernel that stays on its CU between the hops of a stream (rt_fused.hip rt_fused_resident_kernel:
// mailbox, idle time-out, exit word; hpr.hip resident_*).  Twiddles and window are loaded once per launch; the previous hop
// and the carries stay in registers from hop to hop; the history rows of the time box are the ring rows this very
// workgroup wrote during the hops before: visible after the acquire fence of resident_next_hop.
template <int LOG2N>
__global__ __launch_bounds__(LatGeo<LOG2N>::TF) void rt_sse_lat_resident_kernel(RtFusedArgs a0, int len_t, int len_f, float fac_h, float fac_p,
                                                                                 const ResidentCtl* ctl, ResidentOut* ro, unsigned seq_start,
                                                                                 unsigned long long idle_ticks, unsigned max_hops)
{
	using GEO = LatGeo<LOG2N>;
	extern __shared__ float2 lds_all[];
	unsigned* s_cmd = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds_all) + GEO::LDS_BYTES);
	const int t = (int)threadIdx.x;
	zfft::LTwRegs<LOG2N, GEO::LOG2V> twr;
	float win[GEO::V / 2];
	if constexpr (LOG2N < 12) {
		twr.fill(t, a0.tw);
		lat_window<LOG2N>(a0.window, t, win);
	}
	LatKeep<GEO::QV> keep;
	keep.valid = false;
	int ring_slot = (int)(a0.row0 % a0.ring_rows);
	unsigned last = seq_start, k = 0;
	for (;;) {
		unsigned sq;
		if (!resident_next_hop(ctl, last, idle_ticks, k >= max_hops, s_cmd, &sq))
			break;
		HopVar hv;
		hv.in_ = a0.in;
		hv.seq_ = sq;
		hv.row0_ = a0.row0 + k;
		hv.tail_prev_ = (k & 1u) ? a0.tail_next : a0.tail_prev;
		hv.tail_next_ = (k & 1u) ? const_cast<float*>(a0.tail_prev) : a0.tail_next;
		hv.prev_frames_ = k > 0u ? 1 : a0.prev_frames;
		// (the thread index opaque per hop: otherwise every address that depends on it alone is hoisted out of the loop and kept
		// in registers, rt_sse.hip)
		int t_o = t;
		asm volatile("" : "+v"(t_o));
		if constexpr (LOG2N >= 12) { // (512 threads: 256 registers each -- twiddles and window again per hop, from the L2, rather than spills)
			twr.fill(t_o, a0.tw);
			lat_window<LOG2N>(a0.window, t_o, win);
		}
		rt_sse_lat_body<LOG2N, true, HopVar>(a0, len_t, len_f, fac_h, fac_p, 0u, hv, t_o, ring_slot, twr, win, keep);
		__syncthreads();
		last = sq;
		++k;
		ring_slot = ring_slot + 1 == (int)a0.ring_rows ? 0 : ring_slot + 1;
	}
	resident_leave(ro, last, k);
}

template <int LOG2N>
int launch_lat_t(const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, hipStream_t stream)
{
	const size_t lds = LatGeo<LOG2N>::LDS_BYTES;
	auto kern = rt_sse_lat_kernel<LOG2N>;
	if (lds > 60 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(kern, dim3((unsigned)a.n_streams), dim3(LatGeo<LOG2N>::TF), lds, stream, a, len_t, len_f, fac_h, fac_p);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int LOG2N>
int launch_lat_res_t(const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, const ResidentCtl* ctl, ResidentOut* ro,
                     unsigned seq_start, unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream)
{
	const size_t lds = LatGeo<LOG2N>::LDS_BYTES + 16;
	auto kern = rt_sse_lat_resident_kernel<LOG2N>;
	if (lds > 60 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(kern, dim3(1), dim3(LatGeo<LOG2N>::TF), lds, stream, a, len_t, len_f, fac_h, fac_p, ctl, ro, seq_start, idle_ticks,
	                   max_hops);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace

int launch_rt_sse_lat_resident(int log2n, const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, const ResidentCtl* ctl,
                               ResidentOut* ro, unsigned seq_start, unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream)
{
	switch (log2n) {
	case 9: return launch_lat_res_t<9>(a, len_t, len_f, fac_h, fac_p, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 10: return launch_lat_res_t<10>(a, len_t, len_f, fac_h, fac_p, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 11: return launch_lat_res_t<11>(a, len_t, len_f, fac_h, fac_p, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 12: return launch_lat_res_t<12>(a, len_t, len_f, fac_h, fac_p, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "no resident SSE kernel for nfft 2^%d", log2n);
	}
}

int launch_rt_sse_lat(int log2n, const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, hipStream_t stream)
{
	switch (log2n) {
	case 9: return launch_lat_t<9>(a, len_t, len_f, fac_h, fac_p, stream);
	case 10: return launch_lat_t<10>(a, len_t, len_f, fac_h, fac_p, stream);
	case 11: return launch_lat_t<11>(a, len_t, len_f, fac_h, fac_p, stream);
	case 12: return launch_lat_t<12>(a, len_t, len_f, fac_h, fac_p, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "no single-launch SSE kernel for nfft 2^%d", log2n);
	}
}

} // namespace zen_hip_impl
