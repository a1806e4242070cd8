// rt_hop_lat.hip -- ONE hop of the causal median path per call (HPRRealtime<GPU>::process_next_hop, libzen/hps.cu:334-339 ->
// HPR::process_next_hop :429-486 -> apply_median_filter :488-580) laid out for the latency of that call.
//
// rt_fused.hip's kernel is built for blocks of hops: 16 values per thread, three workgroups per CU, the magnitude image
// inside the frame image.  A single hop runs it as one workgroup of nfft/16 threads -- one wavefront at hop 256, two at
// hop 512, four at hop 1024 -- each working through 16 values per pass with nobody to overlap its LDS and barrier waits
// (hop 1024: 3.9 us forward, 3.1 us inverse of a 10.2 us kernel, profiles/r05_rt_latency.jsonl).  Here, as in
// rt_sse_lat.hip:
//
//   * lfft_dev.h's transform: 4 values per thread up to nfft 2048, 8 at nfft 4096 (128 to 512 threads: two wavefronts per
//     SIMD at the two large sizes), one barrier per pass; the spectrum stays in the registers of the thread that owns the bin
//     (idx = t + slot * TF) from the forward transform's last pass to the inverse transform's first;
//   * |S| goes to an LDS row image of its own (median_net.h's chunk layout with the replicate border), the frequency-direction
//     median reads it -- the sorting-network medians of median_net.h in chunks of four or eight outputs, every thread busy; at
//     nfft 4096 with 47 taps the block scheme of median47_core.h on three of the eight wavefronts -- and writes a P row;
//     nothing aliases the frame images, so no barrier guards them;
//   * the hard masks by exact comparison (masks.h mask_value_thr), the causal time median is the identity (H = |S| of the same
//     row);
//   * every load that does not depend on this frame at the top (twiddles, window, previous hop, carries); a resident launch
//     keeps twiddles, window, the previous hop and the carries in registers from hop to hop.
//
// Same arithmetic, same carry protocol, same rings as rt_fused.hip's single-hop builds: interchangeable call by call with them,
// with block calls and with the three-kernel path.  Option "no_hop_lat" (zen_hip_set_option) selects rt_fused.hip's builds.
//
// Same box, profiles/r05_rt_latency*.jsonl and the A/B runs of DESIGN.md section 5: per launch 14.4 / 16.1 / 16.6-18.0 -> 10.4-10.9 /
// 12.3-12.6 / 14.7-15.5 us at hop 256 / 512 / 1024, resident 9.2-10.1 / 10.7-11.4 / 11.7-12.3 -> 6.0-6.7 / 8.2-8.8 / 11.1-12.1 us.
// Two things about the masks mattered as much as the layout.  The HARDP builds (the percussive output alone, hard mask: the
// realtime default) have nothing else in them.  And in the builds that carry every variant the masks must not be
// loop-invariant: |S| and P do not change from one output to the next, so the compiler computed the masks of EVERY kind in
// front of the output loop -- the divisions of the variants nobody asked for included, 600 instructions per thread,
// 0.9-1.1 us by the stamps -- and chose by selects; an empty asm on the inputs inside the loop keeps each variant in its
// branch (harmonic output: 13.1-13.3 -> 12.5-12.8 us per call at hop 512, resident 9.1-9.2 -> 8.5-8.8).
#include "common.h"
#include "lfft_dev.h"
#include "masks.h"
#include "median47_core.h"
#include "median_net.h"
#include "rt_fused.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

// LDS row image of the median stage: T-word chunks spaced T+PAD apart (rt_fused.hip RtImage, median_net.hip RowImage)
template <int T>
struct LImage {
	static constexpr int PAD = T >= 8 ? 4 : 0;
	static constexpr int STRIDE = T + PAD;
	static constexpr int LOG2T = T == 16 ? 4 : (T == 8 ? 3 : 2);
	static __device__ __forceinline__ int addr(int g) { return (g >> LOG2T) * STRIDE + (g & (T - 1)); }
	static constexpr int caddr(int g) { return (g / T) * STRIDE + (g % T); }
	static constexpr int words(int n) { return ((n + T - 1) / T) * STRIDE; }
};

#ifndef ZEN_HOP_LAT_V12
#define ZEN_HOP_LAT_V12 3
#endif
#ifndef ZEN_HOP_LAT_V11
#define ZEN_HOP_LAT_V11 2
#endif
template <int LOG2N, int W>
struct HopGeo {
	static constexpr int LOG2V = LOG2N <= 10 ? 2 : (LOG2N == 11 ? ZEN_HOP_LAT_V11 : ZEN_HOP_LAT_V12);
	using PL = zfft::LPlan<LOG2N, LOG2V>;
	static constexpr int N = PL::N, V = PL::V, TF = PL::TF, QV = V / 4; // hop = nfft / 4 = QV * TF
	static constexpr int MID = W / 2;
	static constexpr bool BLOCK47 = (W == 47 && LOG2N == 12);
	static constexpr int T0 = znet::outputs_per_thread(W);
	static constexpr int T = BLOCK47 ? 16 : (T0 < V ? T0 : V); // outputs per median chunk
	static_assert(T >= 4, "needs the 16-byte LDS path");
	using IM = LImage<T>;
	static constexpr int MID_AL = (MID + 3) & ~3, DELTA = MID_AL - MID;
	static constexpr int NV = (DELTA + W + T - 1 + 3) / 4, NE = NV * 4;
	static constexpr int NCHUNK = N / T, CPT = V / T; // median chunks in the row / per thread (generic stage)
	static_assert(BLOCK47 || CPT * TF == NCHUNK, "every chunk has its thread");
	static constexpr int IMG_WORDS = (IM::words((NCHUNK - 1) * T + NE) + 3) & ~3;
	static constexpr size_t LDS_BYTES = sizeof(float2) * PL::LDS_FLOAT2 + sizeof(int) * IMG_WORDS + sizeof(float) * N + (BLOCK47 ? 1024 : 0);
	static_assert(!BLOCK47 || (IM::STRIDE == zm47::RSTR && MID_AL == 24 && TF >= 192), "image layout of median47_core.h");
};

template <int V>
struct HopRegs {
	float2 S[V]; // the frame's spectrum, bins t + slot * TF
	float mag[V]; // |S| of the same bins: from the LDS image after the median stage
};

template <int V>
struct HopFwdIn {
	const float* xw; // the windowed samples of slots 0 .. V/2-1 (the rest of the frame is the zero padding)
	__device__ __forceinline__ float2 operator()(int, int slot) const { return make_float2(xw[slot], 0.0f); }
};

template <class GEO>
struct HopFwdOut {
	HopRegs<GEO::V>* r;
	float2* S;  // ring row (bins 0..n/2)
	float* mag; // ring row, all n bins
	int* img;   // LDS image of the magnitudes, word 0 = column -MID_AL
	__device__ __forceinline__ void operator()(int idx, float2 X, bool lower, int slot) const
	{
		constexpr int n = GEO::N;
		r->S[slot] = X;
		// The spectrum of a real frame is exactly Hermitian (fft_dev.h), so |S[n-k]| == |S[k]| bit for bit:
		// the owner of bin k <= n/2 computes the double-precision hypot once and stores it for both bins.
		if (lower || (slot == GEO::V / 2 && idx == (n >> 1))) {
			const float m = zfft::cabs_exact(X.x, X.y); // complex_abs_functor hps.h:82-89
			r->mag[slot] = m;
			const int key = __float_as_int(m);          // |S| >= +0: the bits are the ordering key
			const int mir = (idx == 0 || idx == (n >> 1)) ? idx : n - idx;
			img[GEO::IM::addr(idx + GEO::MID_AL)] = key;
			img[GEO::IM::addr(mir + GEO::MID_AL)] = key;
			ZH_CHK(S + idx, 1);
			ZH_CHK(mag + idx, 1);
			ZH_CHK(mag + mir, 1);
			S[idx] = X;
			mag[idx] = m;
			mag[mir] = m;
		}
	}
};

template <int V>
struct HopInvIn {
	const float2* v; // the masked spectrum by slot
	__device__ __forceinline__ float2 operator()(int, int slot) const { return v[slot]; }
};

template <int V>
struct HopInvOut {
	float* Y;
	float cola;
	float* ready;    // the finished hop = carry + first half of this frame (hps.cu:526-528 + :341-363), or null
	const float* cv; // the thread's carry samples by slot (< V/4)
	float* keep;     // receives the second half of the frame by slot - V/4: the next hop's carries
	__device__ __forceinline__ void operator()(int idx, float2 x, bool, int slot) const
	{
		const float y = x.x * cola;
		ZH_CHK(Y + idx, 1);
		Y[idx] = y;
		if (slot < V / 4) { // idx < hop; a system-scope (write-through) store: see rt_fused.hip publish_ready
			if (ready) {
				ZH_CHK(ready + idx, 1);
				__hip_atomic_store(ready + idx, cv[slot < V / 4 ? slot : 0] + y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
		else
			keep[slot - V / 4 < V / 4 ? slot - V / 4 : 0] = y;
	}
};

template <int QV>
struct HopKeep { // what a resident launch keeps in registers from one hop to the next (memory is kept up to date all the same)
	float prev[QV];
	float carry[QV];
	bool valid;
};

// HARDP: the percussive output alone, hard mask by exact comparison (the realtime default, rt_fused.hip's HARDP builds): nothing
// else is compiled in -- the other outputs' carries, the soft-mask and divide variants of every bin's mask are cold code threaded
// through the hot path otherwise.
template <int LOG2N, int W, bool RESIDENT, bool HARDP, class HV>
__device__ __forceinline__ void rt_hop_lat_body(const RtFusedArgs& a, const unsigned bid, const HV& hv, const int t, const int ring_slot,
                                                const zfft::LTwRegs<LOG2N, HopGeo<LOG2N, W>::LOG2V>& twr,
                                                const float (&win)[HopGeo<LOG2N, W>::V / 2], HopKeep<HopGeo<LOG2N, W>::QV>& keep)
{
	using GEO = HopGeo<LOG2N, W>;
	using PL = typename GEO::PL;
	using IM = typename GEO::IM;
	constexpr int N = GEO::N, V = GEO::V, TF = GEO::TF, QV = GEO::QV, LOG2V = GEO::LOG2V, T = GEO::T, MID = GEO::MID, MID_AL = GEO::MID_AL;
	constexpr int NO = HARDP ? 1 : 3; // output ids that can be enabled
	extern __shared__ float2 lds[]; // [two FFT images | magnitude image | P row | wave-edge records of the 47-tap scheme]
	int* img = reinterpret_cast<int*>(lds + PL::LDS_FLOAT2);
	float* Prow = reinterpret_cast<float*>(img + GEO::IMG_WORDS);
	const int hop = a.hop, s = bid;
	const float* cur = hv.in() + (long long)s * a.in_stride;
	// diagnostic (tools/rt_latency.cpp --stamps): phase times of the call (100 MHz), kept in registers until the end
	unsigned long long stamps[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // (6..8: inside the last output's synthesis)
	auto stamp = [&](int k) {
		if (a.stamps)
			stamps[k] = __builtin_amdgcn_s_memrealtime();
	};
	stamp(0);

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
		for (int o = 0; o < NO; ++o) {
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
	stamp(1);

	// ---- analysis: window (window_functor hps.h:24-33), forward transform of the zero-padded frame, |S| (hps.cu:452-472, :492)
	HopRegs<V> r;
	{
		float xw[V / 2];
#pragma unroll
		for (int i = 0; i < V / 2; ++i)
			xw[i] = x[i] * win[i];
		HopFwdIn<V> in{xw};
		const long long row = ring_slot + (long long)s * a.ring_rows;
		HopFwdOut<GEO> out;
		out.r = &r;
		out.S = a.S + row * a.s_stride;
		out.mag = a.mag + row * N; // (a later use_sse_filter() reads the magnitudes of the frames before it, hps.h:289)
		out.img = img;
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
		for (int o = 0; o < NO; ++o) {
			if (!a.carry[o])
				continue;
#pragma unroll
			for (int i = 0; i < QV; ++i) {
				ZH_CHK(a.carry[o] + ((long long)s * hop + t + i * TF), 1);
				a.carry[o][(long long)s * hop + t + i * TF] = cv(o)[i];
			}
		}
	}
	__syncthreads(); // the magnitude image is complete
	stamp(2);
	// replicate border of the magnitude row (ippBorderRepl)
	{
		const int v0 = img[IM::addr(MID_AL)], v1 = img[IM::addr(N - 1 + MID_AL)];
		for (int g = t; g < MID_AL; g += TF)
			img[IM::addr(g)] = v0;
		for (int g = N + MID_AL + t; g < (GEO::NCHUNK - 1) * T + GEO::NE; g += TF)
			img[IM::addr(g)] = v1;
	}
	__syncthreads();

	// ---- percussive estimate: frequency-direction median of the new row (hps.cu:496)
	if constexpr (GEO::BLOCK47) {
		// rt_fused.hip's stage: waves 0 and 1 filter bins 0..2047 (blocks 0..127); wave 2 supplies what is left: lanes 0..31
		// blocks 128..159 (bin 2048 is wanted, and blocks 128/129 feed wave 1's last lanes), lanes 32..63 blocks 224..255
		// (the last MID bins).  P[N-k] is read as P[k] below.  The other wavefronts sit the stage out.
		int(*edge)[64] = reinterpret_cast<int(*)[64]>(Prow + N);
		const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
		const int blk = wave < 2 ? t : (lane < 32 ? 128 + lane : 192 + lane);
		zm47::Pieces pc;
		if (wave < 3)
			zm47::m47_sort_and_publish(img, edge, blk, lane, wave, t == 0, blk == 255, pc);
		__syncthreads();
		if (wave < 3) {
			int out[16];
			zm47::m47_select(img, edge, blk, wave, pc, out);
#pragma unroll
			for (int v = 0; v < 4; ++v)
				*reinterpret_cast<int4*>(&Prow[blk * 16 + 4 * v]) = make_int4(out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]);
		}
	}
	else {
#pragma unroll
		for (int ci = 0; ci < GEO::CPT; ++ci) {
			const int ch = t * GEO::CPT + ci;
			int ld[GEO::NE], e[W + T - 1], out[T];
			const int* mine = &img[ch * IM::STRIDE];
#pragma unroll
			for (int v = 0; v < GEO::NV; ++v) {
				const int4 q = *reinterpret_cast<const int4*>(mine + IM::caddr(4 * v));
				ld[4 * v] = q.x;
				ld[4 * v + 1] = q.y;
				ld[4 * v + 2] = q.z;
				ld[4 * v + 3] = q.w;
			}
#pragma unroll
			for (int q = 0; q < W + T - 1; ++q)
				e[q] = ld[q + GEO::DELTA];
			znet::medians<W, T, W + T - 1>(e, out);
#pragma unroll
			for (int v = 0; v < T / 4; ++v)
				*reinterpret_cast<int4*>(&Prow[ch * T + 4 * v]) = make_int4(out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]);
		}
	}
	// the magnitudes of the thread's upper-half bins from the image (written by the owners of their mirror bins; the lower half
	// is in registers since the forward transform)
#pragma unroll
	for (int slot = V / 2; slot < V; ++slot)
		r.mag[slot] = __int_as_float(img[IM::addr(t + slot * TF + MID_AL)]);
	__syncthreads(); // P row complete
	stamp(3);

	// ---- synthesis per enabled output: hps.cu:498-579 (H = |S| of the same row: causal)
	const MaskCfg cfg{a.beta, a.beta_h, a.soft, a.power, 0, a.out_h, a.out_p};
	const HardThr thr{a.thr, a.thr_h, a.thr_inclusive, a.thr_h_inclusive};
	float pv[V]; // P of the thread's bins
#pragma unroll
	for (int slot = 0; slot < V; ++slot) {
		const int idx = t + slot * TF;
		int pi = idx;
		if (GEO::BLOCK47) { // only bins 0..N/2 and the last MID bins were filtered: P[N-k] == P[k] in between (rt_fused.hip InvIn)
			const int lo = slot * TF, hi = lo + TF - 1;
			if (hi <= N / 2)
				pi = idx;
			else if (lo > N / 2 && hi < N - MID)
				pi = N - idx;
			else
				pi = (idx > N / 2 && idx < N - MID) ? N - idx : idx;
		}
		pv[slot] = Prow[pi];
	}
	stamp(4);
	for (int oi = 0; oi < (HARDP ? 1 : a.n_out); ++oi) {
		const int which = HARDP ? 0 : a.out_id[oi];
		float* ready = (a.ready[which]) ? a.ready[which] + (long long)s * hop : nullptr;
		{
			// the mask of every bin (mask_value_thr), the kind of mask decided once per output and not once per bin: the hard masks by
			// exact comparison are a handful of instructions each, the cold variants stay out of their way
			float2 v[V]; // S * mask (apply_mask_functor hps.h:58-66)
			if constexpr (!HARDP) {
				// (opaque per output: |S| and P do not change from one output to the next, so the masks of EVERY kind -- the
				// divisions of the variants nobody asked for included -- were computed in front of the loop and chosen by
				// selects: 600 instructions per thread, 0.9-1.1 us, where the branch taken needs 30)
#pragma unroll
				for (int i = 0; i < V; ++i)
					asm volatile("" : "+v"(r.mag[i]), "+v"(pv[i]));
			}
			if (HARDP || (which == 0 && !a.soft && a.thr != 0.0)) {
#pragma unroll
				for (int i = 0; i < V; ++i) {
					const float m = hard_mask_exact(pv[i], r.mag[i] + FLT_EPSILON, a.thr); // hps.cu:501-505
					v[i] = make_float2(r.S[i].x * m, r.S[i].y * m);
				}
			}
			else if constexpr (!HARDP) {
				if (which == 1 && !a.soft && a.thr_h != 0.0) {
#pragma unroll
					for (int i = 0; i < V; ++i) {
						const float m = hard_mask_exact(r.mag[i], pv[i] + FLT_EPSILON, a.thr_h); // hps.cu:535-540
						v[i] = make_float2(r.S[i].x * m, r.S[i].y * m);
					}
				}
				else {
#pragma unroll
					for (int i = 0; i < V; ++i) {
						const float m = mask_value_thr(which, r.mag[i], pv[i], cfg, thr);
						v[i] = make_float2(r.S[i].x * m, r.S[i].y * m);
					}
				}
			}
			stamp(6);
			HopInvIn<V> in{v};
			float cw[QV], kp[QV];
#pragma unroll
			for (int i = 0; i < QV; ++i)
				cw[i] = (HARDP || which == 0) ? cvp[i] : (which == 1 ? cvh[i] : cvr[i]);
			HopInvOut<V> out;
			out.Y = a.Y[which] + (long long)s * a.y_stream_stride;
			out.cola = a.cola;
			out.ready = ready;
			out.cv = cw;
			out.keep = kp;
			zfft::lfft_frame<LOG2N, LOG2V, true, false, true>(t, lds, twr, in, out);
			stamp(7);
			if (RESIDENT) {
#pragma unroll
				for (int i = 0; i < QV; ++i)
					keep.carry[i] = kp[i];
				keep.valid = true;
			}
		}
		if (ready && a.publish_seq) { // the host polls the word behind the finished hop (rt_fused.hip publish_ready)
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
		stamp(8);
		__syncthreads(); // (the next transform's first pass writes the image this one's last pass has read)
	}
	stamp(5);
	if (a.stamps && bid == 0 && t == 0) {
#pragma unroll
		for (int k = 0; k < 9; ++k)
			a.stamps[k] = stamps[k];
	}
}

template <int LOG2N, int W>
__device__ __forceinline__ void hop_window(const float* __restrict__ window, int t, float (&win)[HopGeo<LOG2N, W>::V / 2])
{
	using GEO = HopGeo<LOG2N, W>;
#pragma unroll
	for (int i = 0; i < GEO::V / 2; ++i) {
		ZH_CHK(window + t + i * GEO::TF, 1);
		win[i] = window[t + i * GEO::TF];
	}
}

template <int LOG2N, int W, bool HARDP>
__global__ __launch_bounds__((HopGeo<LOG2N, W>::TF)) void rt_hop_lat_kernel(RtFusedArgs a)
{
	using GEO = HopGeo<LOG2N, W>;
	const int t = (int)threadIdx.x;
	zfft::LTwRegs<LOG2N, GEO::LOG2V> twr;
	twr.fill(t, a.tw);
	float win[GEO::V / 2];
	hop_window<LOG2N, W>(a.window, t, win);
	HopKeep<GEO::QV> keep;
	keep.valid = false;
	// (the ring slot of the frame: one 64-bit remainder per call, scalar, behind the loads above)
	rt_hop_lat_body<LOG2N, W, false, HARDP>(a, blockIdx.x, HopOfArgs{a}, t, (int)(a.row0 % a.ring_rows), twr, win, keep);
}

// The same body inside a kernel that stays on its CU between the hops of a stream (rt_fused.hip rt_fused_resident_kernel:
// mailbox, idle time-out, exit word; hpr.hip resident_*).
template <int LOG2N, int W, bool HARDP>
__global__ __launch_bounds__((HopGeo<LOG2N, W>::TF)) void rt_hop_lat_resident_kernel(RtFusedArgs a0, const ResidentCtl* ctl, ResidentOut* ro,
                                                                                    unsigned seq_start, unsigned long long idle_ticks,
                                                                                    unsigned max_hops)
{
	using GEO = HopGeo<LOG2N, W>;
	extern __shared__ float2 lds_all[];
	unsigned* s_cmd = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds_all) + GEO::LDS_BYTES);
	const int t = (int)threadIdx.x;
	zfft::LTwRegs<LOG2N, GEO::LOG2V> twr;
	twr.fill(t, a0.tw);
	float win[GEO::V / 2];
	hop_window<LOG2N, W>(a0.window, t, win);
	HopKeep<GEO::QV> keep;
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
		rt_hop_lat_body<LOG2N, W, true, HARDP, HopVar>(a0, 0u, hv, t_o, ring_slot, twr, win, keep);
		__syncthreads();
		last = sq;
		++k;
		ring_slot = ring_slot + 1 == (int)a0.ring_rows ? 0 : ring_slot + 1;
	}
	resident_leave(ro, last, k);
}

template <int LOG2N, int W>
int launch_hop_t(const RtFusedArgs& a, hipStream_t stream)
{
	using GEO = HopGeo<LOG2N, W>;
	const size_t lds = GEO::LDS_BYTES;
	const bool hardp = a.n_out == 1 && a.out_id[0] == 0 && !a.soft && a.thr != 0.0;
	auto kern = hardp ? rt_hop_lat_kernel<LOG2N, W, true> : rt_hop_lat_kernel<LOG2N, W, false>;
	if (lds > 60 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(kern, dim3((unsigned)a.n_streams), dim3(GEO::TF), lds, stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int LOG2N, int W>
int launch_hop_res_t(const RtFusedArgs& a, const ResidentCtl* ctl, ResidentOut* ro, unsigned seq_start, unsigned long long idle_ticks,
                     unsigned max_hops, hipStream_t stream)
{
	using GEO = HopGeo<LOG2N, W>;
	const size_t lds = GEO::LDS_BYTES + 16;
	const bool hardp = a.n_out == 1 && a.out_id[0] == 0 && !a.soft && a.thr != 0.0;
	auto kern = hardp ? rt_hop_lat_resident_kernel<LOG2N, W, true> : rt_hop_lat_resident_kernel<LOG2N, W, false>;
	if (lds > 60 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(kern, dim3(1), dim3(GEO::TF), lds, stream, a, ctl, ro, seq_start, idle_ticks, max_hops);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace

// the (transform size, frequency mask) pairs of rt_fused_available
int launch_rt_hop_lat(int log2n, int freq_len, const RtFusedArgs& a, hipStream_t stream)
{
	if (a.n_frames != 1)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "rt_hop_lat: single hops");
	switch (log2n * 100 + freq_len) {
	case 907: return launch_hop_t<9, 7>(a, stream);
	case 1011: return launch_hop_t<10, 11>(a, stream);
	case 1013: return launch_hop_t<10, 13>(a, stream);
	case 1121: return launch_hop_t<11, 21>(a, stream);
	case 1123: return launch_hop_t<11, 23>(a, stream);
	case 1243: return launch_hop_t<12, 43>(a, stream);
	case 1247: return launch_hop_t<12, 47>(a, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "no single-hop kernel for nfft 2^%d, mask %d", log2n, freq_len);
	}
}

int launch_rt_hop_lat_resident(int log2n, int freq_len, const RtFusedArgs& a, const ResidentCtl* ctl, ResidentOut* ro, unsigned seq_start,
                               unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream)
{
	switch (log2n * 100 + freq_len) {
	case 907: return launch_hop_res_t<9, 7>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 1011: return launch_hop_res_t<10, 11>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 1013: return launch_hop_res_t<10, 13>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 1121: return launch_hop_res_t<11, 21>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 1123: return launch_hop_res_t<11, 23>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 1243: return launch_hop_res_t<12, 43>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 1247: return launch_hop_res_t<12, 47>(a, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "no resident single-hop kernel for nfft 2^%d, mask %d", log2n, freq_len);
	}
}

} // namespace zen_hip_impl
