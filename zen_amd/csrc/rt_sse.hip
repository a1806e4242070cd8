// rt_sse.hip -- one launch per hop for the causal SSE path (HPRRealtime<GPU>::use_sse_filter, BASELINE configs[4]):
// HPR::process_next_hop (libzen/hps.cu:429-486) -> apply_sse_filter (:582-652) for ONE hop per stream.
//
// The general engine needs four launches for such a call (analysis, time box, frequency box, synthesis).  Unlike
// the median path the causal SSE filter is not the identity in time: the harmonic estimate of a frame is the box
// mean of 1/|S|^2 over the frames before it (the taps after it replicate the frame itself, hps.h:265-268), so a
// BLOCK of hops cannot be one launch of independent workgroups -- but a single hop can: its history is in the
// magnitude ring, written by the calls before it.  One workgroup per stream:
//
//   window + zero-padded forward FFT            -> spectrum in registers (bins tf + slot*TF, as rt_fused.hip)
//   |S| (double-precision hypot per pair)       -> magnitude ring row (whole row: later calls read it) and
//   1/|S|^2                                      -> an LDS row with the replicate border (ippBorderRepl)
//   P = (l_perc+1) / boxmean_freq(1/|S|^2)       -> from the LDS row, taps added in ascending bin order
//   H = (l_harm+1) / boxmean_time(1/|S|^2)       -> history rows from the ring, ascending frame order, the frame
//                                                  itself for every tap at or after it
//   per output: Wiener mask (hps.h:132-140), S*mask, inverse FFT, *COLA, overlap-add with the carried half
//   (hps.cu:526-528) -> Y row + the finished hop in `ready`, sequence word published (rt_fused.hip publish).
//
// Same arithmetic, operation for operation, as stft_kernel + box_time_kernel + box_freq_kernel + istft_kernel
// (sum of the taps in ascending order, then / len, then (1/x)*factor): bit-identical outputs, interchangeable
// call by call with the four-launch path.
#include "common.h"
#include "exact_div.h"
#include "fft_dev.h"
#include "masks.h"
#include "rt_fused.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

using zfft::Plan;

constexpr int SSE_HALO = 128; // floats of replicate border on either side of the 1/|S|^2 row (box <= 255 taps)

struct SseRegs {
	float2 S[16];
};

struct SseFwdIn {
	const float* prev;
	const float* cur;
	const float* window;
	int hop;
	__device__ __forceinline__ float2 operator()(int idx, int) const
	{
		ZH_CHK(idx < hop ? prev + idx : cur + (idx - hop), 1);
		ZH_CHK(window + idx, 1);
		const float x = idx < hop ? prev[idx] : cur[idx - hop];
		return make_float2(x * window[idx], 0.0f); // window_functor hps.h:24-33
	}
};

struct SseFwdOut {
	SseRegs* r;
	float2* S;  // ring row
	float* mag; // ring row, all n bins
	float* pre; // LDS: pre[SSE_HALO + k] = (1 / (|S[k]| * |S[k]|)) * 1   (hps.h:91-98, :45-56)
	int n;
	__device__ __forceinline__ void operator()(int idx, float2 X, bool lower, int slot) const
	{
		r->S[slot] = X;
		if (lower || idx == (n >> 1)) { // |S[n-k]| == |S[k]| bit for bit: one hypot per pair
			const float m = zfft::cabs_exact(X.x, X.y);
			const float sq[1] = {m * m};
			float z[1];
			zdiv::recip_batch<1>(sq, z); // the short exact reciprocal (exact_div.h)
			const float p = z[0] * 1.0F;
			const int mir = (idx == 0 || idx == (n >> 1)) ? idx : n - idx;
			ZH_CHK(S + idx, 1);
			ZH_CHK(mag + idx, 1);
			ZH_CHK(mag + mir, 1);
			S[idx] = X;
			mag[idx] = m;
			mag[mir] = m;
			pre[SSE_HALO + idx] = p;
			pre[SSE_HALO + mir] = p;
		}
	}
};

struct SseInvIn {
	const SseRegs* r;
	const float* H; // LDS rows of the two estimates
	const float* P;
	MaskCfg cfg;
	int which;
	__device__ __forceinline__ float2 operator()(int idx, int slot) const
	{
		const float2 z = r->S[slot];
		const float m = mask_value(which, H[idx], P[idx], cfg);
		return make_float2(z.x * m, z.y * m); // apply_mask_functor hps.h:58-66
	}
};

struct SseInvOut {
	float* Y;
	float cola;
	float* ready;
	const float* cv; // the thread's four carry samples, idx = tf + slot*TF, slot < 4 (hop == 4*TF): in registers since
	int hop;         // before the transforms -- a load here would queue behind the stores of the previous outputs
	__device__ __forceinline__ void operator()(int idx, float2 x, bool, int slot) const
	{
		const float y = x.x * cola;
		ZH_CHK(Y + idx, 1);
		Y[idx] = y;
		if (idx < hop) { // hps.cu:526-528 + :341-363; a system-scope (write-through) store: see the publication below
			ZH_CHK(ready + idx, 1);
			__hip_atomic_store(ready + idx, cv[slot & 3] + y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
};

// Threads.  The transforms are the work of TF = nfft/16 threads (fft_dev.h).  A single wavefront running the
// dependent chains of the box sums (LDS read -> add, the IEEE divisions) is bound by instruction latency, not
// throughput, so:
//   * nfft <= 1024 (TF <= 64: the frame is synchronised inside its wavefront, frame_sync, and the other wavefronts
//     need not take part in the transforms): the workgroup has 512 threads.  While wavefront 0 transforms, the
//     seven others save the carries and the input tail and sum the history rows of the time box; the box sums of
//     the frame are theirs too, two or three bins per thread.  Wavefront 0 only transforms: it never waits on its
//     own stores of the spectrum / magnitude rows (on gfx9 stores and loads share one in-order counter).
//   * nfft >= 2048: every thread is a transform thread (the transform's barriers are workgroup barriers); 16 bins
//     per thread.
// The sums of a thread's bins are interleaved tap by tap: independent chains hide each other's latency.
template <int LOG2N>
struct SseGeo {
	static constexpr int N = Plan<LOG2N>::N, TF = Plan<LOG2N>::TF;
	static constexpr bool HELPERS = TF <= 64;
	static constexpr int NT = HELPERS ? 512 : TF;      // threads of the workgroup
	static constexpr int NH = HELPERS ? NT - 64 : NT;  // threads that own bins of the box sums ...
	static constexpr int H0 = HELPERS ? 64 : 0;        // ... from this one on
	static constexpr int BH = (N + NH - 1) / NH;       // bins per such thread: (t - H0) + i * NH (< N)
	static constexpr int CP = (N / 2 + NH - 1) / NH;   // samples of a hop per such thread
};

// (the kernel's body as a device function: rt_sse_kernel runs it once per workgroup, rt_sse_resident_kernel once per hop
// it is handed; what varies from hop to hop comes through `hv`, rt_fused.h)
template <int LOG2N, class HV = HopOfArgs>
__device__ __forceinline__ void rt_sse_body(const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, const unsigned bid,
                                            const HV& hv, const int tid)
{
	using PL = Plan<LOG2N>;
	using GEO = SseGeo<LOG2N>;
	constexpr int N = PL::N, TF = PL::TF, NT = GEO::NT, NH = GEO::NH, H0 = GEO::H0, BH = GEO::BH, CP = GEO::CP;
	extern __shared__ float2 lds[]; // [FFT image | 1/|S|^2 row with its borders | H row | P row]
	float* pre = reinterpret_cast<float*>(lds + PL::LDS_FLOAT2);
	float* Hrow = pre + N + 2 * SSE_HALO;
	float* Prow = Hrow + N;
	const int t = tid, hop = a.hop, s = bid; // (tid: threadIdx.x, made opaque per hop by the resident kernel)
	const bool fft_thread = t < TF;
	const float* cur = hv.in() + (long long)s * a.in_stride;
	// diagnostic (tools/rt_latency.cpp --stamps): phase times of the call, kept in registers until the end (a store
	// to the host-mapped stamp buffer in front of a barrier would be waited for there)
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
	const int ring_slot = (int)(ar % a.ring_rows);

	const int th = t - H0;
	const bool box_thread = t >= H0;
	float hist[BH];
	if (box_thread) {
		// the next call's previous hop, and the overlap-add carries: the second half of the previous call's last Y
		// row (as rt_fused_kernel).  Loads first, then stores: nothing orders a load behind an unrelated store.
		{
			float v[CP];
#pragma unroll
			for (int i = 0; i < CP; ++i) {
				ZH_CHK(cur + (th + i * NH < hop ? th + i * NH : hop - 1), 1);
				v[i] = cur[th + i * NH < hop ? th + i * NH : hop - 1];
			}
#pragma unroll
			for (int i = 0; i < CP; ++i)
				if (th + i * NH < hop) {
					ZH_CHK(hv.tail_next() + ((long long)s * hop + th + i * NH), 1);
					hv.tail_next()[(long long)s * hop + th + i * NH] = v[i];
				}
		}
		if (hv.prev_frames() > 0) {
			for (int o = 0; o < 3; ++o) {
				if (!a.carry[o])
					continue;
				const float* y = a.Y[o] + (long long)s * a.y_stream_stride + (long long)(hv.prev_frames() - 1) * (2 * hop) + hop;
				float v[CP];
#pragma unroll
				for (int i = 0; i < CP; ++i) {
					ZH_CHK(y + (th + i * NH < hop ? th + i * NH : hop - 1), 1);
					v[i] = y[th + i * NH < hop ? th + i * NH : hop - 1];
				}
#pragma unroll
				for (int i = 0; i < CP; ++i)
					if (th + i * NH < hop) {
						ZH_CHK(a.carry[o] + ((long long)s * hop + th + i * NH), 1);
						a.carry[o][(long long)s * hop + th + i * NH] = v[i];
					}
			}
		}
		// time box, history part: rows ar-mid_t .. ar-1 (clamped below at row 0, box_time_kernel), summed in
		// ascending tap order -- they do not depend on this frame.  Eight rows of loads are in flight at a time
		// (the rows were written by earlier launches: every load is a trip to HBM / MALL).
		for (int j0 = 0; j0 < mid_t; j0 += 8) {
			float m[8][BH];
#pragma unroll
			for (int jj = 0; jj < 8; ++jj) {
				// ring slot of row max(ar - d, 0), d = mid_t - j <= mid_t < ring_rows: one 64-bit remainder per call
				const int d = mid_t - (j0 + jj < mid_t ? j0 + jj : mid_t - 1);
				int slot = ar - d < 0 ? 0 : ring_slot - d;
				slot = slot < 0 ? slot + (int)a.ring_rows : slot;
				const float* mrow = a.mag + (slot + ring_base) * N;
#pragma unroll
				for (int i = 0; i < BH; ++i) {
					const int idx = th + i * NH;
					ZH_CHK(mrow + (idx < N ? idx : N - 1), 1);
					m[jj][i] = mrow[idx < N ? idx : N - 1];
				}
			}
#pragma unroll
			for (int jj = 0; jj < 8; ++jj) {
				if (j0 + jj < mid_t) {
					float sq[BH], z[BH]; // (1 / (|S| |S|)) * 1 by the short exact reciprocal (exact_div.h)
#pragma unroll
					for (int i = 0; i < BH; ++i)
						sq[i] = m[jj][i] * m[jj][i];
					zdiv::recip_batch<BH>(sq, z);
#pragma unroll
					for (int i = 0; i < BH; ++i) {
						const float v = z[i] * 1.0F;
						hist[i] = (j0 + jj) == 0 ? v : hist[i] + v;
					}
				}
			}
		}
	}
	stamp(1);
	SseRegs r;
	zfft::TwRegs<LOG2N> twr;
	float cv[3][4]; // the carries of this hop per output, as the synthesis will want them
	if (fft_thread) {
		// every twiddle of both transforms up front: one round trip instead of one per pass (fft_dev.h TwRegs)
		twr.fill(t, a.tw);
#pragma unroll
		for (int o = 0; o < 3; ++o) {
			if (!a.carry[o])
				continue;
			const float* y = hv.prev_frames() > 0
			                     ? a.Y[o] + (long long)s * a.y_stream_stride + (long long)(hv.prev_frames() - 1) * (2 * hop) + hop
			                     : a.carry[o] + (long long)s * hop;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				ZH_CHK(y + t + i * TF, 1);
				cv[o][i] = y[t + i * TF];
			}
		}
		SseFwdIn in;
		in.prev = hv.tail_prev() + (long long)s * hop;
		in.cur = cur;
		in.window = a.window;
		in.hop = hop;
		const long long row = ring_slot + ring_base;
		SseFwdOut out;
		out.r = &r;
		out.S = a.S + row * a.s_stride;
		out.mag = a.mag + row * N;
		out.pre = pre;
		out.n = N;
		zfft::fft_frame<LOG2N, false, true, false>(t, lds, twr, in, out, true);
	}
	__syncthreads();
	stamp(2);
	{ // replicate border of the 1/|S|^2 row
		const float v0 = pre[SSE_HALO], v1 = pre[SSE_HALO + N - 1];
		for (int g = t; g < SSE_HALO; g += NT) {
			pre[g] = v0;
			pre[SSE_HALO + N + g] = v1;
		}
	}
	__syncthreads();

	// ---- harmonic / percussive estimates of this frame (hps.cu:596-604)
	if (box_thread) {
		const float flen_t = (float)len_t, flen_f = (float)len_f;
		int idx[BH];
		float accf[BH], acct[BH], own[BH];
#pragma unroll
		for (int i = 0; i < BH; ++i) {
			idx[i] = th + i * NH < N ? th + i * NH : N - 1; // (a thread past the row repeats its last bin)
			accf[i] = pre[SSE_HALO + idx[i] - mid_f];
			own[i] = pre[SSE_HALO + idx[i]];
			acct[i] = mid_t == 0 ? own[i] : hist[i] + own[i]; // the history sum, then the frame itself ...
		}
		for (int j = 1; j < len_f; ++j) { // frequency box: taps idx-mid_f .. idx+mid_f in ascending order (box_freq_kernel)
			float t[BH]; // (the reads of a tap in flight together, then the sums: sse_block.hip)
#pragma unroll
			for (int i = 0; i < BH; ++i)
				t[i] = pre[SSE_HALO + idx[i] - mid_f + j];
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int i = 0; i < BH; ++i)
				accf[i] = accf[i] + t[i];
			__builtin_amdgcn_sched_barrier(0);
		}
		for (int j = mid_t + 1; j < len_t; ++j) { // ... for the remaining len_t - mid_t taps of the time box
#pragma unroll
			for (int i = 0; i < BH; ++i)
				acct[i] = acct[i] + own[i];
		}
		{ // the box means (sum / length) and their reciprocals, IEEE-exact in three instructions each (exact_div.h)
			float rf[BH], rt[BH], zf[BH], zt[BH];
			zdiv::div_const_batch<BH>(accf, flen_f, 1.0f / flen_f, rf);
			zdiv::div_const_batch<BH>(acct, flen_t, 1.0f / flen_t, rt);
			zdiv::recip_batch<BH>(rf, zf);
			zdiv::recip_batch<BH>(rt, zt);
#pragma unroll
			for (int i = 0; i < BH; ++i) {
				Prow[idx[i]] = zf[i] * fac_p;
				Hrow[idx[i]] = zt[i] * fac_h;
			}
		}
	}
	__syncthreads(); // the estimates are complete; nobody reads the FFT image between the transforms
	stamp(3);

	// ---- synthesis per computed output (percussive, harmonic; the SSE path has no residual, hps.cu:582-652)
	for (int oi = 0; oi < a.n_out; ++oi) {
		const int which = a.out_id[oi];
		float* ready = a.ready[which] + (long long)s * hop;
		if (fft_thread) {
			SseInvIn in;
			in.r = &r;
			in.H = Hrow;
			in.P = Prow;
			in.cfg = MaskCfg{a.beta, a.beta_h, a.soft, a.power, 1, a.out_h, a.out_p};
			in.which = which;
			SseInvOut out;
			out.Y = a.Y[which] + (long long)s * a.y_stream_stride;
			out.cola = a.cola;
			out.ready = ready;
			float cw[4];
#pragma unroll
			for (int i = 0; i < 4; ++i)
				cw[i] = which == 0 ? cv[0][i] : (which == 1 ? cv[1][i] : cv[2][i]);
			out.cv = cw;
			out.hop = hop;
			zfft::fft_frame<LOG2N, true, false, true>(t, lds, twr, in, out, true);
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
		__syncthreads();
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
__global__ __launch_bounds__(SseGeo<LOG2N>::NT) void rt_sse_kernel(RtFusedArgs a, int len_t, int len_f, float fac_h, float fac_p)
{
	rt_sse_body<LOG2N>(a, len_t, len_f, fac_h, fac_p, blockIdx.x, HopOfArgs{a}, (int)threadIdx.x);
}

template <int LOG2N>
constexpr size_t sse_lds_bytes() { return sizeof(float2) * Plan<LOG2N>::LDS_FLOAT2 + sizeof(float) * (3 * Plan<LOG2N>::N + 2 * SSE_HALO); }

// The same body inside a kernel that stays on its CU between the hops of a stream (rt_fused.hip rt_fused_resident_kernel:
// mailbox, idle time-out, exit word; hpr.hip resident_*).  The history rows of the time box are the ring rows this very
// workgroup wrote during the hops before: visible after the acquire fence of resident_next_hop.
template <int LOG2N>
__global__ __launch_bounds__(SseGeo<LOG2N>::NT) void rt_sse_resident_kernel(RtFusedArgs a0, int len_t, int len_f, float fac_h, float fac_p,
                                                                            const ResidentCtl* ctl, ResidentOut* ro, unsigned seq_start,
                                                                            unsigned long long idle_ticks, unsigned max_hops)
{
	extern __shared__ float2 lds_all[];
	unsigned* s_cmd = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds_all) + sse_lds_bytes<LOG2N>());
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
		// The thread index is made opaque per hop: otherwise every address of the body that depends on it alone is hoisted out
		// of the loop and kept in registers (nfft 512 / 1024: 140-184 bytes of scratch per lane at the 256 registers of a
		// 512-thread workgroup).
		int tid_o = (int)threadIdx.x;
		asm volatile("" : "+v"(tid_o));
		rt_sse_body<LOG2N, HopVar>(a0, len_t, len_f, fac_h, fac_p, 0u, hv, tid_o);
		__syncthreads();
		last = sq;
		++k;
	}
	resident_leave(ro, last, k);
}

template <int LOG2N>
int launch_sse_t(const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, hipStream_t stream)
{
	using PL = Plan<LOG2N>;
	const size_t lds = sse_lds_bytes<LOG2N>();
	auto kern = rt_sse_kernel<LOG2N>;
	if (lds > 64 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(kern, dim3((unsigned)a.n_streams), dim3(SseGeo<LOG2N>::NT), lds, stream, a, len_t, len_f, fac_h, fac_p);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int LOG2N>
int launch_sse_res_t(const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, const ResidentCtl* ctl, ResidentOut* ro,
                     unsigned seq_start, unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream)
{
	const size_t lds = sse_lds_bytes<LOG2N>() + 16;
	auto kern = rt_sse_resident_kernel<LOG2N>;
	if (lds > 60 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(kern, dim3(1), dim3(SseGeo<LOG2N>::NT), lds, stream, a, len_t, len_f, fac_h, fac_p, ctl, ro, seq_start, idle_ticks,
	                   max_hops);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace

int launch_rt_sse_resident(int log2n, const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, const ResidentCtl* ctl,
                           ResidentOut* ro, unsigned seq_start, unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream)
{
	if (a.n_out != 1 || a.n_frames != 1 || a.n_streams != 1 || !a.publish_seq)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "resident kernel: one stream, one output, single hops, host-mapped hop buffer");
	if (!g_opt_no_sse_lat) // the frame over all four SIMDs (rt_sse_lat.hip)
		return launch_rt_sse_lat_resident(log2n, a, len_t, len_f, fac_h, fac_p, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	switch (log2n) {
	case 9: return launch_sse_res_t<9>(a, len_t, len_f, fac_h, fac_p, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 10: return launch_sse_res_t<10>(a, len_t, len_f, fac_h, fac_p, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 11: return launch_sse_res_t<11>(a, len_t, len_f, fac_h, fac_p, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	case 12: return launch_sse_res_t<12>(a, len_t, len_f, fac_h, fac_p, ctl, ro, seq_start, idle_ticks, max_hops, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "no resident SSE kernel for nfft 2^%d", log2n);
	}
}

// hops 128 .. 1024 (transform sizes 512 .. 4096): one workgroup per stream
bool rt_sse_available(int log2n, int len_t, int len_f)
{
	return log2n >= 9 && log2n <= 12 && len_f <= 2 * SSE_HALO - 1 && len_t >= 1 && len_t <= 255;
}

int launch_rt_sse(int log2n, const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, hipStream_t stream)
{
	if (!g_opt_no_sse_lat)
		return launch_rt_sse_lat(log2n, a, len_t, len_f, fac_h, fac_p, stream);
	switch (log2n) {
	case 9: return launch_sse_t<9>(a, len_t, len_f, fac_h, fac_p, stream);
	case 10: return launch_sse_t<10>(a, len_t, len_f, fac_h, fac_p, stream);
	case 11: return launch_sse_t<11>(a, len_t, len_f, fac_h, fac_p, stream);
	case 12: return launch_sse_t<12>(a, len_t, len_f, fac_h, fac_p, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "no single-launch SSE kernel for nfft 2^%d", log2n);
	}
}

} // namespace zen_hip_impl
