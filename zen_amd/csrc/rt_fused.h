// rt_fused.h -- launch interface of the fused causal-realtime kernel (rt_fused.hip): one workgroup per
// hop, one launch per call (one hop of every stream, or a block of consecutive hops).
#pragma once
#include <hip/hip_runtime.h>

namespace zen_hip_impl {

struct RtFusedArgs {
	const float* in;        // stream s: in[s*in_stride .. +n_frames*hop)
	long long in_stride;
	const float* tail_prev; // [n_streams][hop]
	float* tail_next;
	const float* window;
	const float2* tw;
	float2* S;              // spectrum ring (bins 0..nfft/2 per row); null: rings are not written
	long long s_stride;
	float* mag;             // magnitude ring
	int keep_mag_rows;      // block calls (S == null) still write the magnitude rows of their last keep_mag_rows frames
	long long ring_rows;
	long long row0;         // absolute row of the first frame
	int hop;
	int n_frames;           // consecutive hops per stream in this call
	int n_streams;
	int prev_frames;        // frames of the previous call (overlap-add carry source)
	float* carry[3];        // indexed by output id: 0 percussive, 1 harmonic, 2 residual (null = not computed)
	float* Y[3];            // indexed by output id
	long long y_stream_stride;
	float* ready[3];        // indexed by output id; single-hop calls: receives carry + first half of the new frame
	unsigned seq;           // single-hop calls: written to ready[o][hop] (as an integer) once ready[o] is complete,
	int publish_seq;        // when publish_seq != 0 (ready is host-mapped: the host polls it instead of synchronising);
	                        // 2: with the system-scope release form (option "publish_release"), see publish_ready
	int n_out;              // enabled outputs
	int out_id[3];
	float beta, beta_h, cola;
	int soft, power, out_h, out_p;
	// hard percussive mask without the divide (host: hard_mask_threshold): fl(p / d) >= beta  <=>  p >= thr * d
	// (or > when !thr_inclusive) in exact arithmetic, thr the rounding boundary just below beta; 0: divide
	double thr;
	int thr_inclusive;
	double thr_h;           // the same for the hard harmonic mask: fl(h / (p + Eps)) >= beta - Eps
	int thr_h_inclusive;
	// rt_wide.hip only: the percussive estimate row, the exchange buffer of the two-step transforms ([n_streams][nfft])
	// and the grid-barrier words ([n_streams][4]: arrivals, timeout flag, two vote words; bar_base = arrivals before this call)
	float* P;
	long long p_stream_stride;
	float2* xch;
	unsigned* bar;
	unsigned bar_base;
	int bar_parity;         // which of the two placement-vote words this call uses (alternates)
	unsigned* wide_fail;    // host-mapped word: set (and never cleared by the device) when a grid barrier gave up waiting
	// Block calls of the hard-mask lean builds (nfft 4096, 47 taps; one output or several): a workgroup finishes the hop
	// of the workgroup 128 items before it in its XCD's run -- second half of that frame's predecessor + its first half,
	// read from the L2 the workgroups of an XCD share -- and writes it to out_direct[output id] itself: no overlap-add
	// launch.  blk_flag[item]: (blk_seq << 4) | XCD of the workgroup that has written item's Y rows; blk_need[item] = 1:
	// hop not finished in the kernel (first hop of a stream or of an XCD's run, end of a run, rows not published from
	// this XCD): launch_rt_fused_fixup adds it up afterwards.  direct_on = 0: off.
	float* out_direct[3];
	long long out_direct_stride;
	int direct_on;
	unsigned* blk_flag;
	unsigned* blk_need;
	unsigned blk_seq;
	int diag;               // 0; 1 / 2: timing diagnostics of rt_fused_kernel (results are not valid)
	unsigned long long* stamps; // diagnostic: 8 s_memrealtime stamps (100 MHz) of workgroup 0's phases, or null
};

// rt_resident.hip: the mailbox the host writes (device-visible: fine-grained device memory behind the BAR, or pinned host
// memory) and the word the kernel leaves behind (pinned host memory).  See rt_fused_resident_kernel.
struct ResidentCtl {
	unsigned seq;              // number of the hop to process; the kernel acts when it differs from the last one it did
	unsigned stop;             // != 0: leave at the next look
	unsigned long long pad;    // (seq and stop are read as one 8-byte word)
};
struct ResidentOut {
	unsigned exited;           // 1: the kernel has left (written last, system-scope release)
	unsigned last_seq;         // the last sequence number it processed (the launch's seq_start if none)
	unsigned hops;             // hops it processed
	unsigned pad;
};
int launch_rt_fused_resident(int log2n, int freq_len, const RtFusedArgs& a, const ResidentCtl* ctl, ResidentOut* ro, unsigned seq_start,
                             unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream);
// the same for the causal SSE path (rt_sse.hip)
int launch_rt_sse_resident(int log2n, const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, const ResidentCtl* ctl,
                           ResidentOut* ro, unsigned seq_start, unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream);

#if defined(__HIPCC__)
// What changes from hop to hop of a stream: taken from the launch's arguments (HopOfArgs: one launch per call), or handed
// in by the resident kernel, which derives it for every hop it is given (HopVar).
struct HopOfArgs {
	static constexpr bool KEEP = false;
	const RtFusedArgs& a;
	__device__ __forceinline__ const float* in() const { return a.in; }
	__device__ __forceinline__ unsigned seq() const { return a.seq; }
	__device__ __forceinline__ long long row0() const { return a.row0; }
	__device__ __forceinline__ const float* tail_prev() const { return a.tail_prev; }
	__device__ __forceinline__ float* tail_next() const { return a.tail_next; }
	__device__ __forceinline__ int prev_frames() const { return a.prev_frames; }
};
struct HopVar {
	// KEEP: between the hops of a resident launch every thread keeps, in registers, the four samples of the previous hop it
	// will window again (keep_prev: idx = slot*TF + tf, slot < 4 of the forward transform's first pass) and the four samples
	// of the last frame's second half it will add to the next frame's first (keep_carry: the same thread stored them as
	// slots 4..7 of the inverse transform's last pass) -- the loads of the per-launch kernel's housekeeping, and the wait
	// for them in front of the input loads, go away.  Memory is kept up to date all the same (tail, carry, Y row): the
	// next launch starts from it.
	static constexpr bool KEEP = true;
	mutable float keep_prev[4], keep_carry[4];
	const float* in_;
	unsigned seq_;
	long long row0_;
	const float* tail_prev_;
	float* tail_next_;
	int prev_frames_;
	__device__ __forceinline__ const float* in() const { return in_; }
	__device__ __forceinline__ unsigned seq() const { return seq_; }
	__device__ __forceinline__ long long row0() const { return row0_; }
	__device__ __forceinline__ const float* tail_prev() const { return tail_prev_; }
	__device__ __forceinline__ float* tail_next() const { return tail_next_; }
	__device__ __forceinline__ int prev_frames() const { return prev_frames_; }
};

// The resident kernels' wait for work (all threads call it; s_cmd: two words of LDS nobody else uses).  Thread 0 polls the
// mailbox -- one 8-byte look per turn: the sequence word and the stop word side by side -- while the other waves sleep at
// the barrier; returns true with *sq = the new sequence number when a hop has been posted, false when the kernel is to
// leave (stop word, idle_ticks of the 100 MHz clock without a hop, or `leave`).  On true the caller's next loads see what
// the host wrote before the word and what this workgroup stored during the last hop (system-scope acquire fence).
__device__ __forceinline__ bool resident_next_hop(const ResidentCtl* ctl, unsigned last, unsigned long long idle_ticks, bool leave,
                                                   unsigned* s_cmd, unsigned* sq_out)
{
	if (threadIdx.x == 0) {
		const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
		unsigned cmd = 2, sq = last; // 1: a hop, 2: leave
		for (;;) {
			const unsigned long long w =
			    __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&ctl->seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			sq = (unsigned)w;
			if (sq != last) {
				cmd = 1;
				break;
			}
			if ((unsigned)(w >> 32) != 0u)
				break;
			if (__builtin_amdgcn_s_memrealtime() - t0 > idle_ticks)
				break;
			__builtin_amdgcn_s_sleep(1);
		}
		s_cmd[0] = leave ? 2u : cmd;
		s_cmd[1] = sq;
	}
	__syncthreads();
	const unsigned cmd = s_cmd[0];
	*sq_out = s_cmd[1];
	if (cmd != 1u)
		return false;
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
	return true;
}

// the kernel's last act: how far it got
__device__ __forceinline__ void resident_leave(ResidentOut* ro, unsigned last, unsigned k)
{
	if (threadIdx.x == 0) {
		ro->last_seq = last;
		ro->hops = k;
		__hip_atomic_store(&ro->exited, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}
#endif

bool rt_fused_available(int log2n, int freq_len);
// true if launch_rt_fused(log2n, freq_len, a) will run the build that finishes hops itself (RtFusedArgs::out_direct)
bool rt_fused_direct_out_available(int log2n, int freq_len, const RtFusedArgs& a);
// after such a launch: the hops it left (blk_need) through the plain overlap-add, every computed output
int launch_rt_fused_fixup(const RtFusedArgs& a, hipStream_t stream);
int launch_rt_fused(int log2n, int freq_len, const RtFusedArgs& a, hipStream_t stream);
int launch_rt_fused_multi(int log2n, int freq_len, const RtFusedArgs& a, hipStream_t stream); // n_out > 1 (rt_fused_multi.hip)
// rt_hop_lat.hip: single hops (n_frames == 1, any number of outputs and streams) with the frame spread over all four SIMDs of
// a CU -- what launch_rt_fused / launch_rt_fused_resident run for them unless option "no_hop_lat" is set
int launch_rt_hop_lat(int log2n, int freq_len, const RtFusedArgs& a, hipStream_t stream);
int launch_rt_hop_lat_resident(int log2n, int freq_len, const RtFusedArgs& a, const ResidentCtl* ctl, ResidentOut* ro, unsigned seq_start,
                               unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream);
int launch_rt_fused_multi_lean(const RtFusedArgs& a, hipStream_t stream); // nfft 4096, 47 taps, n_out > 1, hard masks, blocks (rt_fused_multi_lean.hip)

// rt_sse.hip: the causal SSE path (apply_sse_filter, hps.cu:582-652) for ONE hop per stream in one launch.  len_t /
// len_f: the odd box lengths (time, frequency), fac_h / fac_p: l_harm + 1, l_perc + 1 (hps.cu:599-604).
bool rt_sse_available(int log2n, int len_t, int len_f);
int launch_rt_sse(int log2n, const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, hipStream_t stream);
// rt_sse_lat.hip: the same call with the frame spread over all four SIMDs of the CU (what the two above run unless option
// "no_sse_lat" is set)
int launch_rt_sse_lat(int log2n, const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, hipStream_t stream);
int launch_rt_sse_lat_resident(int log2n, const RtFusedArgs& a, int len_t, int len_f, float fac_h, float fac_p, const ResidentCtl* ctl,
                               ResidentOut* ro, unsigned seq_start, unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream);

// rt_wide.hip: ONE hop per stream of the causal median path at nfft 8192 / 16384 in one launch, the frame spread
// over nfft/4096 cooperating workgroups.  rt_wide_arrivals: what a call adds to every stream's barrier word.
bool rt_wide_available(int log2n, int freq_len);
unsigned rt_wide_arrivals(int log2n, int n_out);
int launch_rt_wide(int log2n, int freq_len, const RtFusedArgs& a, hipStream_t stream);
// the same as a resident kernel (zen_hip_hpr_set_resident): `go` is a zeroed 8-byte word of device memory the cooperating
// workgroups of the launch agree through
int launch_rt_wide_resident(int log2n, int freq_len, const RtFusedArgs& a, const ResidentCtl* ctl, ResidentOut* ro, unsigned long long* go,
                            unsigned seq_start, unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream);

} // namespace zen_hip_impl
