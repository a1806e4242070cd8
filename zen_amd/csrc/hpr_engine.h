// hpr_engine.h -- state of one streaming HPSS engine (zen_hip_hpr_t), shared by hpr.hip (the engine and its
// C-ABI) and hpri.hip (the two-pass offline driver built on two engines).
#pragma once
#include "common.h"
#include "rt_fused.h"

#include <utility>
#include <vector>

#include <climits>

// Where the finished hops of one output of a block call go (zen_hip_impl::hpr_process_spec, used by hpri.hip): the
// offline driver's "add P and R", "drop the lag*hop delay" and "truncate to the clip" folded into the overlap-add.
struct HprOutSpec {
	float* dst = nullptr;   // null: the output is not wanted
	long long stride = 0;   // floats between the streams' rows of dst
	long long shift = 0;    // the sample at stream position p (0 = first sample of the call) goes to dst[p - shift] ...
	long long len = 0;      // ... if 0 <= p - shift < len
	int add = -1;           // >= 0: id of an output whose finished hops are added first (P + R: hps.cu:153-160)
	long long dup_from = LLONG_MAX, dup_shift = 0, dup_len = 0; // positions >= dup_from also go to dst[p - dup_shift]
};

struct zen_hip_hpr {
	float fs;
	size_t hop, nwin, nfft;
	float beta;
	int l_harm, l_perc, lag;
	size_t W;
	int causality, log2n, mt, mf;
	float cola;
	bool out_h, out_p, out_r, use_sse, soft;
	size_t n_streams, max_hops; // max_hops: hops per chunk the buffers are allocated for right now ...
	size_t max_hops_cap;        // ... and the most they may grow to (zen_hip_hpr_process grows them to the call's size)
	long long ring_rows;
	size_t s_stride; // float2 per spectrum-ring row: nfft/2 + 1 bins, padded to a 64-byte multiple
	hipStream_t stream;

	float* d_window = nullptr;
	float2* d_tw = nullptr;
	float* d_tail[2] = {nullptr, nullptr};
	int tail_sel = 0;
	float2* d_S = nullptr;
	float* d_mag = nullptr;
	// block calls of the causal one-output hard-mask engine at nfft 4096 / 47 taps: the fused kernel finishes the hops
	// itself (RtFusedArgs::out_direct); per item a publication word and a "left for the fix-up" mark
	unsigned* d_blk_flag = nullptr;
	unsigned* d_blk_need = nullptr;
	unsigned blk_seq = 0;
	float* direct_out[3] = {nullptr, nullptr, nullptr}; // set by zen_hip_hpr_process around run_chunk: where output o's hops go
	long long direct_stride = 0;
	bool direct_done[3] = {false, false, false};        // run_chunk delivered output o's hops itself: no finalize launch
	float* d_H = nullptr;
	float* d_P = nullptr;
	float* d_Mh = nullptr;        // soft masks computed by the median kernel: the harmonic mask's rows (the percussive one's go where P would)
	unsigned* d_bits_t = nullptr; // the same in the synthesis threads' order (IstftArgs::bits_t)
	unsigned* d_bits = nullptr; // hard masks of the consumed rows as two bits per bin (IstftArgs::bits), blocks of frames only
	float* d_Y[3] = {nullptr, nullptr, nullptr};     // 0 percussive, 1 harmonic, 2 residual
	float* d_carry[3] = {nullptr, nullptr, nullptr};
	long long abs_frame = 0;
	size_t last_frames = 0;
	// A pass of the offline driver synthesised in runs (stft.h IstftRunArgs; hpr_process_spec decides per pass): the kernel
	// delivers the finished hops itself and keeps no Y rows.  run_*: the pass's destinations (groups of outputs, one per
	// HprOutSpec with a destination), what run_chunk needs beside the chunk; d_run_carry[b][o]: second half of a chunk's last
	// frame of output o for the next chunk (b alternates: a launch reads one set and writes the other).  rows_stale: such a
	// pass has run since the last reset -- the Y rows and carries do not hold the stream's last frame, and the other paths
	// refuse to continue the stream.
	bool run_mode = false, rows_stale = false;
	int run_n_groups = 0;
	struct RunGroup {
		int n_out, which[2];
		const HprOutSpec* spec;
	} run_groups[3];
	long long run_pos0 = 0;
	float* d_run_carry[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
	int run_sel = 0;
	size_t run_pick_M = 0; // pick_wide_run's last answer (hpr.hip): frames per chunk, groups signature -> run length, efficiency
	int run_pick_sig = 0, run_pick = 0;
	double run_pick_eff = 0.0;
	float* d_run_sink = nullptr; // IstftRunArgs::sink
	// An output that stops being computed (residual after use_soft_mask / use_sse_filter, hps.cu:562, :582-652)
	// still owes the second half of its last frame: the reference's accumulator is shifted, not cleared.
	// 0: nothing owed.  1: switched off, no hop processed since: copy_* still hands out the last finished hop(s), as
	// the reference's untouched accumulator does.  2: one call processed since: the first hop of that call's copy_* is
	// the parked tail, the rest zeros (hps.cu:435-449 rotates the accumulator of every ENABLED output each hop, whether
	// or not something is added to it).  Back to 0 with the call after that.
	int drain[3] = {0, 0, 0};
	// Single-hop calls (the realtime API): the synthesis kernel itself adds the two overlapping halves and
	// leaves the finished hop in `ready` -- host-mapped staging memory for a single stream, so that copy_*
	// is a fence plus a 4*hop-byte host copy instead of a second launch; device memory otherwise.
	float* ready_dev[3] = {nullptr, nullptr, nullptr};  // device-visible address the kernels write to
	float* ready_host[3] = {nullptr, nullptr, nullptr}; // host address of the same memory (single stream only)
	bool ready_valid[3] = {false, false, false};
	// rt_wide.hip (single hops at nfft 8192 / 16384): exchange buffer, grid-barrier words and their arrival count
	float2* d_wide_xch = nullptr;
	unsigned* d_wide_bar = nullptr;
	unsigned wide_arrivals = 0;
	unsigned wide_calls = 0;
	bool wide_votes_stale = false; // a resident launch of the cooperative kernel left a placement-vote word full (hpr.hip wide_args)
	unsigned* wide_fail_host = nullptr; // pinned, mapped: a grid barrier of rt_wide.hip that gave up waiting sets it;
	unsigned* wide_fail_dev = nullptr;  // every copy_* looks at it (a hop computed past a timed-out barrier is garbage)
	unsigned hop_seq = 0; // number of the last single-hop call; the kernels publish it behind the finished hop
	unsigned long long* dbg_stamps = nullptr;      // device alias of ...
	unsigned long long* dbg_stamps_host = nullptr; // ... the mapped stamp buffer of zen_hip_hpr_debug_stamps
	const void* out_query_dev = nullptr; // last copy_* destination looked up with hipPointerGetAttributes ...
	void* out_query_host = nullptr;      // ... and its host address (null: not host memory)
	bool async_pending = false;          // a copy_output_async since the last stream synchronise: the polling fast path of
	                                     // the synchronous copy_* synchronises first, so that "everything before is done" holds
	unsigned out_query_gen = 0;          // zen_hip_host_free count at the time of the lookup (a freed buffer's address may be reused)

	// Resident single-hop kernel (zen_hip_hpr_set_resident; rt_resident.hip): one workgroup stays on the device between the
	// hops of the stream and takes each hop from a mailbox instead of a launch.
	int res_idle_ms = 0;       // 0: off.  Else: the kernel leaves after this long without a hop (and is launched again by the next)
	bool res_active = false;   // launched and not yet seen to have left
	zen_hip_impl::ResidentCtl* res_ctl = nullptr;     // mailbox: host address == device address (fine-grained device memory), or
	zen_hip_impl::ResidentCtl* res_ctl_dev = nullptr; // pinned host memory and its device alias
	zen_hip_impl::ResidentOut* res_out = nullptr;     // pinned host memory
	zen_hip_impl::ResidentOut* res_out_dev = nullptr;
	hipStream_t res_stream = nullptr;
	hipEvent_t res_event = nullptr;
	zen_hip_impl::RtFusedArgs res_args;               // the arguments of the hop posted last: what a relaunch starts from
	unsigned long long res_launches = 0, res_hops = 0; // statistics (zen_hip_hpr_resident_stats)

	// zen_hip_hpr_process_host: device images of the caller's host block (input, one per delivered output), the two copy
	// streams and the events of the pieces
	float* hstage_in = nullptr;
	float* hstage_out[3] = {nullptr, nullptr, nullptr};
	size_t hstage_cap = 0; // floats
	hipStream_t hs_in = nullptr, hs_out = nullptr;
	std::vector<hipEvent_t> hevents;

	// profiling hook (bench.py): HIP events around every launch, per kernel class
	enum { K_STFT = 0, K_FREQ = 1, K_TIME = 2, K_ISTFT = 3, K_FINALIZE = 4, K_FUSED = 5, K_COUNT = 6 };
	bool prof = false;
	double prof_ms[K_COUNT] = {0, 0, 0, 0, 0, 0};
	unsigned long long prof_launches[K_COUNT] = {0, 0, 0, 0, 0, 0};
	unsigned long long prof_elements = 0; // elements filtered by the frequency-direction kernel
	struct Pending {
		int k;
		hipEvent_t e0, e1;
	};
	std::vector<Pending> prof_pending;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
};

namespace zen_hip_impl {
// zen_hip_hpr_process with (a) an input of which only the first in_valid samples of every row exist -- the rest reads
// as zero, no padded copy -- and (b) the outputs delivered as the specs say (indexed by output id: 0 percussive,
// 1 harmonic, 2 residual).  The engine must be an anticausal one (the two passes of HPRIOffline).
int hpr_process_spec(zen_hip_hpr* h, const float* in_dev, size_t n_hops, size_t in_stride, long long in_valid,
                     const HprOutSpec (&spec)[3]);
int hpr_reserve_hops(zen_hip_hpr* h, size_t n_hops); // grows the engine's buffers for calls of n_hops hops (up to its cap)
} // namespace zen_hip_impl
