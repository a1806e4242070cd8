// hpr.hip -- the streaming HPSS engine behind the C-ABI (the two-pass offline driver is hpri.hip).
//
//   zen_hip_hpr_*  : HPR<Backend::GPU> (libzen/hps.h:152-322, libzen/hps.cu:429-652) as a chunked
//                    streaming engine.  Per chunk of M hops (x n_streams):
//                      1 launch  stft_kernel      frames -> spectrum ring + magnitude ring
//                      1 launch  median/box freq  magnitude ring -> P   (consumed rows only)
//                      0-1 launch median/box time magnitude ring -> H   (identity when causal, Q1/Q2)
//                      1 launch  istft_kernel     masks, inverse FFT, *COLA -> Y
//                      1 launch  finalize_kernel  per requested output: overlap-add + copy-out
//                    against ~21-35 launches PER HOP in the reference.
//                    A causal engine (HPRRealtime) runs the whole chunk as ONE fused launch instead
//                    (rt_fused.hip) + the overlap-add.
//
// State between calls (what the reference keeps in HPR<B> members): previous hop of input (`input`),
// the last stft_width-1 spectra/magnitudes (`sliding_stft`, recomputed `s_mag`), and per output the
// second half of the last synthesised frame (`*_out[hop:]`).  Rows live in a ring addressed by an
// absolute frame counter, so nothing is ever shifted (the reference moves (W-1)*nfft complex per hop).
#include "common.h"
#include "memguard.h"
#include "filters.h"
#include "hpr_engine.h"
#include "host_pipe.h"

#include <algorithm>
#include <functional>
#include "sse_block.h"
#include "masks.h"
#include "rt_fused.h"
#include "stft.h"

#include <cfloat>
#include <climits>
#include <cmath>
#include <utility>
#include <vector>

using namespace zen_hip_impl;


namespace {
struct ProfScope {
	zen_hip_hpr* e;
	int k;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	ProfScope(zen_hip_hpr* e_, int k_)
	    : e(e_)
	    , k(k_)
	{
		if (!e->prof)
			return;
		if (!e->prof_pool.empty()) {
			e0 = e->prof_pool.back().first;
			e1 = e->prof_pool.back().second;
			e->prof_pool.pop_back();
		}
		else if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
			e0 = e1 = nullptr;
			return;
		}
		(void)hipEventRecord(e0, e->stream);
	}
	~ProfScope()
	{
		if (!e0)
			return;
		(void)hipEventRecord(e1, e->stream);
		e->prof_pending.push_back({k, e0, e1});
		e->prof_launches[k] += 1;
	}
};
} // namespace

namespace {

int which_index(unsigned which)
{
	switch (which) {
	case ZEN_HIP_OUTPUT_PERCUSSIVE: return 0;
	case ZEN_HIP_OUTPUT_HARMONIC: return 1;
	case ZEN_HIP_OUTPUT_RESIDUAL: return 2;
	default: return -1;
	}
}

bool output_computed(const zen_hip_hpr* e, int o)
{
	if (o == 0)
		return e->out_p;
	if (o == 1)
		return e->out_h;
	// hps.cu:562 : residual only with hard masks; the SSE path has no residual branch (hps.cu:582-652)
	return e->out_r && !e->soft && !e->use_sse;
}

void free_all(zen_hip_hpr* e)
{
	if (e->res_ctl) {
		(void)zen_hip_host_free(e->res_ctl);
		(void)zh_host_free(e->res_out);
		(void)hipStreamDestroy(e->res_stream);
		(void)hipEventDestroy(e->res_event);
		e->res_ctl = nullptr;
	}
	(void)zh_free(e->hstage_in);
	for (int o = 0; o < 3; ++o)
		(void)zh_free(e->hstage_out[o]);
	if (e->hs_in)
		(void)hipStreamDestroy(e->hs_in);
	if (e->hs_out)
		(void)hipStreamDestroy(e->hs_out);
	for (hipEvent_t ev : e->hevents)
		(void)hipEventDestroy(ev);
	e->hevents.clear();
	(void)zh_free(e->d_window);
	(void)zh_free(e->d_tw);
	(void)zh_free(e->d_tail[0]);
	(void)zh_free(e->d_tail[1]);
	(void)zh_free(e->d_S);
	(void)zh_free(e->d_mag);
	(void)zh_free(e->d_H);
	(void)zh_free(e->d_P);
	(void)zh_free(e->d_bits);
	(void)zh_free(e->d_bits_t);
	(void)zh_free(e->d_Mh);
	(void)zh_free(e->d_blk_flag);
	(void)zh_free(e->d_blk_need);
	(void)zh_free(e->d_run_sink);
	for (int o = 0; o < 3; ++o) {
		(void)zh_free(e->d_run_carry[0][o]);
		(void)zh_free(e->d_run_carry[1][o]);
	}
	for (int o = 0; o < 3; ++o) {
		(void)zh_free(e->d_Y[o]);
		(void)zh_free(e->d_carry[o]);
		if (e->ready_host[o])
			(void)zh_host_free(e->ready_host[o]);
		else
			(void)zh_free(e->ready_dev[o]);
	}
	if (e->dbg_stamps_host)
		(void)zh_host_free(e->dbg_stamps_host);
	if (e->d_wide_xch)
		(void)zh_free(e->d_wide_xch);
	if (e->d_wide_bar)
		(void)zh_free(e->d_wide_bar);
	if (e->wide_fail_host)
		(void)zh_host_free(e->wide_fail_host);
	for (auto& p : e->prof_pending) {
		(void)hipEventDestroy(p.e0);
		(void)hipEventDestroy(p.e1);
	}
	for (auto& p : e->prof_pool) {
		(void)hipEventDestroy(p.first);
		(void)hipEventDestroy(p.second);
	}
}

// Up to eight pitched regions zeroed by one launch (reset_state: seven memsets of a few KB to MB each were 0.1 ms of
// launch gaps per offline step, two engines reset per call).  Everything in 16-byte units: the engine's buffers are.
struct ZeroJobs {
	void* p[12];
	long long pitch16[12], width16[12]; // in 16-byte units
	int rows[12];
	int n;
};
__global__ __launch_bounds__(256) void zero_regions_kernel(ZeroJobs z)
{
	const int j = blockIdx.y;
	uint4* base = reinterpret_cast<uint4*>(z.p[j]);
	const long long w = z.width16[j], total = w * z.rows[j];
	for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
		const long long r = i / w, c = i - r * w;
		base[r * z.pitch16[j] + c] = make_uint4(0u, 0u, 0u, 0u);
	}
}
bool add_zero_job(ZeroJobs& z, void* p, size_t pitch_bytes, size_t width_bytes, size_t rows)
{
	if (!p || width_bytes == 0 || rows == 0)
		return true;
	if (z.n >= 12 || ((reinterpret_cast<uintptr_t>(p) | pitch_bytes | width_bytes) & 15) != 0 || rows > 0x7fffffff)
		return false;
	z.p[z.n] = p;
	z.pitch16[z.n] = (long long)(pitch_bytes >> 4);
	z.width16[z.n] = (long long)(width_bytes >> 4);
	z.rows[z.n] = (int)rows;
	++z.n;
	return true;
}

int reset_state(zen_hip_hpr* e)
{
	const size_t S = e->n_streams;
	// Only the W-1 history rows of the rings (absolute rows 0..W-2 of every stream) are read before they are
	// written; every other ring row, and every Y row, is produced by the call that consumes it.
	const size_t srow = sizeof(float2) * e->s_stride, mrow = sizeof(float) * e->nfft, hopb = sizeof(float) * S * e->hop;
	ZeroJobs z;
	memset(&z, 0, sizeof(z));
	bool ok = add_zero_job(z, e->d_tail[0], hopb, hopb, 1) && add_zero_job(z, e->d_tail[1], hopb, hopb, 1);
	if (e->W > 1)
		ok = ok && add_zero_job(z, e->d_S, srow * e->ring_rows, srow * (e->W - 1), S)
		     && add_zero_job(z, e->d_mag, mrow * e->ring_rows, mrow * (e->W - 1), S);
	for (int o = 0; o < 3; ++o)
		ok = ok && add_zero_job(z, e->d_carry[o], hopb, hopb, 1);
	for (int o = 0; o < 3; ++o) // (synthesis in runs: the carry a fresh stream's first run starts from)
		ok = ok && add_zero_job(z, e->d_run_carry[e->run_sel][o], hopb, hopb, 1);
	if (ok && z.n > 0) {
		hipLaunchKernelGGL(zero_regions_kernel, dim3(64, (unsigned)z.n), dim3(256), 0, e->stream, z);
		ZH_HIP(hipGetLastError());
	}
	else if (!ok) { // (a hop that is not a multiple of four samples: the memsets)
		ZH_HIP(hipMemsetAsync(e->d_tail[0], 0, hopb, e->stream));
		ZH_HIP(hipMemsetAsync(e->d_tail[1], 0, hopb, e->stream));
		if (e->W > 1) {
			ZH_HIP(hipMemset2DAsync(e->d_S, srow * e->ring_rows, 0, srow * (e->W - 1), S, e->stream));
			ZH_HIP(hipMemset2DAsync(e->d_mag, mrow * e->ring_rows, 0, mrow * (e->W - 1), S, e->stream));
		}
		for (int o = 0; o < 3; ++o)
			if (e->d_carry[o])
				ZH_HIP(hipMemsetAsync(e->d_carry[o], 0, hopb, e->stream));
		for (int o = 0; o < 3; ++o)
			if (e->d_run_carry[e->run_sel][o])
				ZH_HIP(hipMemsetAsync(e->d_run_carry[e->run_sel][o], 0, hopb, e->stream));
	}
	e->tail_sel = 0;
	e->rows_stale = false;
	e->abs_frame = (long long)e->W - 1; // rows 0..W-2 are the all-zero history of a fresh stream
	e->last_frames = 0;
	e->drain[0] = e->drain[1] = e->drain[2] = 0;
	e->ready_valid[0] = e->ready_valid[1] = e->ready_valid[2] = false;
	if (e->d_wide_bar) { // a time-out mark of rt_wide.hip must not outlive the stream it happened in
		ZH_HIP(hipStreamSynchronize(e->stream));
		ZH_HIP(hipMemset2DAsync(e->d_wide_bar + 1, sizeof(unsigned) * 4, 0, sizeof(unsigned), S, e->stream));
		*e->wide_fail_host = 0;
	}
	return ZEN_HIP_OK;
}

// Buffers only some paths touch are allocated when one of them first runs: the harmonic / percussive estimates
// of the three-kernel path (the fused causal kernel keeps them on chip) and the synthesis rows of an output
// that is not computed.  Y rows are zeroed once so that no kernel can see uninitialised memory.
int ensure_estimates(zen_hip_hpr* e, bool need_h)
{
	const size_t bytes = sizeof(float) * e->n_streams * e->max_hops * e->nfft;
	if (!e->d_P) { // zeroed like the Y rows: the half-row filters leave the bins nobody reads untouched
		ZH_HIP(zh_malloc((void**)&e->d_P, bytes));
		ZH_HIP(hipMemsetAsync(e->d_P, 0, bytes, e->stream));
	}
	if (need_h && !e->d_H) {
		ZH_HIP(zh_malloc((void**)&e->d_H, bytes));
		ZH_HIP(hipMemsetAsync(e->d_H, 0, bytes, e->stream));
	}
	return ZEN_HIP_OK;
}

int ensure_rows(zen_hip_hpr* e, int o)
{
	if (e->d_Y[o])
		return ZEN_HIP_OK;
	const size_t bytes = sizeof(float) * e->n_streams * e->max_hops * e->nwin;
	ZH_HIP(zh_malloc((void**)&e->d_Y[o], bytes));
	ZH_HIP(hipMemsetAsync(e->d_Y[o], 0, bytes, e->stream));
	return ZEN_HIP_OK;
}

// causal, median path: M hops of every stream in one launch, one workgroup per hop (rt_fused.hip)
enum HopKernel { HOP_FUSED = 0, HOP_SSE = 1, HOP_WIDE = 2 };

// the arguments of a fused causal launch of M hops per stream (side effects: the call's sequence number, ready_valid)
void fused_args(zen_hip_hpr* e, const float* in, size_t in_stride, size_t M, RtFusedArgs& a)
{
	memset(&a, 0, sizeof(a));
	a.in = in;
	a.in_stride = (long long)in_stride;
	a.tail_prev = e->d_tail[e->tail_sel];
	a.tail_next = e->d_tail[e->tail_sel ^ 1];
	a.window = e->d_window;
	a.tw = e->d_tw;
	a.S = M == 1 ? e->d_S : nullptr; // causal: no later frame reads the rings (the time median is the identity)
	a.s_stride = (long long)e->s_stride;
	a.mag = e->d_mag;
	a.keep_mag_rows = (int)e->W - 1;
	a.ring_rows = e->ring_rows;
	a.row0 = e->abs_frame;
	a.hop = (int)e->hop;
	a.n_frames = (int)M;
	a.n_streams = (int)e->n_streams;
	a.prev_frames = (int)e->last_frames;
	a.y_stream_stride = (long long)(e->max_hops * e->nwin);
	for (int o = 0; o < 3; ++o) {
		a.Y[o] = e->d_Y[o];
		e->ready_valid[o] = false;
		if (output_computed(e, o)) {
			a.carry[o] = e->d_carry[o];
			a.out_id[a.n_out++] = o;
			a.ready[o] = e->ready_dev[o];
			e->ready_valid[o] = (M == 1);
		}
	}
	if (M == 1)
		a.seq = ++e->hop_seq;
	a.publish_seq = e->ready_host[0] != nullptr ? (g_opt_publish_release ? 2 : 1) : 0; // rt_fused.hip publish_ready
	a.beta = e->beta;
	a.beta_h = e->beta - FLT_EPSILON;
	a.cola = e->cola;
	a.soft = e->soft ? 1 : 0;
	a.power = (int)e->beta;
	a.out_h = e->out_h ? 1 : 0;
	a.out_p = e->out_p ? 1 : 0;
	a.thr = ZH_DIAG_OPT(g_opt_mask_divide) ? 0.0 : hard_mask_threshold(e->beta, &a.thr_inclusive);
	a.thr_h = ZH_DIAG_OPT(g_opt_mask_divide) ? 0.0 : hard_mask_threshold(a.beta_h, &a.thr_h_inclusive);
	a.diag = ZH_DIAG_OPT(g_opt_rt_fused_diag);
	a.stamps = e->dbg_stamps;
}

// what a cooperative single-hop launch (rt_wide.hip) needs on top of fused_args, and the engine's account of the stream's
// barrier word: every call adds rt_wide_arrivals to it and takes the other of the two placement-vote words.  A resident
// launch votes once and leaves its word full: the words are cleared (on `stream`, which is ordered behind the kernel that
// left) before the next launch of either kind.
int wide_args(zen_hip_hpr* e, RtFusedArgs& a, hipStream_t stream, bool clear_votes = true)
{
	ZH_TRY(ensure_estimates(e, false));
	if (!e->d_wide_xch) {
		ZH_HIP(zh_malloc((void**)&e->d_wide_xch, sizeof(float2) * e->n_streams * e->nfft));
		ZH_HIP(zh_malloc((void**)&e->d_wide_bar, sizeof(unsigned) * 4 * e->n_streams + 16)); // (+ the resident launch's `go` word)
		ZH_HIP(hipMemsetAsync(e->d_wide_bar, 0, sizeof(unsigned) * 4 * e->n_streams + 16, e->stream));
		e->wide_arrivals = 0;
		e->wide_calls = 0;
		void* dev = nullptr;
		ZH_HIP(zh_host_malloc((void**)&e->wide_fail_host, 64, hipHostMallocMapped | hipHostMallocPortable));
		*e->wide_fail_host = 0;
		ZH_HIP(zh_host_device_pointer(&dev, e->wide_fail_host));
		e->wide_fail_dev = (unsigned*)dev;
	}
	if (e->wide_votes_stale && clear_votes) {
		for (size_t st = 0; st < e->n_streams; ++st)
			ZH_HIP(hipMemsetAsync(e->d_wide_bar + 4 * st + 2, 0, 2 * sizeof(unsigned), stream));
		e->wide_votes_stale = false;
	}
	a.wide_fail = e->wide_fail_dev;
	a.P = e->d_P;
	a.p_stream_stride = (long long)(e->max_hops * e->nfft);
	a.xch = e->d_wide_xch;
	a.bar = e->d_wide_bar;
	a.bar_base = e->wide_arrivals;
	a.bar_parity = (int)(e->wide_calls++ & 1u);
	e->wide_arrivals += rt_wide_arrivals(e->log2n, a.n_out);
	return ZEN_HIP_OK;
}

int run_hop_fused(zen_hip_hpr* e, const float* in, size_t in_stride, size_t M, HopKernel kind = HOP_FUSED)
{
	const bool sse = kind == HOP_SSE;
	for (int o = 0; o < 3; ++o)
		if (output_computed(e, o))
			ZH_TRY(ensure_rows(e, o));
	RtFusedArgs a;
	fused_args(e, in, in_stride, M, a);
	if (kind == HOP_WIDE)
		ZH_TRY(wide_args(e, a, e->stream));
	// block calls of the headline configuration: the kernel finishes the hops itself where the caller has said where they go
	int direct_o = -1;
	for (int o = 0; o < 3; ++o)
		e->direct_done[o] = false;
	bool all_dst = a.n_out > 0;
	for (int i = 0; i < a.n_out; ++i)
		all_dst = all_dst && e->direct_out[a.out_id[i]] != nullptr;
	if (kind == HOP_FUSED && all_dst && !g_opt_no_direct_out && rt_fused_direct_out_available(e->log2n, e->mf, a)) {
		direct_o = a.out_id[0];
		if (!e->d_blk_flag) {
			const size_t bytes = sizeof(unsigned) * e->n_streams * e->max_hops;
			ZH_HIP(zh_malloc((void**)&e->d_blk_flag, bytes));
			ZH_HIP(zh_malloc((void**)&e->d_blk_need, bytes));
			ZH_HIP(hipMemsetAsync(e->d_blk_flag, 0, bytes, e->stream));
			ZH_HIP(hipMemsetAsync(e->d_blk_need, 0, bytes, e->stream));
			e->blk_seq = 0;
		}
		for (int o = 0; o < 3; ++o)
			a.out_direct[o] = e->direct_out[o];
		a.out_direct_stride = e->direct_stride;
		a.direct_on = 1;
		a.blk_flag = e->d_blk_flag;
		a.blk_need = e->d_blk_need;
		e->blk_seq = (e->blk_seq + 1) & 0x0fffffffu;
		if (e->blk_seq == 0) // (the words start at zero: sequence number zero is never used)
			e->blk_seq = 1;
		a.blk_seq = e->blk_seq;
	}
	{
		ProfScope ps(e, zen_hip_hpr::K_FUSED);
		if (kind == HOP_WIDE)
			ZH_TRY(launch_rt_wide(e->log2n, e->mf, a, e->stream));
		else if (sse) // single hop of the SSE path (rt_sse.hip); box lengths and factors as launch_box gets them below
			ZH_TRY(launch_rt_sse(e->log2n, a, e->mt, e->mf, (float)e->l_harm + 1.0F, (float)e->l_perc + 1.0F, e->stream));
		else
			ZH_TRY(launch_rt_fused(e->log2n, e->mf, a, e->stream));
	}
	if (direct_o >= 0) {
		ProfScope ps(e, zen_hip_hpr::K_FINALIZE);
		ZH_TRY(launch_rt_fused_fixup(a, e->stream));
		for (int i = 0; i < a.n_out; ++i)
			e->direct_done[a.out_id[i]] = true;
	}
	e->tail_sel ^= 1;
	e->abs_frame += (long long)M;
	e->last_frames = M;
	return ZEN_HIP_OK;
}

// An output switched off in mid-stream (see zen_hip_hpr::drain): the reference rotates its accumulator once per hop
// all the same (hps.cu:435-449), so the call after the switch hands out the second half of the last frame that was
// still added to it, and the call after that zeros.
int advance_drain(zen_hip_hpr* e)
{
	for (int o = 0; o < 3; ++o) {
		if (e->drain[o] == 2) {
			e->drain[o] = 0;
		}
		else if (e->drain[o] == 1) {
			const float* y = e->d_Y[o] + (e->last_frames - 1) * e->nwin + e->hop;
			ZH_HIP(hipMemcpy2DAsync(e->d_carry[o], sizeof(float) * e->hop, y, sizeof(float) * e->max_hops * e->nwin,
			                        sizeof(float) * e->hop, e->n_streams, hipMemcpyDeviceToDevice, e->stream));
			e->drain[o] = 2;
		}
	}
	return ZEN_HIP_OK;
}

// ---- resident single-hop kernel (zen_hip_hpr_set_resident; rt_resident.hip) --------------------------------------
// Host side of the mailbox protocol.  Posting a hop: its arguments are remembered (res_args), the kernel is launched if
// none is there, the input pointer and then the sequence word go to the mailbox (store fences in between: the mailbox
// and the caller's input buffer may be write-combined device memory behind the BAR).  A kernel that has left -- idle
// for res_idle_ms, or the hop posted in the window between its last look and its exit word -- is noticed by whoever
// waits for a hop (resident_kick) and launched again with the remembered arguments: no hop is lost, none runs twice
// (the exit word carries the last sequence number the kernel processed).
inline void store_fence() { __builtin_ia32_sfence(); }

// hops 2048 / 4096 on the median path: the cooperative kernel of rt_wide.hip (where run_chunk would pick it)
bool resident_wide(const zen_hip_hpr* e)
{
	return !e->use_sse && !rt_fused_available(e->log2n, e->mf) && e->n_streams <= 8 && rt_wide_available(e->log2n, e->mf);
}

bool resident_eligible(const zen_hip_hpr* e)
{
	int n_out = 0, o1 = -1;
	for (int o = 0; o < 3; ++o)
		if (output_computed(e, o)) {
			++n_out;
			o1 = o;
		}
	const bool kernel = e->use_sse ? rt_sse_available(e->log2n, e->mt, e->mf)
	                               : (rt_fused_available(e->log2n, e->mf) || resident_wide(e));
	return e->res_idle_ms > 0 && e->causality == ZEN_HIP_TIME_CAUSAL && !g_opt_no_rt_fused && e->n_streams == 1 && n_out == 1
	       && e->ready_host[o1] != nullptr && kernel && !e->prof && !(e->dbg_stamps && resident_wide(e)) && !(e->drain[0] | e->drain[1] | e->drain[2]);
}

int resident_launch(zen_hip_hpr* e) // from res_args: the hop whose number is res_args.seq is the first the kernel will see
{
	if (!e->res_ctl) {
		void *h = nullptr, *d = nullptr;
		ZH_TRY(zen_hip_host_alloc_mapped(sizeof(ResidentCtl), 1, &h, &d)); // device memory behind the BAR where there is one
		e->res_ctl = (ResidentCtl*)h;
		e->res_ctl_dev = (ResidentCtl*)d;
		ZH_HIP(zh_host_malloc((void**)&e->res_out, sizeof(ResidentOut), hipHostMallocMapped | hipHostMallocPortable));
		ZH_HIP(zh_host_device_pointer(&d, e->res_out));
		e->res_out_dev = (ResidentOut*)d;
		ZH_HIP(hipStreamCreateWithFlags(&e->res_stream, hipStreamNonBlocking));
		ZH_HIP(hipEventCreateWithFlags(&e->res_event, hipEventDisableTiming));
		e->res_ctl->seq = 0;
		e->res_ctl->stop = 0;
		e->res_ctl->pad = 0;
	}
	const unsigned seq_start = e->res_args.seq - 1u;
	if (e->res_ctl->seq != e->res_args.seq) // (a relaunch finds the pending hop's number already there)
		e->res_ctl->seq = seq_start;
	e->res_ctl->stop = 0;
	store_fence();
	__atomic_store_n(&e->res_out->exited, 0u, __ATOMIC_RELEASE);
	// behind whatever the engine's own stream still has in flight (an earlier per-launch hop, a reset)
	ZH_HIP(hipEventRecord(e->res_event, e->stream));
	ZH_HIP(hipStreamWaitEvent(e->res_stream, e->res_event, 0));
	const unsigned long long ticks = (unsigned long long)e->res_idle_ms * 100000ull; // s_memrealtime: 100 MHz
	if (resident_wide(e)) { // the cooperating workgroups agree through a word behind the barrier words (wide_args); zeroed per launch
		unsigned long long* go = reinterpret_cast<unsigned long long*>(e->d_wide_bar + 4 * e->n_streams);
		ZH_HIP(hipMemsetAsync(go, 0, sizeof(unsigned long long), e->res_stream));
		ZH_HIP(hipMemsetAsync(e->d_wide_bar + 2, 0, 2 * sizeof(unsigned), e->res_stream)); // both vote words (one stream: resident_eligible)
		ZH_TRY(launch_rt_wide_resident(e->log2n, e->mf, e->res_args, e->res_ctl_dev, e->res_out_dev, go, seq_start, ticks, 0x7fffffffu,
		                               e->res_stream));
		e->wide_votes_stale = true; // (it votes once and never clears its word)
	}
	else if (e->use_sse) // the single-launch SSE kernel's body (rt_sse.hip); box lengths and factors as run_hop_fused passes them
		ZH_TRY(launch_rt_sse_resident(e->log2n, e->res_args, e->mt, e->mf, (float)e->l_harm + 1.0F, (float)e->l_perc + 1.0F, e->res_ctl_dev,
		                              e->res_out_dev, seq_start, ticks, 0x7fffffffu, e->res_stream));
	else
		ZH_TRY(launch_rt_fused_resident(e->log2n, e->mf, e->res_args, e->res_ctl_dev, e->res_out_dev, seq_start, ticks, 0x7fffffffu,
		                                e->res_stream));
	e->res_active = true;
	++e->res_launches;
	return ZEN_HIP_OK;
}

// has the kernel left?  If so, and the hop posted last was not processed, launch it again (with that hop pending).
int resident_kick(zen_hip_hpr* e)
{
	if (!e->res_active || !__atomic_load_n(&e->res_out->exited, __ATOMIC_ACQUIRE))
		return ZEN_HIP_OK;
	e->res_active = false;
	e->res_hops += e->res_out->hops;
	if (e->res_out->last_seq != e->res_args.seq)
		ZH_TRY(resident_launch(e));
	return ZEN_HIP_OK;
}

// wait until the hop posted last is in its host-mapped buffer (the sequence word behind it)
int resident_wait(zen_hip_hpr* e)
{
	if (!e->res_active)
		return ZEN_HIP_OK;
	const int o = e->res_args.out_id[0];
	const unsigned* flag = reinterpret_cast<const unsigned*>(e->ready_host[o] + e->hop);
	for (long spin = 0; spin < 2000000000L; ++spin) {
		if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == e->res_args.seq)
			return ZEN_HIP_OK;
		if ((spin & 63) == 63)
			ZH_TRY(resident_kick(e));
		__builtin_ia32_pause();
	}
	ZH_FAIL(ZEN_HIP_E_HIP, "resident kernel: hop %u never arrived", e->res_args.seq);
}

// every entry point that touches the engine in any other way first sends the kernel home
int resident_stop(zen_hip_hpr* e)
{
	if (!e->res_ctl)
		return ZEN_HIP_OK;
	if (e->res_active) {
		ZH_TRY(resident_wait(e)); // (the hop in flight, if any, is finished by whoever holds it)
		e->res_ctl->stop = 1;
		store_fence();
		ZH_HIP(hipStreamSynchronize(e->res_stream));
		e->res_hops += e->res_out->hops;
		e->res_active = false;
		e->res_ctl->stop = 0;
		store_fence();
	}
	return ZEN_HIP_OK;
}

int resident_post(zen_hip_hpr* e, const float* in)
{
	for (int o = 0; o < 3; ++o)
		if (output_computed(e, o))
			ZH_TRY(ensure_rows(e, o));
	ZH_TRY(resident_wait(e)); // the mailbox holds one hop: the one before must have been taken (and finished)
	if (e->res_active && e->res_args.in != in) // the kernel reads the input buffer it was launched with (IOGPU::device_in,
		ZH_TRY(resident_stop(e));              // normally the same for every hop): another buffer, another launch
	fused_args(e, in, e->hop, 1, e->res_args);
	e->res_args.stamps = e->dbg_stamps; // (diagnostic: the phase stamps of every resident hop, tools/rt_latency.cpp)
	if (resident_wide(e)) // (the engine's account of the barrier word advances hop by hop, as the kernel's does: a relaunch, or a
		ZH_TRY(wide_args(e, e->res_args, e->stream, /*clear_votes=*/false)); // per-launch hop later, starts from the right count)
	ZH_TRY(resident_kick(e));
	if (!e->res_active)
		ZH_TRY(resident_launch(e));
	store_fence(); // the caller's samples (write-combined stores, possibly) before the word
	__atomic_store_n(&e->res_ctl->seq, e->res_args.seq, __ATOMIC_RELEASE);
	store_fence();
	e->tail_sel ^= 1;
	e->abs_frame += 1;
	e->last_frames = 1;
	return ZEN_HIP_OK;
}

int pick_wide_run(zen_hip_hpr* e, size_t M, size_t S, const int* group_outputs, int n_groups, int log2n, double min_eff); // below

// may_post: the call is zen_hip_hpr_process_next_hop -- the only entry point whose result is collected through copy_* /
// resident_wait.  A block call of one hop (zen_hip_hpr_process, n_hops == 1) queues its overlap-add on the engine's own
// stream right behind this: it must not hand the hop to a kernel that works asynchronously on another stream.
int run_chunk(zen_hip_hpr* e, const float* in, size_t in_stride, size_t M, long long in_valid = LLONG_MAX, bool may_post = false)
{
	const size_t S = e->n_streams, N = e->nfft;
	if (may_post && M == 1 && in_stride == e->hop && resident_eligible(e))
		return resident_post(e, in);
	ZH_TRY(resident_stop(e));
	if (e->rows_stale && !e->run_mode)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr: the stream's last pass kept no synthesis rows (offline driver); zen_hip_hpr_reset_buffers first");
	if (e->drain[0] | e->drain[1] | e->drain[2])
		ZH_TRY(advance_drain(e));
	if (e->causality == ZEN_HIP_TIME_CAUSAL && !e->use_sse && !g_opt_no_rt_fused && (M == 1 || !g_opt_no_block_fused)
	    && rt_fused_available(e->log2n, e->mf))
		return run_hop_fused(e, in, in_stride, M);
	if (e->causality == ZEN_HIP_TIME_CAUSAL && e->use_sse && M == 1 && !g_opt_no_rt_fused
	    && rt_sse_available(e->log2n, e->mt, e->mf))
		return run_hop_fused(e, in, in_stride, 1, HOP_SSE);
	// (the cooperative kernel keeps its workgroups on one XCD -- 32 CUs: right for the latency of a few streams,
	// wrong for the throughput of many, which the general engine spreads over the whole device)
	if (e->causality == ZEN_HIP_TIME_CAUSAL && !e->use_sse && M == 1 && !g_opt_no_rt_fused && e->n_streams <= 8
	    && rt_wide_available(e->log2n, e->mf))
		return run_hop_fused(e, in, in_stride, 1, HOP_WIDE);
	// Half rows: |S| is exactly Hermitian, so the median path stores and filters bins 0..nfft/2 only (and the
	// last mf/2 bins of P, whose replicate border differs): half the magnitude / H / P traffic, half the
	// median work.  Not for the SSE box mean (its ascending summation is not mirror symmetric), nor where a
	// kernel on the way does not know such rows.  The last W-1 frames of a chunk keep whole magnitude rows: a
	// later use_sse_filter() (hps.h:289) reads them.
	const bool time_identity = (e->causality == ZEN_HIP_TIME_CAUSAL || e->mt == 1);
	// SSE path, blocks of frames: both box filters, the masks and the synthesis in one launch behind the analysis
	// (sse_block.hip); no H / P rows at all
	int n_computed = 0;
	for (int o = 0; o < 3; ++o)
		n_computed += output_computed(e, o) ? 1 : 0;
	const bool sse_block = e->use_sse && M >= 2 && sse_block_available(e->log2n, e->mt, e->mf, n_computed, e->ring_rows);
	if (!sse_block)
		ZH_TRY(ensure_estimates(e, e->use_sse || !time_identity));
	for (int o = 0; o < 3; ++o)
		if (output_computed(e, o) && !e->run_mode) // (synthesis in runs: no Y rows)
			ZH_TRY(ensure_rows(e, o));
	const bool half = !e->use_sse && !g_opt_no_half_rows && !g_opt_median_general
	                  && filter_supports_hermitian(e->mf, (int)N) && (time_identity || e->mt <= 63);
	// ---- analysis
	StftArgs sa;
	memset(&sa, 0, sizeof(sa));
	sa.in = in;
	sa.in_stride = (long long)in_stride;
	sa.in_valid = in_valid;
	sa.tail_prev = e->d_tail[e->tail_sel];
	sa.tail_next = e->d_tail[e->tail_sel ^ 1];
	sa.window = e->d_window;
	sa.tw = e->d_tw;
	sa.S = e->d_S;
	sa.s_stride = (long long)e->s_stride;
	sa.mag = e->d_mag;
	sa.mag_full_from = (half || sse_block) ? (int)M - ((int)e->W - 1) : 0; // (sse_block reads bins 0..nfft/2 only)
	sa.ring_rows = e->ring_rows;
	sa.row0 = e->abs_frame;
	sa.n_frames = (int)M;
	sa.hop = (int)e->hop;
	sa.n_streams = (int)S;
	sa.prev_frames = (int)e->last_frames;
	for (int o = 0; o < 3; ++o) {
		sa.carry[o] = (output_computed(e, o) && !e->run_mode) ? e->d_carry[o] : nullptr;
		sa.Y[o] = e->d_Y[o];
	}
	sa.y_stream_stride = (long long)(e->max_hops * e->nwin);
	{
		ProfScope ps(e, zen_hip_hpr::K_STFT);
		ZH_TRY(launch_stft(e->log2n, sa, e->stream));
	}
	e->tail_sel ^= 1;

	// ---- harmonic / percussive estimates of the consumed rows (row W-lag of the sliding matrix)
	const long long crow0 = e->abs_frame - (e->lag - 1);
	if (sse_block) {
		SseBlockArgs sb;
		memset(&sb, 0, sizeof(sb));
		IstftArgs& ia = sb.ia;
		ia.S = e->d_S;
		ia.s_stride = (long long)e->s_stride;
		ia.ring_rows = e->ring_rows;
		ia.crow0 = crow0;
		ia.tw = e->d_tw;
		ia.y_stream_stride = (long long)(e->max_hops * e->nwin);
		ia.n_frames = (int)M;
		ia.n_streams = (int)S;
		ia.hop = (int)e->hop;
		for (int o = 0; o < 3; ++o) {
			e->ready_valid[o] = false;
			if (output_computed(e, o)) {
				ia.Y[ia.n_out] = e->d_Y[o];
				ia.out_id[ia.n_out] = o;
				++ia.n_out;
			}
		}
		ia.beta = e->beta;
		ia.beta_h = e->beta - FLT_EPSILON;
		ia.soft = e->soft ? 1 : 0;
		ia.power = (int)e->beta;
		ia.sse = 1;
		ia.out_h = e->out_h ? 1 : 0;
		ia.out_p = e->out_p ? 1 : 0;
		ia.cola = e->cola;
		sb.mag = e->d_mag;
		sb.clamp_lo = 0; // as the FilterArgs of the box launches below
		sb.clamp_hi = LLONG_MAX / 4;
		sb.len_t = e->mt;
		sb.len_f = e->mf;
		sb.causal_self = (e->causality == ZEN_HIP_TIME_CAUSAL) ? 1 : 0;
		sb.fac_h = (float)e->l_harm + 1.0F; // hps.cu:602-604
		sb.fac_p = (float)e->l_perc + 1.0F; // hps.cu:599-601
		{
			ProfScope ps(e, zen_hip_hpr::K_ISTFT);
			ZH_TRY(launch_sse_block(e->log2n, sb, e->stream));
		}
		e->abs_frame += (long long)M;
		e->last_frames = M;
		return ZEN_HIP_OK;
	}
	FilterArgs fa;
	memset(&fa, 0, sizeof(fa));
	fa.src = e->d_mag;
	fa.src_stream_stride = e->ring_rows * (long long)N;
	fa.dst_stream_stride = (long long)(e->max_hops * N);
	fa.n_streams = (int)S;
	fa.cols = (int)N;
	fa.ring_rows = e->ring_rows;
	fa.first_row = crow0;
	fa.n_out_rows = (int)M;
	fa.clamp_lo = 0;
	fa.clamp_hi = LLONG_MAX / 4;
	fa.nonneg = 1; // |S| >= +0

	// Blocks of frames with hard masks: the comparisons are made once per bin -- by the frequency-direction median kernel
	// itself where it knows how (FilterArgs::bits: two bits per bin instead of the P row), else in a launch of their own
	// -- and the synthesis loads two bits per bin instead of H and P (per output, and per mirror image).
	const float beta_h = e->beta - FLT_EPSILON; // hps.cu:540 hard_mask_functor(beta - Eps)
	const HardThr thr = hard_mask_thresholds(e->beta, beta_h, ZH_DIAG_OPT(g_opt_mask_divide) != 0);
	const int p_mid = e->use_sse ? (int)(N / 2) : e->mf / 2;
	const bool use_bits = M >= 8 && mask_bits_supported((int)N, p_mid) && !e->soft && !e->use_sse && thr.p != 0.0 && thr.h != 0.0
	                      && !g_opt_no_mask_bits;
	const int bits_row_words = mask_bits_row_words((int)N, p_mid);
	const long long bits_stream_stride = (long long)e->max_hops * bits_row_words;
	const long long bits_t_stream_stride = (long long)e->max_hops * (long long)(N / 16);
	if (use_bits && !e->d_bits) {
		ZH_HIP(zh_malloc((void**)&e->d_bits, sizeof(unsigned) * S * (size_t)bits_stream_stride));
		ZH_HIP(zh_malloc((void**)&e->d_bits_t, sizeof(unsigned) * S * (size_t)bits_t_stream_stride));
	}
	int bits_done = 0; // 1: the frequency-direction kernel wrote IstftArgs::bits, 2: ::bits_t
	// the masks an enabled output reads: its own, and for the residual those of hps.cu:562-567
	const int need_pm = (output_computed(e, 0) || (output_computed(e, 2) && e->out_p)) ? 1 : 0;
	const int need_hm = (output_computed(e, 1) || (output_computed(e, 2) && e->out_h)) ? 1 : 0;

	bool h_is_ring = false; // time direction -> H   (hps.cu:495 / :596)
	bool fuse_tf = false;
	FilterArgs ft = fa;
	ft.dst = e->d_H;
	ft.len = e->mt;
	ft.direction = e->causality;
	if (half) { // only the stored half of every row (rounded up to whole 16-byte vectors)
		ft.cols = (int)N / 2 + 4;
		ft.pitch = (int)N;
	}
	if (e->use_sse) {
		ft.sse_pre = 1;
		ft.sse_post = 1;
		ft.post_factor = (float)e->l_harm + 1.0F; // hps.cu:602-604
		ft.causal_self = (e->causality == ZEN_HIP_TIME_CAUSAL);
		ProfScope ps(e, zen_hip_hpr::K_TIME);
		ZH_TRY(launch_box(ft, e->stream));
	}
	else if (e->causality == ZEN_HIP_TIME_CAUSAL || e->mt == 1) {
		// Causal: the consumed row is the last row of the sliding matrix; the centred mask's upper half
		// replicates it, so mid+1 of 2*mid+1 taps equal the row itself and the median is the identity
		// (SURVEY Q1).  mt == 1: a one-tap median.  No launch.
		h_is_ring = true;
	}
	else if (use_bits && half && !g_opt_no_median_bits && median_tf_fused_available(e->mt, e->mf, (int)N)) {
		fuse_tf = true; // the frequency-direction launch below computes H itself (median_tf_herm_bits_kernel): no H rows
	}
	else {
		ProfScope ps(e, zen_hip_hpr::K_TIME);
		ZH_TRY(launch_median(ft, e->stream));
	}

	FilterArgs ff = fa; // frequency direction -> P   (hps.cu:496 / :597)
	ff.dst = e->d_P;
	ff.len = e->mf;
	ff.direction = ZEN_HIP_FREQUENCY;
	ff.hermitian = half ? 1 : 0;
	if (use_bits && half && !g_opt_no_median_bits) { // (H: the magnitude row itself, or what the time-direction launch above left)
		ff.bits = e->d_bits;
		ff.bits_stream_stride = bits_stream_stride;
		ff.bits_row_words = bits_row_words;
		ff.bits_t = e->d_bits_t;
		ff.bits_t_stream_stride = bits_t_stream_stride;
		ff.thr_p = thr.p;
		ff.thr_h = thr.h;
		ff.need_pm = need_pm;
		ff.need_hm = need_hm;
		if (fuse_tf) {
			ff.time_len = e->mt;
		}
		else if (!h_is_ring) {
			ff.hrows = e->d_H;
			ff.h_stream_stride = (long long)(e->max_hops * N);
		}
	}
	// Soft masks, blocks of frames: the median kernel that knows how leaves the masks themselves (the percussive one where
	// P would go, the harmonic one in d_Mh) and the synthesis loads one value per bin and output instead of H and P.
	// (Only where the harmonic estimate is the magnitude row itself -- the long-mask kernel of pass 1: in the sorting-network
	// kernel of pass 2 the extra H row, the divisions and the second row of stores cost 0.13 ms per offline-long step, twice
	// what the synthesis gained.)
	const bool use_soft_rows = M >= 8 && e->soft && !e->use_sse && half && h_is_ring && !g_opt_no_mask_bits && !g_opt_no_median_bits;
	if (use_soft_rows) {
		ff.soft_rows = 1;
		ff.soft_power = (int)e->beta; // hps.h:117-121: soft_mask_functor(int _power) truncates beta
		ff.need_pm = need_pm;
		ff.need_hm = need_hm;
		if (need_hm) {
			if (!e->d_Mh)
				ZH_HIP(zh_malloc((void**)&e->d_Mh, sizeof(float) * S * e->max_hops * N));
			ff.mh_dst = e->d_Mh;
			ff.mh_stream_stride = (long long)(e->max_hops * N);
		}
	}
	{
		ProfScope ps(e, zen_hip_hpr::K_FREQ);
		if (e->use_sse) {
			ff.sse_pre = 1;
			ff.sse_post = 1;
			ff.post_factor = (float)e->l_perc + 1.0F; // hps.cu:599-601
			ZH_TRY(launch_box(ff, e->stream));
		}
		else {
			ZH_TRY(launch_median(ff, e->stream, &bits_done));
		}
		if (e->prof) // elements the kernel is asked for: whole rows, or bins 0..nfft/2 and the last mf/2
			e->prof_elements += (unsigned long long)(M * (half ? N / 2 + 1 + (size_t)(e->mf / 2) : N) * S);
	}

	// ---- masks, inverse FFT, *COLA
	IstftArgs ia;
	memset(&ia, 0, sizeof(ia));
	ia.S = e->d_S;
	ia.s_stride = (long long)e->s_stride;
	ia.ring_rows = e->ring_rows;
	ia.crow0 = crow0;
	ia.H = h_is_ring ? e->d_mag : e->d_H;
	ia.h_stream_stride = (long long)(e->max_hops * N);
	ia.h_is_ring = h_is_ring ? 1 : 0;
	ia.P = e->d_P;
	ia.p_stream_stride = (long long)(e->max_hops * N);
	// The median is an order statistic: P[k] == P[nfft-k] bit for bit away from the borders.  The SSE box mean
	// adds its taps in ascending bin order, which the mirrored window reverses: rounding differs, no symmetry.
	ia.p_mid = p_mid;
	ia.tw = e->d_tw;
	ia.y_stream_stride = (long long)(e->max_hops * e->nwin);
	ia.n_frames = (int)M;
	ia.n_streams = (int)S;
	ia.n_out = 0;
	ia.hop = (int)e->hop;
	for (int o = 0; o < 3; ++o) {
		e->ready_valid[o] = false;
		if (output_computed(e, o)) {
			ia.Y[ia.n_out] = e->d_Y[o];
			ia.out_id[ia.n_out] = o;
			ia.ready[ia.n_out] = e->ready_dev[o];
			ia.carry[ia.n_out] = e->d_carry[o];
			e->ready_valid[o] = (M == 1);
			++ia.n_out;
		}
	}
	if (M == 1)
		ia.seq = ++e->hop_seq;
	ia.publish_seq = e->ready_host[0] != nullptr ? (g_opt_publish_release ? 2 : 1) : 0;
	ia.beta = e->beta;
	ia.beta_h = beta_h;
	ia.soft = e->soft ? 1 : 0;
	ia.power = (int)e->beta; // hps.h:117-121 : soft_mask_functor(int _power) truncates beta
	ia.sse = e->use_sse ? 1 : 0;
	ia.out_h = e->out_h ? 1 : 0;
	ia.out_p = e->out_p ? 1 : 0;
	ia.cola = e->cola;
	ia.thr_p = thr.p;
	ia.thr_h = thr.h;
	ia.thr_p_inc = thr.p_inc;
	ia.thr_h_inc = thr.h_inc;
	if (bits_done == 3) { // the frequency-direction launch left the soft masks
		ia.mask_rows = 1;
		ia.Hm = e->d_Mh;
	}
	if (use_bits) {
		ia.bits_row_words = bits_row_words;
		ia.bits_stream_stride = bits_stream_stride;
		ia.bits = e->d_bits;
		ia.bits_t_stream_stride = bits_t_stream_stride;
		ia.bits_t = e->d_bits_t;
		ia.need_pm = need_pm;
		ia.need_hm = need_hm;
		ProfScope ps(e, zen_hip_hpr::K_ISTFT);
		if (bits_done == 0)
			ZH_TRY(launch_mask_bits((int)N, ia, e->d_bits, e->stream));
		if (bits_done != 2)
			ZH_TRY(launch_mask_bits_transpose((int)N, ia, e->d_bits_t, e->stream));
	}
	if (e->run_mode) { // synthesis in runs, finished hops straight to the pass's destinations (hpr_process_spec)
		if (!use_bits || e->run_n_groups < 1)
			ZH_FAIL(ZEN_HIP_E_HIP, "hpr: synthesis in runs without mask bits (internal)");
		IstftRunArgs ra;
		memset(&ra, 0, sizeof(ra));
		ra.S = e->d_S;
		ra.s_stride = (long long)e->s_stride;
		ra.ring_rows = e->ring_rows;
		ra.crow0 = crow0;
		ra.tw = e->d_tw;
		ra.bits_t = e->d_bits_t;
		ra.bits_t_stream_stride = bits_t_stream_stride;
		ra.n_frames = (int)M;
		ra.n_streams = (int)S;
		ra.hop = (int)e->hop;
		ra.out_h = e->out_h ? 1 : 0;
		ra.out_p = e->out_p ? 1 : 0;
		ra.cola = e->cola;
		int run = (e->log2n <= 10 ? g_opt_istft_run : g_opt_istft_run_wide).load(std::memory_order_relaxed);
		if (run <= 0 && e->log2n > 10) { // (a chunk shorter than the pass's first: whatever fills the device best)
			int outs[3];
			for (int g = 0; g < e->run_n_groups; ++g)
				outs[g] = e->run_groups[g].n_out;
			run = pick_wide_run(e, M, S, outs, e->run_n_groups, e->log2n, 0.0);
		}
		ra.run = run > 0 ? run : 16;
		ra.pos0 = e->run_pos0;
		ra.n_groups = e->run_n_groups;
		ra.sink = e->d_run_sink;
		for (int g = 0; g < e->run_n_groups; ++g) {
			const zen_hip_hpr::RunGroup& rg = e->run_groups[g];
			IstftRunGroup& G = ra.g[g];
			G.n_out = rg.n_out;
			for (int k = 0; k < rg.n_out; ++k) {
				G.which[k] = rg.which[k];
				G.carry_prev[k] = e->d_run_carry[e->run_sel][rg.which[k]];
				G.carry_next[k] = e->d_run_carry[e->run_sel ^ 1][rg.which[k]];
			}
			G.out = rg.spec->dst;
			G.out_stride = rg.spec->stride;
			G.shift = rg.spec->shift;
			G.len = rg.spec->len;
			G.dup_from = rg.spec->dup_from;
			G.dup_shift = rg.spec->dup_shift;
			G.dup_len = rg.spec->dup_len;
		}
		e->run_sel ^= 1;
		ProfScope ps(e, zen_hip_hpr::K_ISTFT);
		ZH_TRY(launch_istft_run(e->log2n, ra, e->stream));
		e->rows_stale = true;
	}
	else {
		ProfScope ps(e, zen_hip_hpr::K_ISTFT);
		ZH_TRY(launch_istft(e->log2n, ia, e->stream));
	}

	e->abs_frame += (long long)M;
	e->last_frames = M;
	return ZEN_HIP_OK;
}

// Called when use_sse_filter / use_soft_mask switch an output off in mid-stream.  Nothing moves yet: until the next
// hop is processed copy_* keeps handing out the hop(s) finished before the switch (the Y rows and the carry are
// untouched); advance_drain() takes it from there.
int park_dropped_outputs(zen_hip_hpr* e, const bool (&before)[3])
{
	for (int o = 0; o < 3; ++o) {
		if (!before[o] || output_computed(e, o))
			continue;
		e->drain[o] = (e->last_frames > 0 && e->d_Y[o]) ? 1 : 0;
	}
	return ZEN_HIP_OK;
}

// copy_* of output o hands out what the kernels of the last call left for it
bool output_served(const zen_hip_hpr* e, int o) { return output_computed(e, o) || e->drain[o] == 1; }

int finalize_output(zen_hip_hpr* e, int o, float* out, size_t out_stride, size_t M)
{
	if (!output_served(e, o)) {
		// the reference's accumulator for a disabled output stays all zero (hps.test.cu:321-343) ...
		ZH_HIP(hipMemset2DAsync(out, sizeof(float) * out_stride, 0, sizeof(float) * M * e->hop, e->n_streams,
		                        e->stream));
		if (e->drain[o] == 2) // ... except, in the call after it was switched off, for the tail of its last frame
			ZH_HIP(hipMemcpy2DAsync(out, sizeof(float) * out_stride, e->d_carry[o], sizeof(float) * e->hop,
			                        sizeof(float) * e->hop, e->n_streams, hipMemcpyDeviceToDevice, e->stream));
		return ZEN_HIP_OK;
	}
	FinalizeArgs fa;
	fa.Y = e->d_Y[o];
	fa.carry = e->d_carry[o];
	fa.out = out;
	fa.y_stream_stride = (long long)(e->max_hops * e->nwin);
	fa.out_stride = (long long)out_stride;
	fa.n_frames = (int)M;
	fa.hop = (int)e->hop;
	fa.n_streams = (int)e->n_streams;
	ProfScope ps(e, zen_hip_hpr::K_FINALIZE);
	return launch_finalize(fa, e->stream);
}

// A default-sized engine (max_hops_per_chunk == 0 at create) grows to the block calls it is given.  What has to
// survive: the stft_width-1 history rows of the two rings (they move to their positions in the longer ring) and the
// Y rows of the last call (the next call's carry, and what copy_* hands out).  H and P are scratch.
int grow_buffers(zen_hip_hpr* e, size_t new_hops)
{
	const size_t S = e->n_streams, N = e->nfft, old_hops = e->max_hops;
	const long long old_rows = e->ring_rows, new_rows = (long long)(new_hops + e->W - 1);
	const size_t srow = sizeof(float2) * e->s_stride, mrow = sizeof(float) * N;
	ZH_HIP(hipStreamSynchronize(e->stream));
	float2* nS = nullptr;
	float* nmag = nullptr;
	float* nY[3] = {nullptr, nullptr, nullptr};
	bool ok = zh_malloc((void**)&nS, srow * S * new_rows) == hipSuccess && zh_malloc((void**)&nmag, mrow * S * new_rows) == hipSuccess;
	for (int o = 0; o < 3 && ok; ++o)
		if (e->d_Y[o])
			ok = zh_malloc((void**)&nY[o], sizeof(float) * S * new_hops * e->nwin) == hipSuccess;
	if (!ok) {
		(void)zh_free(nS);
		(void)zh_free(nmag);
		for (int o = 0; o < 3; ++o)
			(void)zh_free(nY[o]);
		(void)hipGetLastError();
		ZH_FAIL(ZEN_HIP_E_HIP, "hpr_process: growing the engine to %zu hops per chunk failed (nfft %zu, streams %zu)", new_hops, N, S);
	}
	ZH_HIP(hipMemsetAsync(nS, 0, srow * S * new_rows, e->stream));
	ZH_HIP(hipMemsetAsync(nmag, 0, mrow * S * new_rows, e->stream));
	for (long long r = e->abs_frame - ((long long)e->W - 1); r < e->abs_frame; ++r) {
		ZH_HIP(hipMemcpy2DAsync(nS + (r % new_rows) * e->s_stride, srow * new_rows, e->d_S + (r % old_rows) * e->s_stride,
		                        srow * old_rows, srow, S, hipMemcpyDeviceToDevice, e->stream));
		ZH_HIP(hipMemcpy2DAsync(nmag + (r % new_rows) * N, mrow * new_rows, e->d_mag + (r % old_rows) * N, mrow * old_rows, mrow, S,
		                        hipMemcpyDeviceToDevice, e->stream));
	}
	for (int o = 0; o < 3; ++o) {
		if (!e->d_Y[o])
			continue;
		ZH_HIP(hipMemsetAsync(nY[o], 0, sizeof(float) * S * new_hops * e->nwin, e->stream));
		if (e->last_frames > 0)
			ZH_HIP(hipMemcpy2DAsync(nY[o], sizeof(float) * new_hops * e->nwin, e->d_Y[o], sizeof(float) * old_hops * e->nwin,
			                        sizeof(float) * e->last_frames * e->nwin, S, hipMemcpyDeviceToDevice, e->stream));
	}
	ZH_HIP(hipStreamSynchronize(e->stream));
	(void)zh_free(e->d_S);
	(void)zh_free(e->d_mag);
	(void)zh_free(e->d_H);
	(void)zh_free(e->d_P);
	(void)zh_free(e->d_bits);
	(void)zh_free(e->d_bits_t);
	(void)zh_free(e->d_Mh);
	e->d_Mh = nullptr;
	e->d_S = nS;
	e->d_mag = nmag;
	e->d_H = e->d_P = nullptr; // ensure_estimates
	e->d_bits = e->d_bits_t = nullptr; // run_chunk
	(void)zh_free(e->d_blk_flag);
	(void)zh_free(e->d_blk_need);
	e->d_blk_flag = e->d_blk_need = nullptr; // run_hop_fused
	for (int o = 0; o < 3; ++o) {
		if (e->d_Y[o]) {
			(void)zh_free(e->d_Y[o]);
			e->d_Y[o] = nY[o];
		}
	}
	e->max_hops = new_hops;
	e->ring_rows = new_rows;
	return ZEN_HIP_OK;
}

// finished hops of output o of the last chunk, delivered as the spec says
int finalize_spec(zen_hip_hpr* e, int o, const HprOutSpec& sp, size_t M, long long pos0)
{
	if (!output_served(e, o))
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_process_spec: output %d is not computed by this engine", o);
	FinalizeArgs fa;
	memset(&fa, 0, sizeof(fa));
	fa.Y = e->d_Y[o];
	fa.carry = e->d_carry[o];
	fa.out = sp.dst;
	fa.y_stream_stride = (long long)(e->max_hops * e->nwin);
	fa.out_stride = sp.stride;
	fa.n_frames = (int)M;
	fa.hop = (int)e->hop;
	fa.n_streams = (int)e->n_streams;
	if (sp.add >= 0) {
		if (output_served(e, sp.add)) {
			fa.Y2 = e->d_Y[sp.add];
			fa.carry2 = e->d_carry[sp.add];
		}
		else {
			fa.add_zero = 1; // (soft masks, SSE: no residual -- the reference adds the all-zero accumulator)
		}
	}
	fa.pos0 = pos0;
	fa.shift = sp.shift;
	fa.len = sp.len;
	fa.dup_from = sp.dup_from;
	fa.dup_shift = sp.dup_shift;
	fa.dup_len = sp.dup_len;
	ProfScope ps(e, zen_hip_hpr::K_FINALIZE);
	return launch_finalize_spec(fa, e->stream);
}

// Run length of istft_run_wide_kernel (a workgroup per run and group of outputs; nfft >= 2048).  Its workgroups are long --
// run + 1 transforms per output, 10 us each at nfft 16384 -- and few, so how evenly they fill the device decides more than
// the 1/run of redundant work does (pass 1 of the offline batch, 64 clips x 324 frames: runs of 54 / 41 / 81 frames take
// 2.90 / 2.70 / 2.63 ms).  The launch is therefore simulated for every number of runs per stream: workgroups in dispatch
// order (groups of two outputs first), each to the slot that is free first; the shortest makespan wins.  Returns 0 when
// even that leaves the device less than `min_eff` busy with useful transforms: the caller keeps the per-frame launches,
// whose workgroups are one transform long.
// (plan_wide_run: the simulation itself, zen_hip_run_plan's body; pick_wide_run: the engine's cached question)
int plan_wide_run(size_t M, size_t S, const int* group_outputs, int n_groups, int log2n, double* eff)
{
	const size_t slots = 256 * (size_t)(log2n >= 14 ? 1 : (log2n == 13 ? 2 : 4)); // workgroups the device holds at once
	size_t useful = 0;
	for (int g = 0; g < n_groups; ++g)
		useful += (size_t)group_outputs[g] * M * S;
	*eff = 0.0;
	if (M == 0 || S == 0 || n_groups <= 0)
		return 0;
	if (S * (size_t)n_groups >= 16 * slots) { // so many streams that any split fills the device: long runs, nothing to simulate
		*eff = 1.0;
		return (int)(M < 96 ? M : (M + (M + 95) / 96 - 1) / ((M + 95) / 96));
	}
	int best = 0;
	double best_t = 0.0;
	std::vector<double> busy(slots);
	for (size_t k = 1; k <= 64 && k <= M; ++k) { // runs per stream
		const size_t run = (M + k - 1) / k, kk = (M + run - 1) / run;
		if (run < 8 || (best != 0 && S * kk * (size_t)n_groups > 64 * slots)) // (more workgroups than that balance themselves)
			break;
		std::fill(busy.begin(), busy.end(), 0.0);
		std::make_heap(busy.begin(), busy.end(), std::greater<double>());
		for (int g = 0; g < n_groups; ++g)
			for (size_t s = 0; s < S; ++s)
				for (size_t r = 0; r < kk; ++r) {
					const size_t len = (r + 1 == kk ? M - r * run : run) + 1;
					std::pop_heap(busy.begin(), busy.end(), std::greater<double>());
					busy.back() += (double)group_outputs[g] * (double)len;
					std::push_heap(busy.begin(), busy.end(), std::greater<double>());
				}
		const double t = *std::max_element(busy.begin(), busy.end());
		if (best == 0 || t < best_t * 0.995) { // (fewer, longer runs on a tie)
			best = (int)run;
			best_t = t;
		}
	}
	if (best != 0)
		*eff = (double)useful / ((double)slots * best_t);
	return best;
}

int pick_wide_run(zen_hip_hpr* e, size_t M, size_t S, const int* group_outputs, int n_groups, int log2n, double min_eff)
{
	int sig = n_groups; // (the simulation costs up to a millisecond: once per shape)
	for (int g = 0; g < n_groups; ++g)
		sig = sig * 4 + group_outputs[g];
	if (!(e->run_pick_M == M && e->run_pick_sig == sig)) {
		e->run_pick_M = M;
		e->run_pick_sig = sig;
		e->run_pick = plan_wide_run(M, S, group_outputs, n_groups, log2n, &e->run_pick_eff);
	}
	return e->run_pick_eff < min_eff ? 0 : e->run_pick;
}

// Can the pass be synthesised in runs (IstftRunArgs)?  Fills e->run_groups.  What it takes: a fresh stream (the pass starts
// from reset buffers: the frame before the first is all zero), an anticausal hard-mask median engine whose chunks all take
// the masks-as-bits road (>= 8 frames; the median kernel or the transposition leaves IstftArgs::bits_t), every destination
// the finished hops of one computed output or the sum of two, no output in two destinations, and a kernel for the
// transform and the groups (istft_run_available).  Computed outputs nobody asks for are not synthesised at all.
bool run_pass_groups(zen_hip_hpr* e, size_t n_hops, const HprOutSpec (&spec)[3])
{
	e->run_n_groups = 0;
	if (g_opt_no_istft_runs == 1 || g_opt_no_mask_bits || g_opt_no_half_rows || g_opt_median_general || g_opt_no_median_bits)
		return false;
	if (e->use_sse || e->soft || e->rows_stale)
		return false;
	if (e->abs_frame != (long long)e->W - 1 || e->last_frames != 0 || (e->drain[0] | e->drain[1] | e->drain[2]))
		return false;
	int used[3] = {0, 0, 0}, ng = 0, max_out = 0;
	for (int o = 0; o < 3; ++o) {
		if (!spec[o].dst)
			continue;
		zen_hip_hpr::RunGroup& g = e->run_groups[ng++];
		g.n_out = 1;
		g.which[0] = o;
		g.spec = &spec[o];
		if (!output_computed(e, o))
			return false;
		++used[o];
		if (spec[o].add >= 0) {
			if (spec[o].add > 2 || !output_computed(e, spec[o].add))
				return false; // (a partner that is not computed: the reference adds its zero accumulator -- left to finalize_spec_kernel)
			g.which[g.n_out++] = spec[o].add;
			++used[spec[o].add];
		}
		max_out = g.n_out > max_out ? g.n_out : max_out;
	}
	if (ng == 0 || used[0] > 1 || used[1] > 1 || used[2] > 1 || !istft_run_available(e->log2n, ng, max_out))
		return false;
	if (e->log2n > 10 && g_opt_no_istft_runs == 2) // ("no_istft_runs" = 2: only the transforms a wavefront holds)
		return false;
	if (e->log2n > 10 && g_opt_istft_run_wide <= 0) { // long workgroups: only where they fill the device (pick_wide_run)
		int outs[3];
		for (int g = 0; g < ng; ++g)
			outs[g] = e->run_groups[g].n_out;
		const size_t M0 = n_hops < e->max_hops ? n_hops : e->max_hops;
		if (pick_wide_run(e, M0, e->n_streams, outs, ng, e->log2n, 0.92) == 0)
			return false;
	}
	const size_t last = n_hops % e->max_hops, first = n_hops < e->max_hops ? n_hops : e->max_hops;
	if (first < 8 || (last != 0 && last < 8)) // (every chunk of the pass: run_chunk's masks-as-bits road starts at 8 frames)
		return false;
	const HardThr thr = hard_mask_thresholds(e->beta, e->beta - FLT_EPSILON, false);
	if (thr.p == 0.0 || thr.h == 0.0 || !mask_bits_supported((int)e->nfft, e->mf / 2) || !filter_supports_hermitian(e->mf, (int)e->nfft)
	    || !(e->mt == 1 || e->mt <= 63))
		return false;
	for (int g = 1; g < ng; ++g) // the groups of two outputs first (istft_run_wide_kernel dispatches them first)
		for (int k = g; k > 0 && e->run_groups[k].n_out > e->run_groups[k - 1].n_out; --k)
			std::swap(e->run_groups[k], e->run_groups[k - 1]);
	e->run_n_groups = ng;
	return true;
}

} // namespace

namespace zen_hip_impl {

int hpr_reserve_hops(zen_hip_hpr* h, size_t n_hops) // the growth a call of n_hops hops would start with, ahead of time
{
	if (n_hops > h->max_hops && h->max_hops < h->max_hops_cap)
		ZH_TRY(grow_buffers(h, n_hops < h->max_hops_cap ? n_hops : h->max_hops_cap));
	return ZEN_HIP_OK;
}

int hpr_process_spec(zen_hip_hpr* h, const float* in_dev, size_t n_hops, size_t in_stride, long long in_valid,
                     const HprOutSpec (&spec)[3])
{
	if (!h || !in_dev)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_process_spec: null argument");
	if (h->causality != ZEN_HIP_TIME_ANTICAUSAL)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_process_spec: anticausal engines only");
	if (n_hops > h->max_hops && h->max_hops < h->max_hops_cap)
		ZH_TRY(grow_buffers(h, n_hops < h->max_hops_cap ? n_hops : h->max_hops_cap));
	const bool runs = run_pass_groups(h, n_hops, spec);
	if (runs) { // the whole pass, or none of it: the two ways keep different state between chunks
		const size_t hopb = sizeof(float) * h->n_streams * h->hop;
		for (int g = 0; g < h->run_n_groups; ++g)
			for (int k = 0; k < h->run_groups[g].n_out; ++k) {
				const int o = h->run_groups[g].which[k];
				if (h->d_run_carry[0][o])
					continue; // (zeroed by the reset the pass started from: reset_state)
				for (int b = 0; b < 2; ++b) {
					ZH_HIP(zh_malloc((void**)&h->d_run_carry[b][o], hopb));
					ZH_HIP(hipMemsetAsync(h->d_run_carry[b][o], 0, hopb, h->stream)); // a fresh stream: nothing to add to its first hop
				}
			}
		if (!h->d_run_sink)
			ZH_HIP(zh_malloc((void**)&h->d_run_sink, hopb));
		h->run_mode = true;
	}
	int rc = ZEN_HIP_OK;
	for (size_t off = 0; off < n_hops && rc == ZEN_HIP_OK; off += h->max_hops) {
		const size_t M = (n_hops - off < h->max_hops) ? n_hops - off : h->max_hops;
		const long long pos0 = (long long)(off * h->hop);
		h->run_pos0 = pos0;
		rc = run_chunk(h, in_dev + off * h->hop, in_stride, M, in_valid - pos0);
		for (int o = 0; o < 3 && rc == ZEN_HIP_OK && !runs; ++o)
			if (spec[o].dst)
				rc = finalize_spec(h, o, spec[o], M, pos0);
	}
	h->run_mode = false;
	h->run_n_groups = 0;
	return rc;
}

} // namespace zen_hip_impl

extern "C" {

int zen_hip_hpr_create(float fs, size_t hop, float beta, unsigned output_flags, int causality,
                       int copy_bord, size_t n_streams, size_t max_hops_per_chunk, zen_hip_hpr_t* h)
{
	(void)copy_bord; // replicate border always (the reference CPU backend ignores it too, mfilt.h:289)
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_create: null handle");
	if (causality != ZEN_HIP_TIME_CAUSAL && causality != ZEN_HIP_TIME_ANTICAUSAL)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_create: causality must be TIME_CAUSAL or TIME_ANTICAUSAL");
	if (hop == 0 || n_streams == 0 || !(fs > 0))
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_create: hop, n_streams and fs must be positive");
	const size_t nwin = 2 * hop, nfft = 4 * hop; // hps.h:224-225
	if (!is_pow2(nfft))
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_create: nfft = 4*hop = %zu must be a power of two (fftw.h:59)", nfft);
	if (nfft < 32 || nfft > 16384)
		ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "hpr_create: nfft %zu outside 32..16384", nfft);
	// hps.h:227-230, evaluated with the reference's mixed float/double arithmetic
	const int l_harm = (int)roundf((float)(0.2 / (double)((float)(nfft - hop) / fs)));
	const int l_perc = (int)roundf(500.0F / (fs / (float)nfft));
	if (l_harm < 1 || l_perc < 1)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_create: degenerate filter lengths l_harm %d l_perc %d", l_harm, l_perc);
	const size_t W = (size_t)(2 * l_harm);
	int mt = 0, mf = 0; // the four filter objects of hps.h:246-258 may throw
	ZH_TRY(check_filter_len((int)W, (int)nfft, l_harm, causality, &mt));
	ZH_TRY(check_filter_len((int)W, (int)nfft, l_perc, ZEN_HIP_FREQUENCY, &mf));
	// (any mask length the reference accepts: beyond the specialised kernels' lengths the general kernels of median.hip /
	// box.hip take over.  In an engine neither mask exceeds 267 taps: l_harm >= 1 needs fs / hop >= 7.5, where l_perc =
	// 2000 hop / fs is 267 at most; l_perc >= 1 needs fs / hop <= 4000, where l_harm = fs / (15 hop) is 267 at most.  The
	// drop-in filter classes take any length up to the matrix dimension.)

	zen_hip_hpr* e = new zen_hip_hpr;
	e->fs = fs;
	e->hop = hop;
	e->nwin = nwin;
	e->nfft = nfft;
	e->beta = beta;
	e->l_harm = l_harm;
	e->l_perc = l_perc;
	e->lag = (causality == ZEN_HIP_TIME_CAUSAL) ? 1 : l_harm; // hps.h:228, :265-268
	e->W = W;
	e->causality = causality;
	e->log2n = ilog2(nfft);
	e->mt = mt;
	e->mf = mf;
	e->out_h = (output_flags & ZEN_HIP_OUTPUT_HARMONIC) != 0; // hps.h:276-284
	e->out_p = (output_flags & ZEN_HIP_OUTPUT_PERCUSSIVE) != 0;
	e->out_r = (output_flags & ZEN_HIP_OUTPUT_RESIDUAL) != 0;
	e->use_sse = false;
	e->soft = false;
	e->n_streams = n_streams;
	if (max_hops_per_chunk == 0) {
		// At most 2^30 ring elements per engine (S 8 B + |S| 4 B each = 13 GB, plus 4 B per element for each of H / P
		// and 2 B for each output's Y rows as they come into use: up to ~25 GB for the three-output pass of 64 offline
		// clips; the device has 288): whole clips go through in one or two chunks, every launch is large (offline batch
		// +6 % against 2^26, measured).  That is a CAP: the buffers start at one hop (a per-hop realtime engine never
		// needs more: its rings stay at stft_width rows) and grow to the size of the block calls that come
		// (grow_buffers), so a three-second clip does not cost what a ten-minute one does.
		size_t m = ((size_t)1 << 30) / (n_streams * nfft);
		e->max_hops_cap = m < 1 ? 1 : (m > 65536 ? 65536 : m);
		max_hops_per_chunk = 1;
	}
	else {
		e->max_hops_cap = max_hops_per_chunk; // the caller's bound on device memory: allocated as asked, never grown
	}
	e->max_hops = max_hops_per_chunk;
	e->ring_rows = (long long)(e->max_hops + W - 1);
	e->s_stride = (nfft / 2 + 1 + 7) & ~(size_t)7;
	e->stream = nullptr;

	// host tables, shared bit-for-bit with the oracle
	std::vector<float> win(nwin), tw(nfft);
	make_window_sqrt_hann(win.data(), nwin); // hps.h:232
	make_twiddles(tw.data(), nfft);
	float cola = 0.0f; // hps.h:270-274
	for (size_t i = 0; i < nwin; ++i)
		cola += win[i] * win[i];
	e->cola = (float)nfft / cola;

	const size_t S = n_streams, MH = e->max_hops;
	bool ok = zh_malloc((void**)&e->d_window, sizeof(float) * nwin) == hipSuccess
	          && zh_malloc((void**)&e->d_tw, sizeof(float) * nfft) == hipSuccess
	          && zh_malloc((void**)&e->d_tail[0], sizeof(float) * S * hop) == hipSuccess
	          && zh_malloc((void**)&e->d_tail[1], sizeof(float) * S * hop) == hipSuccess
	          && zh_malloc((void**)&e->d_S, sizeof(float2) * S * e->ring_rows * e->s_stride) == hipSuccess
	          && zh_malloc((void**)&e->d_mag, sizeof(float) * S * e->ring_rows * nfft) == hipSuccess;
	// d_H, d_P and d_Y[] are allocated by the first call that needs them (ensure_estimates / ensure_rows)
	for (int o = 0; o < 3 && ok; ++o)
		ok = zh_malloc((void**)&e->d_carry[o], sizeof(float) * S * hop) == hipSuccess;
	for (int o = 0; o < 3 && ok; ++o) {
		if (S == 1) { // io.h:24-66 style: pinned, mapped; the kernel writes it over the host link
			void* dev = nullptr;
			ok = zh_host_malloc((void**)&e->ready_host[o], sizeof(float) * (hop + 16), hipHostMallocMapped | hipHostMallocPortable) == hipSuccess
			     && zh_host_device_pointer(&dev, e->ready_host[o]) == hipSuccess;
			e->ready_dev[o] = (float*)dev;
			if (ok)
				memset(e->ready_host[o], 0, sizeof(float) * (hop + 16)); // [hop] = sequence word
		}
		else {
			ok = zh_malloc((void**)&e->ready_dev[o], sizeof(float) * S * hop) == hipSuccess;
		}
	}
	if (ok)
		ok = hipMemcpy(e->d_window, win.data(), sizeof(float) * nwin, hipMemcpyHostToDevice) == hipSuccess
		     && hipMemcpy(e->d_tw, tw.data(), sizeof(float) * nfft, hipMemcpyHostToDevice) == hipSuccess;
	if (!ok) {
		free_all(e);
		delete e;
		(void)hipGetLastError();
		ZH_FAIL(ZEN_HIP_E_HIP, "hpr_create: device allocation failed (nfft %zu, streams %zu, chunk %zu hops)",
		        nfft, n_streams, max_hops_per_chunk);
	}
	// once, so that no kernel can ever see uninitialised memory; reset_buffers() clears only what is read
	(void)hipMemsetAsync(e->d_S, 0, sizeof(float2) * S * e->ring_rows * e->s_stride, e->stream);
	(void)hipMemsetAsync(e->d_mag, 0, sizeof(float) * S * e->ring_rows * nfft, e->stream);
	(void)MH;
	int rc = reset_state(e);
	if (rc != ZEN_HIP_OK) {
		free_all(e);
		delete e;
		return rc;
	}
	*h = e;
	return ZEN_HIP_OK;
}

int zen_hip_hpr_destroy(zen_hip_hpr_t h)
{
	if (h) {
		(void)resident_stop(h);
		(void)hipStreamSynchronize(h->stream);
		free_all(h);
		delete h;
	}
	return ZEN_HIP_OK;
}

int zen_hip_hpr_get_params(zen_hip_hpr_t h, zen_hip_hpr_params* p)
{
	if (!h || !p)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_get_params: null argument");
	p->hop = h->hop;
	p->nwin = h->nwin;
	p->nfft = h->nfft;
	p->stft_width = h->W;
	p->l_harm = h->l_harm;
	p->l_perc = h->l_perc;
	p->lag = h->lag;
	p->time_len = h->mt;
	p->freq_len = h->mf;
	p->cola_factor = h->cola;
	p->n_streams = h->n_streams;
	p->max_hops_per_chunk = h->max_hops_cap;
	return ZEN_HIP_OK;
}

int zen_hip_hpr_set_stream(zen_hip_hpr_t h, void* stream)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_set_stream: null handle");
	ZH_TRY(resident_stop(h));
	ZH_HIP(hipStreamSynchronize(h->stream));
	h->stream = (hipStream_t)stream;
	return ZEN_HIP_OK;
}

int zen_hip_hpr_use_sse_filter(zen_hip_hpr_t h)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null handle");
	ZH_TRY(resident_stop(h));
	const bool before[3] = {output_computed(h, 0), output_computed(h, 1), output_computed(h, 2)};
	h->use_sse = true;
	return park_dropped_outputs(h, before);
}

int zen_hip_hpr_use_soft_mask(zen_hip_hpr_t h)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null handle");
	ZH_TRY(resident_stop(h));
	const bool before[3] = {output_computed(h, 0), output_computed(h, 1), output_computed(h, 2)};
	h->soft = true;
	return park_dropped_outputs(h, before);
}

int zen_hip_hpr_set_resident(zen_hip_hpr_t h, int idle_ms)
{
	if (!h || idle_ms < 0 || idle_ms > 2000)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_set_resident: idle time must be 0 (off) .. 2000 ms");
	ZH_TRY(resident_stop(h));
	h->res_idle_ms = idle_ms;
	return ZEN_HIP_OK;
}

int zen_hip_hpr_resident_stats(zen_hip_hpr_t h, unsigned long long* launches, unsigned long long* hops, int* active)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null handle");
	ZH_TRY(resident_kick(h));
	if (launches)
		*launches = h->res_launches;
	if (hops)
		*hops = h->res_hops; // (of the launches that have ended)
	if (active)
		*active = h->res_active ? 1 : 0;
	return ZEN_HIP_OK;
}

int zen_hip_run_plan(size_t frames, size_t streams, size_t nfft, const int* group_outputs, int n_groups, int* run, double* busy)
{
	if (!group_outputs || n_groups < 1 || n_groups > 3 || !run || !is_pow2(nfft) || nfft < 2048 || nfft > 16384)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "zen_hip_run_plan: 1..3 groups of 1 or 2 outputs, nfft 2048..16384");
	for (int g = 0; g < n_groups; ++g)
		if (group_outputs[g] < 1 || group_outputs[g] > 2)
			ZH_FAIL(ZEN_HIP_E_BAD_ARG, "zen_hip_run_plan: a group has 1 or 2 outputs");
	double eff = 0.0;
	const int r = plan_wide_run(frames, streams, group_outputs, n_groups, ilog2(nfft), &eff);
	*run = eff < 0.92 ? 0 : r; // (run_pass_groups' threshold)
	if (busy)
		*busy = eff;
	return ZEN_HIP_OK;
}

int zen_hip_hpr_reset_buffers(zen_hip_hpr_t h)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null handle");
	ZH_TRY(resident_stop(h));
	return reset_state(h);
}

int zen_hip_hpr_process(zen_hip_hpr_t h, const float* in_dev, size_t n_hops, size_t in_stride,
                        float* out_harm_dev, float* out_perc_dev, float* out_resid_dev, size_t out_stride)
{
	if (!h || !in_dev)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_process: null argument");
	float* outs[3] = {out_perc_dev, out_harm_dev, out_resid_dev};
	if (n_hops > h->max_hops && h->max_hops < h->max_hops_cap)
		ZH_TRY(grow_buffers(h, n_hops < h->max_hops_cap ? n_hops : h->max_hops_cap));
	// A kernel that finishes hops itself stores them while other workgroups of the same launch still read the input: an
	// output that overlaps the input (in place, or shifted) is left to the overlap-add launch, which runs after every
	// read of the chunk -- as safe as before the kernel learned to write the outputs.
	bool aliased = false;
	{
		const size_t S = h->n_streams;
		const char* in0 = (const char*)in_dev;
		const char* in1 = (const char*)(in_dev + (S - 1) * in_stride + n_hops * h->hop);
		for (int o = 0; o < 3; ++o)
			if (outs[o]) {
				const char* o0 = (const char*)outs[o];
				const char* o1 = (const char*)(outs[o] + (S - 1) * out_stride + n_hops * h->hop);
				aliased = aliased || (o0 < in1 && in0 < o1);
			}
	}
	for (size_t off = 0; off < n_hops; off += h->max_hops) {
		const size_t M = (n_hops - off < h->max_hops) ? n_hops - off : h->max_hops;
		for (int o = 0; o < 3; ++o) // (a kernel that finishes hops itself writes them here: run_hop_fused)
			h->direct_out[o] = (outs[o] && !aliased) ? outs[o] + off * h->hop : nullptr;
		h->direct_stride = (long long)out_stride;
		const int rc = run_chunk(h, in_dev + off * h->hop, in_stride, M);
		for (int o = 0; o < 3; ++o)
			h->direct_out[o] = nullptr;
		ZH_TRY(rc);
		for (int o = 0; o < 3; ++o)
			if (outs[o] && !h->direct_done[o])
				ZH_TRY(finalize_output(h, o, outs[o] + off * h->hop, out_stride, M));
		for (int o = 0; o < 3; ++o)
			h->direct_done[o] = false;
	}
	return ZEN_HIP_OK;
}

// The block form on HOST buffers: the timed region of the reference's realtime tool -- host hop in, process, host hop out
// (zen/fakert.h:221-247) -- for a block of hops at once.  Pieces of the block go up, through zen_hip_hpr_process and back
// down on three streams (the pattern of zen_hip_hpri_process): piece k is processed while piece k+1 arrives and piece k-1
// leaves.  Pinned buffers (zen_hip_host_alloc_mapped) are copied asynchronously; pageable ones are registered for the
// duration of the call ("offline_no_register": not), and where that fails their copies block and the loop issues the
// upload of piece k+1 before the download of piece k.  Returns when the outputs are in the caller's buffers.
int zen_hip_hpr_process_host(zen_hip_hpr_t h, const float* in_host, size_t n_hops, float* out_harm_host, float* out_perc_host,
                             float* out_resid_host)
{
	if (!h || !in_host)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_process_host: null argument");
	if (h->n_streams != 1)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_process_host: one stream per engine (host blocks of several streams: one engine each)");
	if (n_hops == 0)
		return ZEN_HIP_OK;
	float* hosts[3] = {out_perc_host, out_harm_host, out_resid_host};
	const size_t n = n_hops * h->hop;
	for (int o = 0; o < 3; ++o) {
		const char *a0 = (const char*)in_host, *a1 = (const char*)(in_host + n);
		if (hosts[o] && (const char*)hosts[o] < a1 && a0 < (const char*)(hosts[o] + n))
			ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_process_host: an output overlaps the input (pieces come down while later pieces go up)");
	}
	ZH_TRY(resident_stop(h));
	if (n > h->hstage_cap || (hosts[0] && !h->hstage_out[0]) || (hosts[1] && !h->hstage_out[1]) || (hosts[2] && !h->hstage_out[2])) {
		ZH_HIP(hipStreamSynchronize(h->stream));
		const size_t cap = n > h->hstage_cap ? n : h->hstage_cap;
		if (cap > h->hstage_cap) {
			(void)zh_free(h->hstage_in);
			h->hstage_in = nullptr;
			for (int o = 0; o < 3; ++o) {
				(void)zh_free(h->hstage_out[o]);
				h->hstage_out[o] = nullptr;
			}
			h->hstage_cap = 0;
			ZH_HIP(zh_malloc((void**)&h->hstage_in, sizeof(float) * cap));
			h->hstage_cap = cap;
		}
		for (int o = 0; o < 3; ++o)
			if (hosts[o] && !h->hstage_out[o])
				ZH_HIP(zh_malloc((void**)&h->hstage_out[o], sizeof(float) * h->hstage_cap));
	}
	// pieces of ~8 MiB of input (2048 hops at hop 1024): long enough for full-rate copies and full-size launches, short
	// enough that the first upload and the last download -- which nothing overlaps -- stay a small share of the call
	size_t piece = g_opt_host_block_hops > 0 ? (size_t)g_opt_host_block_hops : ((size_t)2 << 20) / h->hop;
	if (piece < 1)
		piece = 1;
	const size_t n_pieces = (n_hops + piece - 1) / piece;
	if (!h->hs_in)
		ZH_HIP(hipStreamCreateWithFlags(&h->hs_in, hipStreamNonBlocking));
	if (!h->hs_out)
		ZH_HIP(hipStreamCreateWithFlags(&h->hs_out, hipStreamNonBlocking));
	while (h->hevents.size() < 2 * n_pieces + 1) {
		hipEvent_t ev;
		ZH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		h->hevents.push_back(ev);
	}
	ZH_TRY(hpr_reserve_hops(h, piece < n_hops ? piece : n_hops)); // (the engine's growth synchronises: before the first copy is queued)
	const bool try_register = g_opt_offline_no_register.load(std::memory_order_relaxed) == 0;
	Registered reg_in, reg_out[3];
	reg_in.take(in_host, sizeof(float) * n, try_register);
	for (int o = 0; o < 3; ++o)
		reg_out[o].take(hosts[o], sizeof(float) * n, try_register);
	auto upload = [&](size_t k) -> int { // event 2k
		const size_t b = k * piece * h->hop, e = (k + 1) * piece < n_hops ? (k + 1) * piece * h->hop : n;
		ZH_HIP(hipMemcpyAsync(h->hstage_in + b, in_host + b, sizeof(float) * (e - b), hipMemcpyHostToDevice, h->hs_in));
		ZH_HIP(hipEventRecord(h->hevents[2 * k], h->hs_in));
		return ZEN_HIP_OK;
	};
	auto feed = [&]() -> int {
		// behind whatever the caller queued on the engine's stream (and an earlier call's kernels, which may still read the stage)
		ZH_HIP(hipEventRecord(h->hevents[2 * n_pieces], h->stream));
		ZH_HIP(hipStreamWaitEvent(h->hs_in, h->hevents[2 * n_pieces], 0));
		ZH_TRY(upload(0));
		for (size_t k = 0; k < n_pieces; ++k) {
			const size_t h0 = k * piece, m = (k + 1) * piece < n_hops ? piece : n_hops - h0, b = h0 * h->hop;
			ZH_HIP(hipStreamWaitEvent(h->stream, h->hevents[2 * k], 0));
			ZH_TRY(zen_hip_hpr_process(h, h->hstage_in + b, m, m * h->hop, hosts[1] ? h->hstage_out[1] + b : nullptr,
			                           hosts[0] ? h->hstage_out[0] + b : nullptr, hosts[2] ? h->hstage_out[2] + b : nullptr, m * h->hop));
			ZH_HIP(hipEventRecord(h->hevents[2 * k + 1], h->stream));
			if (k + 1 < n_pieces)
				ZH_TRY(upload(k + 1));
			ZH_HIP(hipStreamWaitEvent(h->hs_out, h->hevents[2 * k + 1], 0));
			for (int o = 0; o < 3; ++o)
				if (hosts[o])
					ZH_HIP(hipMemcpyAsync(hosts[o] + b, h->hstage_out[o] + b, sizeof(float) * m * h->hop, hipMemcpyDeviceToHost, h->hs_out));
		}
		return ZEN_HIP_OK;
	};
	const int rc = feed();
	// nothing may be in flight when the caller's buffers are unregistered and handed back
	const hipError_t e_out = hipStreamSynchronize(h->hs_out), e_run = hipStreamSynchronize(h->stream), e_in = hipStreamSynchronize(h->hs_in);
	ZH_TRY(rc);
	ZH_HIP(e_out);
	ZH_HIP(e_run);
	ZH_HIP(e_in);
	return ZEN_HIP_OK;
}

int zen_hip_hpr_process_next_hop(zen_hip_hpr_t h, const float* in_dev)
{
	if (!h || !in_dev)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_process_next_hop: null argument");
	return run_chunk(h, in_dev, h->hop, 1, LLONG_MAX, /*may_post=*/true);
}

// host address of a copy_* destination if it is (mapped) host memory, else null; one lookup per distinct pointer
static void* host_alias_of(zen_hip_hpr_t h, const void* out_dev)
{
	const unsigned gen = g_host_free_gen.load(std::memory_order_relaxed);
	if (h->out_query_dev == out_dev && h->out_query_gen == gen)
		return h->out_query_host;
	hipPointerAttribute_t at;
	void* host = nullptr;
	if (hipPointerGetAttributes(&at, out_dev) == hipSuccess && at.type == hipMemoryTypeHost && at.hostPointer
	    && at.devicePointer) {
		// interior pointers: whatever the runtime reports for the pair, the two aliases differ by a constant
		host = (char*)at.hostPointer + ((const char*)out_dev - (const char*)at.devicePointer);
	}
	else {
		(void)hipGetLastError();
	}
	h->out_query_dev = out_dev;
	h->out_query_host = host;
	h->out_query_gen = gen;
	return host;
}

// rt_wide.hip: a grid barrier that gave up waiting (a cooperating workgroup never arrived) marks a host-mapped word
// before the kernel goes on with incomplete data; the sequence word is still published.  Every synchronous copy_*
// looks at the mark (the asynchronous one cannot: the next synchronous call, or reset_buffers, reports / clears it).
static int check_wide_fail(zen_hip_hpr_t h)
{
	if (h->wide_fail_host && __atomic_load_n(h->wide_fail_host, __ATOMIC_ACQUIRE))
		ZH_FAIL(ZEN_HIP_E_HIP, "hpr_copy_output: a workgroup of the cooperative single-hop kernel never reached its grid "
		                       "barrier; the hop is invalid (zen_hip_hpr_reset_buffers clears the mark)");
	return ZEN_HIP_OK;
}

static int copy_output_impl(zen_hip_hpr_t h, unsigned which, float* out_dev, bool sync)
{
	if (!h || !out_dev)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_copy_output: null argument");
	const int o = which_index(which);
	if (o < 0)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpr_copy_output: `which` must be exactly one ZEN_HIP_OUTPUT_* flag");
	if (h->last_frames == 0) { // before any hop: the zero accumulator
		ZH_HIP(hipMemsetAsync(out_dev, 0, sizeof(float) * h->hop * h->n_streams, h->stream));
	}
	else if (h->last_frames == 1 && h->ready_valid[o] && output_served(h, o)) {
		// the hop was finished by the synthesis kernel itself (InvOut / IstftOut): no launch here
		const size_t bytes = sizeof(float) * h->hop * h->n_streams;
		void* host = (sync && h->ready_host[o]) ? host_alias_of(h, out_dev) : nullptr;
		if (host) { // mapped host destination (zen::io::IOGPU::device_out): wait for the hop, host copy
			// The kernel publishes the call's sequence number behind the hop once every sample of it is visible
			// system-wide; polling that word saves the completion-signal round trip of a stream synchronise.
			const unsigned* flag = reinterpret_cast<const unsigned*>(h->ready_host[o] + h->hop);
			bool seen = false;
			for (long spin = 0; spin < 50000000L; ++spin) {
				if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == h->hop_seq) {
					seen = true;
					break;
				}
				if (h->res_active && (spin & 63) == 63) // a resident kernel that left with this hop pending is launched again
					ZH_TRY(resident_kick(h));
				__builtin_ia32_pause();
			}
			if (!seen || h->async_pending) { // a fault or a hang: let the runtime report it; or earlier asynchronous copies
				ZH_HIP(hipStreamSynchronize(h->stream));
				h->async_pending = false;
			}
			ZH_TRY(check_wide_fail(h));
			memcpy(host, h->ready_host[o], bytes);
			return ZEN_HIP_OK;
		}
		ZH_TRY(resident_wait(h)); // (a resident kernel works on a stream of its own: the hop must be there before the copy is queued)
		ZH_HIP(hipMemcpyAsync(out_dev, h->ready_dev[o], bytes, hipMemcpyDefault, h->stream));
	}
	else {
		ZH_TRY(resident_stop(h));
		const size_t M = h->last_frames;
		ZH_TRY(finalize_output(h, o, out_dev, M * h->hop, M));
	}
	if (sync) {
		ZH_HIP(hipStreamSynchronize(h->stream));
		h->async_pending = false;
		ZH_TRY(check_wide_fail(h));
	}
	else {
		h->async_pending = true;
	}
	return ZEN_HIP_OK;
}

int zen_hip_hpr_copy_output_async(zen_hip_hpr_t h, unsigned which, float* out_dev)
{
	return copy_output_impl(h, which, out_dev, false);
}

int zen_hip_hpr_copy_output(zen_hip_hpr_t h, unsigned which, float* out_dev)
{
	return copy_output_impl(h, which, out_dev, true);
}

static int prof_drain(zen_hip_hpr_t h)
{
	ZH_HIP(hipStreamSynchronize(h->stream));
	for (auto& p : h->prof_pending) {
		float ms = 0;
		ZH_HIP(hipEventElapsedTime(&ms, p.e0, p.e1));
		h->prof_ms[p.k] += ms;
		h->prof_pool.emplace_back(p.e0, p.e1);
	}
	h->prof_pending.clear();
	return ZEN_HIP_OK;
}

// diagnostic (tools/rt_latency.cpp): mapped host buffer that receives the fused kernel's phase stamps
int zen_hip_hpr_debug_stamps(zen_hip_hpr_t h, unsigned long long** host_stamps)
{
	if (!h || !host_stamps)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null argument");
	ZH_TRY(resident_stop(h));
	if (!h->dbg_stamps_host) {
		void* dev = nullptr;
		ZH_HIP(zh_host_malloc((void**)&h->dbg_stamps_host, 16 * sizeof(unsigned long long), hipHostMallocMapped));
		ZH_HIP(zh_host_device_pointer(&dev, h->dbg_stamps_host));
		h->dbg_stamps = (unsigned long long*)dev;
		memset(h->dbg_stamps_host, 0, 16 * sizeof(unsigned long long));
	}
	*host_stamps = h->dbg_stamps_host;
	return ZEN_HIP_OK;
}

int zen_hip_hpr_profile(zen_hip_hpr_t h, int enable)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null handle");
	ZH_TRY(resident_stop(h));
	ZH_TRY(prof_drain(h));
	h->prof = enable != 0;
	if (enable) {
		for (int k = 0; k < zen_hip_hpr::K_COUNT; ++k) {
			h->prof_ms[k] = 0;
			h->prof_launches[k] = 0;
		}
		h->prof_elements = 0;
	}
	return ZEN_HIP_OK;
}

int zen_hip_hpr_profile_get(zen_hip_hpr_t h, double* median_ms, unsigned long long* median_launches,
                            unsigned long long* median_elements)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null handle");
	ZH_TRY(prof_drain(h));
	if (median_ms)
		*median_ms = h->prof_ms[zen_hip_hpr::K_FREQ];
	if (median_launches)
		*median_launches = h->prof_launches[zen_hip_hpr::K_FREQ];
	if (median_elements)
		*median_elements = h->prof_elements;
	return ZEN_HIP_OK;
}

int zen_hip_hpr_profile_get_all(zen_hip_hpr_t h, double ms[6], unsigned long long launches[6])
{
	if (!h || !ms || !launches)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "null argument");
	ZH_TRY(prof_drain(h));
	for (int k = 0; k < zen_hip_hpr::K_COUNT; ++k) {
		ms[k] = h->prof_ms[k];
		launches[k] = h->prof_launches[k];
	}
	return ZEN_HIP_OK;
}

} // extern "C"

