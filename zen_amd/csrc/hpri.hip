// hpri.hip -- HPRIOffline<Backend::GPU> behind the C-ABI: two cascaded HPR passes ("HPR-I") on two streaming
// engines (hpr.hip), everything resident in HBM.
//
//   zen_hip_hpri_* : HPRIOffline<GPU>::process (libzen/hps.cu:128-221): pass 1 (hop_h; H, P, R) ->
//                    P+R shifted by lag_h*hop_h -> pass 2 (hop_p; P).  The reference walks the clip hop by hop
//                    through two IOGPU buffers (hps.cu:146-166,189-204); here each pass is a handful of
//                    block calls on the engine.  Also: a batch of equal-length clips per call, and any time
//                    range of one clip from its own warm-up halo.
#include "common.h"
#include "memguard.h"
#include "hpr_engine.h"
#include "host_pipe.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <chrono>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>

using namespace zen_hip_impl;

// =================================================================================================
// HPRIOffline<GPU>
// =================================================================================================
struct zen_hip_hpri {
	zen_hip_hpr* eh = nullptr; // p_impl_h : hop_h, H+P+R, anticausal (hps.cu:38-43)
	zen_hip_hpr* ep = nullptr; // p_impl_p : hop_p, P only, anticausal (hps.cu:45-48)
	size_t hop_h, hop_p, n_clips;
	hipStream_t stream = nullptr;
	// scratch, grown on demand: pass 2's input (the only intermediate that has to exist in memory)
	size_t cap2 = 0;
	float* in2 = nullptr;
	float *stage_in = nullptr, *stage_out[2] = {nullptr, nullptr}; // whole-clip device images: input, harmonic, percussive
	size_t stage_cap = 0;
	// zen_hip_hpri_process (host buffers): the copy streams of its pipeline and the events that order them
	hipStream_t s_in = nullptr, s_out = nullptr;
	std::vector<hipEvent_t> events;
	zen_hip_hpri_host_stats stats = {};
	// zen_hip_hpri_process_sink: pinned staging slots the ranges come down into (per output a ring of SINK_SLOTS), handed to
	// the caller's sink from there
	float* pin[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
	size_t pin_cap = 0; // floats per slot
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

void hpri_free_scratch(zen_hip_hpri* o)
{
	(void)zh_free(o->in2);
	o->in2 = nullptr;
	o->cap2 = 0;
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
	// "offline_chunk_hops": a bound on the engines' buffers (hops per chunk, both passes) instead of the device-memory cap
	const int chunk_opt = g_opt_offline_chunk_hops.load(std::memory_order_relaxed);
	const size_t chunk = chunk_opt > 0 ? (size_t)chunk_opt : 0;
	int rc = zen_hip_hpr_create(fs, hop_h, beta_h,
	                            ZEN_HIP_OUTPUT_HARMONIC | ZEN_HIP_OUTPUT_PERCUSSIVE | ZEN_HIP_OUTPUT_RESIDUAL,
	                            ZEN_HIP_TIME_ANTICAUSAL, !nocopybord, n_clips, chunk, &o->eh);
	if (rc == ZEN_HIP_OK)
		rc = zen_hip_hpr_create(fs, hop_p, beta_p, ZEN_HIP_OUTPUT_PERCUSSIVE, ZEN_HIP_TIME_ANTICAUSAL,
		                        !nocopybord, n_clips, chunk, &o->ep);
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
		(void)zh_free(h->stage_in);
		for (int i = 0; i < 2; ++i)
			(void)zh_free(h->stage_out[i]);
		for (hipEvent_t e : h->events)
			(void)hipEventDestroy(e);
		for (int i = 0; i < 2; ++i)
			for (int k = 0; k < 4; ++k)
				if (h->pin[i][k])
					(void)zh_host_free(h->pin[i][k]);
		if (h->s_in)
			(void)hipStreamDestroy(h->s_in);
		if (h->s_out)
			(void)hipStreamDestroy(h->s_out);
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
	// One scratch buffer: pass 2's input.  The reference pads the clip (hps.cu:116-123), collects three padded outputs
	// of pass 1, adds two of them, shifts the sum and the harmonic output left by lag_h*hop_h in place and truncates
	// (hps.cu:153-178), and does the same after pass 2 (hps.cu:209-217).  Here the analysis kernel reads the clip
	// itself (samples beyond n are zero) and the overlap-add kernel writes every finished sample where it ends up
	// (hpr_process_spec): no padded copy, no P1 / R1 / H1 / P2 buffers, no sum / shift / truncate passes.
	if (padded2 > h->cap2) {
		ZH_HIP(hipStreamSynchronize(h->stream));
		hpri_free_scratch(h);
		ZH_HIP(zh_malloc((void**)&h->in2, sizeof(float) * C * padded2));
		h->cap2 = padded2;
	}
	ZH_TRY(zen_hip_hpr_reset_buffers(h->eh)); // each process() call is a fresh pair of HPR objects' state
	ZH_TRY(zen_hip_hpr_reset_buffers(h->ep));
	const size_t sh1 = (size_t)h->eh->lag * h->hop_h, sh2 = (size_t)h->ep->lag * h->hop_p;

	// pass 1: large hop, harmonic + percussive + residual (hps.cu:142-167)
	HprOutSpec s1[3];
	if (harm_dev) { // harm[j] = H1[j + sh1], j < n (hps.cu:171-178; n <= padded1 - sh1 always)
		s1[1].dst = harm_dev;
		s1[1].stride = (long long)out_stride;
		s1[1].shift = (long long)sh1;
		s1[1].len = (long long)n;
	}
	// intermediate[j] = (P1 + R1)[j + sh1] for j < padded1 - sh1; the in-place shift leaves (P1 + R1)[j] in
	// [padded1 - sh1, padded1), which pass 2 reads past size() (hps.cu:186-190); beyond padded1: the
	// reference's allocation ends (zeros here)
	s1[0].dst = h->in2;
	s1[0].stride = (long long)padded2;
	s1[0].shift = (long long)sh1;
	s1[0].len = (long long)padded2;
	s1[0].add = 2;
	s1[0].dup_from = (long long)(padded1 - sh1);
	s1[0].dup_shift = 0;
	s1[0].dup_len = (long long)padded2;
	if (padded2 > padded1)
		ZH_HIP(hipMemset2DAsync(h->in2 + padded1, sizeof(float) * padded2, 0, sizeof(float) * (padded2 - padded1), C, h->stream));
	ZH_TRY(hpr_process_spec(h->eh, audio_dev, (size_t)n1, stride, (long long)n, s1));
	// pass 2: small hop on xp1 + xr1, percussive only (hps.cu:185-205); perc[j] = P2[j + sh2], j < n (hps.cu:209-217)
	HprOutSpec s2[3];
	if (perc_dev) {
		s2[0].dst = perc_dev;
		s2[0].stride = (long long)out_stride;
		s2[0].shift = (long long)sh2;
		s2[0].len = (long long)n;
	}
	ZH_TRY(hpr_process_spec(h->ep, h->in2, (size_t)n2, padded2, (long long)padded2, s2));
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
	const size_t c2 = (p.m1 - p.q2) * hop_p;
	if (c2 > h->cap2) {
		ZH_HIP(hipStreamSynchronize(h->stream));
		hpri_free_scratch(h);
		ZH_HIP(zh_malloc((void**)&h->in2, sizeof(float) * (c2 ? c2 : 1)));
		h->cap2 = c2;
	}
	ZH_TRY(zen_hip_hpr_reset_buffers(h->eh));
	ZH_TRY(zen_hip_hpr_reset_buffers(h->ep));
	// Positions below are relative to the first sample each pass is run on: base1 = q1*hop_h of the clip for pass 1,
	// base2 = q2*hop_p for pass 2.  Same destinations as zen_hip_hpri_process_device, offset by the range.
	const long long base1 = (long long)(p.q1 * hop_h), base2 = (long long)(p.q2 * hop_p);
	const long long sh1 = (long long)p.sh1, sh2 = (long long)p.sh2, cnt = (long long)(end - begin);
	HprOutSpec s1[3];
	if (harm_dev) { // harm[begin + i] = H1[begin + i + sh1]
		s1[1].dst = harm_dev;
		s1[1].shift = sh1 + (long long)begin - base1;
		s1[1].len = cnt;
	}
	if (c2) { // pass-2 input positions base2 + i, i < c2: (P1 + R1)[base2 + i + sh1], or the unshifted leftovers near the end (Q9)
		s1[0].dst = h->in2;
		s1[0].shift = sh1 + base2 - base1;
		s1[0].len = (long long)c2;
		s1[0].add = 2;
		s1[0].dup_from = (long long)p.padded1 - sh1 - base1;
		s1[0].dup_shift = base2 - base1;
		s1[0].dup_len = (long long)c2;
		if (base2 + (long long)c2 > (long long)p.padded1) { // beyond the reference's allocation: zeros
			const long long z0 = (long long)p.padded1 > base2 ? (long long)p.padded1 - base2 : 0;
			ZH_HIP(hipMemsetAsync(h->in2 + z0, 0, sizeof(float) * ((long long)c2 - z0), h->stream));
		}
	}
	ZH_TRY(hpr_process_spec(h->eh, audio_dev + base1, p.k1 - p.q1, (p.k1 - p.q1) * hop_h, (long long)n - base1, s1));
	if (c2) {
		HprOutSpec s2[3];
		if (perc_dev) { // perc[begin + i] = P2[begin + i + sh2]
			s2[0].dst = perc_dev;
			s2[0].shift = sh2 + (long long)begin - base2;
			s2[0].len = cnt;
		}
		ZH_TRY(hpr_process_spec(h->ep, h->in2, p.m1 - p.q2, c2, (long long)c2, s2));
	}
	else if (perc_dev) {
		ZH_HIP(hipMemsetAsync(perc_dev, 0, sizeof(float) * (end - begin), h->stream));
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

// ---- HPRIOffline<GPU>::process on host vectors (hps.cu:128-221) ----------------------------------------
// What the reference times (zen/offline.h:141-147) is this call, copies included.  Resident in HBM the two passes take
// ~13 ms per hour of audio; the host link needs 4 bytes in and 8 bytes out per sample (~11 + ~22 ms per hour at the
// ~57 GB/s a pinned copy gets), so the call is a pipeline over time ranges of the clip: the input of range k+1 goes up
// and the outputs of range k-1 come down under the kernels of range k (zen_hip_hpri_process_range: bit-identical to the
// whole clip, each range from its own warm-up halo).  Three streams -- copy in, the handle's stream, copy out -- and one
// event per range and edge.  The third output is all zeros: written by the host, never moved.
extern "C++" {
namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void zero_host(float* dst, size_t n) // the reference's never-written residual_out (hps.cu:45-48, :200-204)
{
	const size_t bytes = n * sizeof(float), piece = (size_t)64 << 20;
	unsigned k = (unsigned)(bytes / piece);
	const unsigned hw = std::thread::hardware_concurrency();
	if (k > 8)
		k = 8;
	if (hw && k > hw / 2)
		k = hw / 2;
	if (k < 2) {
		memset(dst, 0, bytes);
		return;
	}
	std::vector<std::thread> th;
	const size_t per = (n / k + 1023) & ~(size_t)1023;
	size_t done_to = 0; // (a thread that cannot be started -- std::system_error must not cross the C ABI -- leaves its piece to this one)
	for (unsigned i = 0; i < k; ++i) {
		const size_t a = (size_t)i * per, b = a + per < n ? a + per : n;
		if (a >= b)
			break;
		try {
			th.emplace_back([=] { memset(dst + a, 0, (b - a) * sizeof(float)); });
			done_to = b;
		}
		catch (...) {
			break;
		}
	}
	if (done_to < n)
		memset(dst + done_to, 0, (n - done_to) * sizeof(float));
	for (auto& t : th)
		t.join();
}

int ensure_copy_streams(zen_hip_hpri* h, size_t n_events)
{
	if (!h->s_in)
		ZH_HIP(hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
	if (!h->s_out)
		ZH_HIP(hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
	while (h->events.size() < n_events) {
		hipEvent_t e;
		ZH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		h->events.push_back(e);
	}
	return ZEN_HIP_OK;
}

} // namespace
} // extern "C++"

extern "C++" {
namespace {
constexpr int SINK_SLOTS = 4;

// Pinned staging for the sink form: SINK_SLOTS slots of `range` samples per wanted output, kept for the handle's life.
int ensure_pins(zen_hip_hpri* h, size_t range, const bool (&want)[2])
{
	if (range > h->pin_cap) {
		for (int i = 0; i < 2; ++i)
			for (int k = 0; k < SINK_SLOTS; ++k)
				if (h->pin[i][k]) {
					(void)zh_host_free(h->pin[i][k]);
					h->pin[i][k] = nullptr;
				}
		h->pin_cap = range;
	}
	for (int i = 0; i < 2; ++i)
		for (int k = 0; k < SINK_SLOTS; ++k)
			if (want[i] && !h->pin[i][k])
				ZH_HIP(zh_host_malloc((void**)&h->pin[i][k], sizeof(float) * h->pin_cap, hipHostMallocDefault));
	return ZEN_HIP_OK;
}

int hpri_process_impl(zen_hip_hpri_t h, const float* audio_host, size_t n, float* harm_host, float* perc_host, float* resid_host,
                      zen_hip_hpri_sink_fn sink, void* sink_user, bool sink_harm, bool sink_perc);
} // namespace
} // extern "C++"

int zen_hip_hpri_process(zen_hip_hpri_t h, const float* audio_host, size_t n, float* harm_host,
                         float* perc_host, float* resid_host)
{
	return hpri_process_impl(h, audio_host, n, harm_host, perc_host, resid_host, nullptr, nullptr, false, false);
}

int zen_hip_hpri_process_sink(zen_hip_hpri_t h, const float* audio_host, size_t n, int want_harm, int want_perc,
                              zen_hip_hpri_sink_fn sink, void* user)
{
	if (!sink)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process_sink: null sink");
	return hpri_process_impl(h, audio_host, n, nullptr, nullptr, nullptr, sink, user, want_harm != 0, want_perc != 0);
}

extern "C++" {
namespace {
int hpri_process_impl(zen_hip_hpri_t h, const float* audio_host, size_t n, float* harm_host, float* perc_host, float* resid_host,
                      zen_hip_hpri_sink_fn sink, void* sink_user, bool sink_harm, bool sink_perc)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process: null handle");
	if (n == 0)
		return ZEN_HIP_OK; // the reference returns three empty vectors (hps.cu:128-221 on an empty input)
	if (!audio_host)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process: null argument");
	if (h->n_clips != 1)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process(host) needs a handle created with n_clips == 1");
	{ // the reference takes the clip by value and returns new vectors: the four buffers are distinct.  Ranges of the clip go
	  // up while finished ranges come down, so an output that overlaps the input (or another output) would be corrupted.
		const char* p[4] = {(const char*)audio_host, (const char*)harm_host, (const char*)perc_host, (const char*)resid_host};
		const size_t bytes = sizeof(float) * n;
		for (int i = 0; i < 4; ++i)
			for (int j = i + 1; j < 4; ++j)
				if (p[i] && p[j] && p[i] < p[j] + bytes && p[j] < p[i] + bytes)
					ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_process: the host buffers must not overlap");
	}
	const double t_start = now_ms();
	if (n > h->stage_cap) {
		ZH_HIP(hipStreamSynchronize(h->stream));
		(void)zh_free(h->stage_in);
		h->stage_in = nullptr; // a failed allocation below must not leave freed pointers behind a stale capacity
		for (int i = 0; i < 2; ++i) {
			(void)zh_free(h->stage_out[i]);
			h->stage_out[i] = nullptr;
		}
		h->stage_cap = 0;
		ZH_HIP(zh_malloc((void**)&h->stage_in, sizeof(float) * n));
		for (int i = 0; i < 2; ++i)
			ZH_HIP(zh_malloc((void**)&h->stage_out[i], sizeof(float) * n));
		h->stage_cap = n;
	}
	float* hosts[2] = {harm_host, perc_host};
	const bool want_o[2] = {sink ? sink_harm : harm_host != nullptr, sink ? sink_perc : perc_host != nullptr};
	// ranges of the pipeline: equal lengths, a multiple of hop_h (the buffers are sized for the largest one up front, below)
	size_t want = (size_t)g_opt_offline_range.load(std::memory_order_relaxed);
	if (want == 0) {
		// 8 Mi samples (32 MB up, 64 MB down, ~0.6 + ~1.1 ms on the link, ~0.8 ms of kernels), 4 Mi for clips that would
		// otherwise be two or three ranges; shorter ranges cost more in kernels than they hide in copies (a one-hour clip:
		// 25.7 ms in ranges of 8 Mi, 38 in ranges of 4 Mi, 88 in ranges of 2 Mi: each range re-runs its warm-up halo, resets
		// the engines and launches grids too small for 256 CUs), and under 8 Mi samples (3 minutes) the clip is one range
		want = n >= ((size_t)1 << 25) ? (size_t)1 << 23 : (size_t)1 << 22;
		if (n < ((size_t)1 << 23))
			want = n;
	}
	size_t n_ranges = ceil_div(n, want);
	size_t range = ceil_div(ceil_div(n, n_ranges), h->hop_h) * h->hop_h;
	n_ranges = ceil_div(n, range);
	zen_hip_hpri_host_stats& st = h->stats;
	memset(&st, 0, sizeof(st));
	st.n_ranges = n_ranges;
	st.range_samples = range;
	std::thread zero_thread;
	if (resid_host) {
		try {
			zero_thread = std::thread([=] { zero_host(resid_host, n); });
		}
		catch (...) { // no thread to be had: zeros written here, before the pipeline starts
			zero_host(resid_host, n);
		}
	}
	struct Joiner {
		std::thread& t;
		~Joiner()
		{
			if (t.joinable())
				t.join();
		}
	} joiner{zero_thread};
	if (sink)
		ZH_TRY(ensure_pins(h, range < n ? range : n, want_o));
	if (n_ranges == 1) { // a short clip: up, both passes, down
		ZH_HIP(hipMemcpyAsync(h->stage_in, audio_host, sizeof(float) * n, hipMemcpyHostToDevice, h->stream));
		ZH_TRY(zen_hip_hpri_process_device(h, h->stage_in, n, n, want_o[0] ? h->stage_out[0] : nullptr,
		                                   want_o[1] ? h->stage_out[1] : nullptr, nullptr, n));
		for (int i = 0; i < 2; ++i)
			if (want_o[i])
				ZH_HIP(hipMemcpyAsync(sink ? h->pin[i][0] : hosts[i], h->stage_out[i], sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
		ZH_HIP(hipStreamSynchronize(h->stream));
		if (sink)
			for (int i = 0; i < 2; ++i)
				if (want_o[i])
					sink(sink_user, i, 0, h->pin[i][0], n);
		st.total_ms = now_ms() - t_start;
		return ZEN_HIP_OK;
	}
	ZH_TRY(ensure_copy_streams(h, 3 * n_ranges + 1)); // per range: upload, kernels, download; + the call's start
	{ // Range 0 is the cheapest one (no warm-up halo in front of it): sized by it, pass 2's input buffer and the engines would
	  // grow again at range 1 -- a device-wide synchronisation, hipFree and hipMalloc in the middle of the pipeline.  Size
	  // everything for the largest range before the first copy is queued.
		size_t c2_max = 0, hops1 = 0, hops2 = 0;
		for (size_t k = 0; k < n_ranges; ++k) {
			const size_t b = k * range, e = b + range < n ? b + range : n;
			RangePlan p;
			ZH_TRY(plan_range(h, n, b, e, &p));
			c2_max = std::max(c2_max, (p.m1 - p.q2) * h->hop_p);
			hops1 = std::max(hops1, p.k1 - p.q1);
			hops2 = std::max(hops2, p.m1 - p.q2);
		}
		if (c2_max > h->cap2) {
			ZH_HIP(hipStreamSynchronize(h->stream));
			hpri_free_scratch(h);
			ZH_HIP(zh_malloc((void**)&h->in2, sizeof(float) * c2_max));
			h->cap2 = c2_max;
		}
		ZH_TRY(hpr_reserve_hops(h->eh, hops1));
		ZH_TRY(hpr_reserve_hops(h->ep, hops2));
	}
	const bool try_register = g_opt_offline_no_register.load(std::memory_order_relaxed) == 0;
	Registered reg_in, reg_out[2];
	// (sink form, round 6: the caller is HPRIOffline::process, whose clip is a vector nobody has pinned -- pinning 635 MB takes
	// 33 ms before the first byte moves, the runtime's staged copy of a pageable range costs the feeding thread about as much
	// but spread over the loop, under the kernels and the downloads; "offline_sink_register": pin it all the same)
	reg_in.take(audio_host, sizeof(float) * n, try_register && (!sink || g_opt_offline_sink_register));
	for (int i = 0; i < 2; ++i)
		reg_out[i].take(hosts[i], sizeof(float) * n, try_register);
	st.input_pinned = reg_in.pinned;
	st.outputs_pinned = (!hosts[0] || reg_out[0].pinned) && (!hosts[1] || reg_out[1].pinned);
	st.setup_ms = now_ms() - t_start;
	// With every buffer known to the runtime all copies are asynchronous and the loop below only enqueues.  A copy from /
	// to pageable memory blocks this thread until it is done: the loop then issues the upload of range k+1 BEFORE the
	// download of range k, so that the thread is never stuck behind kernels it has not fed yet.
	size_t uploaded = 0;
	auto upload_to = [&](size_t k) -> int { // input samples range k reads, beyond what is up already; event 2k
		size_t b = k * range, e = b + range < n ? b + range : n, in_b, in_e;
		ZH_TRY(zen_hip_hpri_range_halo(h, n, b, e, &in_b, &in_e));
		if (in_e > uploaded) {
			ZH_HIP(hipMemcpyAsync(h->stage_in + uploaded, audio_host + uploaded, sizeof(float) * (in_e - uploaded),
			                      hipMemcpyHostToDevice, h->s_in));
			uploaded = in_e;
		}
		ZH_HIP(hipEventRecord(h->events[2 * k], h->s_in));
		return ZEN_HIP_OK;
	};
	// Sink form: range k comes down into slot k % SINK_SLOTS of each wanted output's pinned ring; one consumer thread per
	// output waits for the range's download event and hands the slot to the caller's sink, in order.  The feeding loop does
	// not queue a download into a slot before both consumers are done with the range that was there.
	std::atomic<size_t> enq{0};                 // downloads queued so far (their events recorded)
	std::atomic<size_t> done[2] = {{0}, {0}};   // ranges each consumer has handed over
	std::atomic<int> abort_flag{0};
	std::thread consumers[2];
	int dev = 0;
	(void)hipGetDevice(&dev);
	struct JoinAll {
		std::thread (&t)[2];
		std::atomic<int>& abort_flag;
		~JoinAll()
		{
			abort_flag.store(1);
			for (auto& x : t)
				if (x.joinable())
					x.join();
		}
	} join_consumers{consumers, abort_flag};
	bool inline_sink = false; // no thread to be had: the feeding loop hands the ranges over itself
	if (sink) {
		for (int i = 0; i < 2 && !inline_sink; ++i) {
			if (!want_o[i])
				continue;
			try {
				consumers[i] = std::thread([&, i] {
					(void)hipSetDevice(dev);
					for (size_t k = 0; k < n_ranges; ++k) {
						while (enq.load(std::memory_order_acquire) <= k) {
							if (abort_flag.load(std::memory_order_relaxed))
								return;
							std::this_thread::yield();
						}
						if (hipEventSynchronize(h->events[2 * n_ranges + 1 + k]) != hipSuccess) {
							abort_flag.store(2);
							return;
						}
						const size_t b = k * range, e = b + range < n ? b + range : n;
						sink(sink_user, i, b, h->pin[i][k % SINK_SLOTS], e - b);
						done[i].store(k + 1, std::memory_order_release);
					}
				});
			}
			catch (...) {
				inline_sink = true;
			}
		}
		if (inline_sink) { // (whatever was started goes home first)
			abort_flag.store(1);
			for (auto& x : consumers)
				if (x.joinable())
					x.join();
			abort_flag.store(0);
		}
	}
	auto download = [&](size_t k) -> int {
		const size_t b = k * range, e = b + range < n ? b + range : n;
		if (sink && k >= (size_t)SINK_SLOTS && !inline_sink) { // the slot's previous range must have been handed over
			for (int i = 0; i < 2; ++i)
				while (want_o[i] && done[i].load(std::memory_order_acquire) < k - SINK_SLOTS + 1) {
					if (abort_flag.load(std::memory_order_relaxed))
						ZH_FAIL(ZEN_HIP_E_HIP, "hpri_process_sink: a consumer thread failed waiting for its range");
					std::this_thread::yield();
				}
		}
		ZH_HIP(hipStreamWaitEvent(h->s_out, h->events[2 * k + 1], 0));
		for (int i = 0; i < 2; ++i)
			if (want_o[i])
				ZH_HIP(hipMemcpyAsync(sink ? h->pin[i][k % SINK_SLOTS] : hosts[i] + b, h->stage_out[i] + b, sizeof(float) * (e - b),
				                      hipMemcpyDeviceToHost, h->s_out));
		if (sink) {
			ZH_HIP(hipEventRecord(h->events[2 * n_ranges + 1 + k], h->s_out));
			enq.store(k + 1, std::memory_order_release);
			if (inline_sink) {
				ZH_HIP(hipEventSynchronize(h->events[2 * n_ranges + 1 + k]));
				for (int i = 0; i < 2; ++i)
					if (want_o[i])
						sink(sink_user, i, b, h->pin[i][k % SINK_SLOTS], e - b);
			}
		}
		return ZEN_HIP_OK;
	};
	// the copy streams do not know the handle's stream: order them behind whatever the caller queued there (and behind an
	// earlier call's kernels, which may still read the stage buffers)
	auto feed = [&]() -> int {
		ZH_HIP(hipEventRecord(h->events[2 * n_ranges], h->stream));
		ZH_HIP(hipStreamWaitEvent(h->s_in, h->events[2 * n_ranges], 0));
		ZH_TRY(upload_to(0));
		for (size_t k = 0; k < n_ranges; ++k) {
			const size_t b = k * range, e = b + range < n ? b + range : n;
			ZH_HIP(hipStreamWaitEvent(h->stream, h->events[2 * k], 0));
			ZH_TRY(zen_hip_hpri_process_range(h, h->stage_in, n, b, e, want_o[0] ? h->stage_out[0] + b : nullptr,
			                                  want_o[1] ? h->stage_out[1] + b : nullptr));
			ZH_HIP(hipEventRecord(h->events[2 * k + 1], h->stream));
			if (k + 1 < n_ranges)
				ZH_TRY(upload_to(k + 1));
			ZH_TRY(download(k));
		}
		return ZEN_HIP_OK;
	};
	const int rc = feed();
	st.enqueue_ms = now_ms() - t_start - st.setup_ms;
	if (rc != ZEN_HIP_OK)
		abort_flag.store(1);
	// Whatever happened above, nothing may still be in flight when the caller's buffers are unregistered (end of this
	// scope) and handed back: wait for all three streams (and the consumers), then report the first failure.
	const hipError_t e_out = hipStreamSynchronize(h->s_out), e_run = hipStreamSynchronize(h->stream), e_in = hipStreamSynchronize(h->s_in);
	for (auto& x : consumers)
		if (x.joinable())
			x.join();
	ZH_TRY(rc);
	ZH_HIP(e_out);
	ZH_HIP(e_run);
	ZH_HIP(e_in);
	if (abort_flag.load() == 2)
		ZH_FAIL(ZEN_HIP_E_HIP, "hpri_process_sink: a consumer thread failed waiting for its range");
	st.total_ms = now_ms() - t_start;
	return ZEN_HIP_OK;
}
} // namespace
} // extern "C++"

int zen_hip_hpri_host_stats_get(zen_hip_hpri_t h, zen_hip_hpri_host_stats* out)
{
	if (!h || !out)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "hpri_host_stats_get: null argument");
	*out = h->stats;
	return ZEN_HIP_OK;
}

} // extern "C"
