// api.hip -- runtime, memory and wrapper-level entry points of the C-ABI (include/zen_hip.h):
// zen_hip_init, device/mapped memory, FFTC2CWrapperGPU, MedianFilterGPU, BoxFilterGPU.
#include "common.h"
#include "memguard.h"
#include "filters.h"
#include "stft.h"

#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace zen_hip_impl {

static thread_local char g_err[512] = "";
opt_t g_opt_median_general{0};
opt_t g_opt_no_rt_fused{0};
opt_t g_opt_no_block_fused{0};
opt_t g_opt_no_median47_neighbour{0};
opt_t g_opt_no_istft_multi{0};
opt_t g_opt_no_median47_dpp{0};
opt_t g_opt_median47_variant{0};
opt_t g_opt_rt_fused_diag{0};
opt_t g_opt_mask_divide{0};
opt_t g_opt_no_half_rows{0};
opt_t g_opt_no_mask_bits{0};
opt_t g_opt_no_median_bits{0};
opt_t g_opt_no_direct_out{0};
opt_t g_opt_no_istft_xcd_map{0};
opt_t g_opt_no_sse_block{0};
opt_t g_opt_no_median_tf{0};
opt_t g_opt_no_istft_runs{0};
opt_t g_opt_offline_chunk_hops{0};
opt_t g_opt_istft_run{0};
opt_t g_opt_istft_run_wide{0};
opt_t g_opt_offline_range{0};
opt_t g_opt_offline_no_register{0};
opt_t g_opt_no_rfft{0};
opt_t g_opt_no_sse_lat{0};
opt_t g_opt_no_hop_lat{0};
opt_t g_opt_host_block_hops{0};
opt_t g_opt_offline_sink_register{[] { const char* v = getenv("ZEN_HIP_SINK_REGISTER"); return (v && *v && *v != '0') ? 1 : 0; }()};
// How a single hop is published to its host-mapped buffer (rt_fused.hip publish_ready).  Default since round 6: the form the
// memory model backs -- system-scope release fence + release store -- as the reference's synchronising thrust::copy made
// host_out readable (hps.cu:341-363).  The light form (write-through sample stores + a relaxed flag: ~1-1.5 us less per
// hop, and it leans on the order of posted writes on the way to host memory) is opt-in: ZEN_HIP_PUBLISH_LIGHT=1, or
// zen_hip_set_option("publish_release", 0).  (ZEN_HIP_PUBLISH_RELEASE=0 / 1, the switch of round 5, is still read.)
opt_t g_opt_publish_release{[] {
	const char* r = getenv("ZEN_HIP_PUBLISH_RELEASE");
	if (r && *r)
		return *r != '0' ? 1 : 0;
	const char* l = getenv("ZEN_HIP_PUBLISH_LIGHT");
	return (l && *l && *l != '0') ? 0 : 1;
}()};
std::atomic<unsigned> g_host_free_gen{0};

void set_error(const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

int check_filter_len(int time, int frequency, int filter_len, int direction, int* odd_len)
{
	if (time <= 0 || frequency <= 0 || filter_len <= 0)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "filter: non-positive dimension (time %d, frequency %d, len %d)", time,
		        frequency, filter_len);
	if (direction != ZEN_HIP_TIME_CAUSAL && direction != ZEN_HIP_TIME_ANTICAUSAL
	    && direction != ZEN_HIP_FREQUENCY)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "filter: unknown direction %d", direction);
	// mfilt.h:78-86 / box.h:69-77 : compared before the length is made odd
	if ((direction != ZEN_HIP_FREQUENCY && filter_len > time)
	    || (direction == ZEN_HIP_FREQUENCY && filter_len > frequency))
		ZH_FAIL(ZEN_HIP_E_FILTER_TOO_BIG, "median filter bigger than matrix dimension");
	*odd_len = filter_len + (1 - (filter_len % 2)); // mfilt.h:89
	return ZEN_HIP_OK;
}

void make_window_sqrt_hann(float* w, size_t n)
{
	const float PI = 3.14159265359F; // win.h:13
	const float N = (float)n;        // periodic: divide by the window size (win.h:31-34)
	for (size_t i = 0; i < n; ++i)
		w[i] = sqrtf(0.5F * (1.0F - cosf(2.0F * PI * (float)i / N)));
}

void make_twiddles(float* tw, size_t n)
{
	// tw[j] = exp(-2 pi i j / n), j < n/2.  First octant from double libm rounded to float; the rest by
	// symmetry, so tw[j + n/4] == -i * tw[j] exactly.  Same construction as oracle zo_twiddles().
	const size_t half = n / 2;
	if (half == 0)
		return;
	if (n < 8) {
		tw[0] = 1.0F;
		tw[1] = 0.0F;
		if (n == 4) {
			tw[2] = 0.0F;
			tw[3] = -1.0F;
		}
		return;
	}
	const size_t Q = n / 4, O = n / 8;
	std::vector<float> c(Q + 1), s(Q + 1);
	const double two_pi = 6.283185307179586476925286766559;
	for (size_t j = 0; j <= O; ++j) {
		const double th = two_pi * (double)j / (double)n;
		c[j] = (float)cos(th);
		s[j] = (float)sin(th);
	}
	c[0] = 1.0F;
	s[0] = 0.0F;
	for (size_t j = O + 1; j <= Q; ++j) {
		c[j] = s[Q - j];
		s[j] = c[Q - j];
	}
	for (size_t j = 0; j < half; ++j) {
		const float cj = (j <= Q) ? c[j] : -s[j - Q];
		const float sj = (j <= Q) ? s[j] : c[j - Q];
		tw[2 * j] = cj;
		tw[2 * j + 1] = -sj;
	}
}

} // namespace zen_hip_impl

using namespace zen_hip_impl;

struct zen_hip_fft {
	size_t nfft;
	int log2n;
	float2* tw;
	float2* xch = nullptr; // nfft 32768 only: exchange buffer of the two-step transform (fft_big.hip), grown on demand
	size_t xch_batch = 0;
};

struct zen_hip_filter {
	int time, frequency, len, direction;
	bool is_box;
	bool assume_nonneg; // zen_hip_mfilt_assume_nonneg
};

extern "C" {

const char* zen_hip_last_error(void) { return g_err; }
const char* zen_hip_version(void) { return "zen-mi355x 0.1 (gfx950)"; }

int zen_hip_init(int device)
{
	// No process-wide side effects by default.  Two latency switches a realtime host may opt into, both BEFORE its first
	// HIP call (documented in INTEGRATION.md; tools/rt_latency.cpp and the bench's per-hop leg do):
	//   HIP_FORCE_DEV_KERNARG=1 in the environment (a ROCm runtime switch, read when the runtime starts): kernel
	//     arguments in device memory instead of host memory; the first instructions of a launch wait for them, and a
	//     single-hop call is short enough to notice (0.7-1 us of 17-21 us).
	//   ZEN_HIP_SCHEDULE=spin: hipDeviceScheduleSpin for this device -- every synchronising HIP call of the process
	//     then polls (one busy core per synchronising thread) instead of sleeping on an interrupt.  The mapped-memory
	//     copy_* path polls its own sequence word and does not need it; the other copy_* paths gain the wake-up time.
	int n = 0;
	ZH_HIP(hipGetDeviceCount(&n));
	if (device < 0 || device >= n)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "zen_hip_init: device %d of %d", device, n);
	// core.cu:4-6 sets cudaDeviceMapHost before a context exists; on ROCm mapped host memory needs
	// no device flag, the call is kept for symmetry and its "already active" status is ignored.
	const char* sched = getenv("ZEN_HIP_SCHEDULE");
	const unsigned spin = (sched && !strcmp(sched, "spin")) ? (unsigned)hipDeviceScheduleSpin : 0u;
	ZH_HIP(hipSetDevice(device));
	(void)hipSetDeviceFlags(hipDeviceMapHost | spin);
	(void)hipGetLastError();
	return ZEN_HIP_OK;
}

int zen_hip_device_count(int* n)
{
	if (!n)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "zen_hip_device_count: null argument");
	ZH_HIP(hipGetDeviceCount(n));
	return ZEN_HIP_OK;
}

int zen_hip_set_option(const char* name, int value)
{
	static const struct {
		const char* name;
		opt_t* var;
	} table[] = {{"median_general", &g_opt_median_general},
	             {"no_rt_fused", &g_opt_no_rt_fused},
	             {"no_block_fused", &g_opt_no_block_fused},
	             {"no_median47_neighbour", &g_opt_no_median47_neighbour},
	             {"no_istft_multi", &g_opt_no_istft_multi},
	             {"block_fused_minb", &g_opt_block_fused_minb},
	             {"no_median47_dpp", &g_opt_no_median47_dpp},
	             {"median47_variant", &g_opt_median47_variant},
#ifdef ZEN_HIP_DIAG
	             {"rt_fused_diag", &g_opt_rt_fused_diag},
	             {"mask_divide", &g_opt_mask_divide},
#endif
	             {"no_half_rows", &g_opt_no_half_rows},
	             {"no_mask_bits", &g_opt_no_mask_bits},
	             {"no_median_bits", &g_opt_no_median_bits},
	             {"no_direct_out", &g_opt_no_direct_out},
	             {"no_istft_xcd_map", &g_opt_no_istft_xcd_map},
	             {"no_sse_block", &g_opt_no_sse_block},
	             {"no_median_tf", &g_opt_no_median_tf},
	             {"no_istft_runs", &g_opt_no_istft_runs},
	             {"offline_chunk_hops", &g_opt_offline_chunk_hops},
	             {"istft_run", &g_opt_istft_run},
	             {"istft_run_wide", &g_opt_istft_run_wide},
	             {"offline_range", &g_opt_offline_range},
	             {"offline_no_register", &g_opt_offline_no_register},
	             {"publish_release", &g_opt_publish_release},
	             {"no_rfft", &g_opt_no_rfft},
	             {"no_sse_lat", &g_opt_no_sse_lat},
	             {"no_hop_lat", &g_opt_no_hop_lat},
	             {"host_block_hops", &g_opt_host_block_hops},
	             {"offline_sink_register", &g_opt_offline_sink_register}};
#ifndef ZEN_HIP_DIAG
	if (name && (!strcmp(name, "rt_fused_diag") || !strcmp(name, "mask_divide") || (!strcmp(name, "median47_variant") && value > 1)))
		ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "zen_hip_set_option: '%s' = %d is a diagnostic of -DZEN_HIP_DIAG builds", name, value);
#endif
	for (const auto& t : table) {
		if (name && !strcmp(name, t.name)) {
			t.var->store(value, std::memory_order_relaxed);
			return ZEN_HIP_OK;
		}
	}
	ZH_FAIL(ZEN_HIP_E_BAD_ARG, "zen_hip_set_option: unknown option '%s'", name ? name : "(null)");
}

int zen_hip_device_name(char* buf, size_t n)
{
	int dev = 0;
	ZH_HIP(hipGetDevice(&dev));
	hipDeviceProp_t p;
	ZH_HIP(hipGetDeviceProperties(&p, dev));
	snprintf(buf, n, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
	return ZEN_HIP_OK;
}

int zen_hip_synchronize(void* stream)
{
	ZH_HIP(hipStreamSynchronize((hipStream_t)stream));
	return ZEN_HIP_OK;
}

int zen_hip_stream_create(void** stream)
{
	if (!stream)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "stream_create: null argument");
	hipStream_t st = nullptr;
	ZH_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	*stream = (void*)st;
	return ZEN_HIP_OK;
}
int zen_hip_stream_destroy(void* stream)
{
	if (stream)
		ZH_HIP(hipStreamDestroy((hipStream_t)stream));
	return ZEN_HIP_OK;
}

int zen_hip_event_create(void** event)
{
	if (!event)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "event_create: null argument");
	hipEvent_t e = nullptr;
	ZH_HIP(hipEventCreate(&e));
	*event = (void*)e;
	return ZEN_HIP_OK;
}
int zen_hip_event_record(void* event, void* stream)
{
	if (!event)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "event_record: null event");
	ZH_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
	return ZEN_HIP_OK;
}
int zen_hip_event_elapsed_ms(void* start, void* stop, float* ms)
{
	if (!start || !stop || !ms)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "event_elapsed_ms: null argument");
	ZH_HIP(hipEventSynchronize((hipEvent_t)stop));
	ZH_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
	return ZEN_HIP_OK;
}
int zen_hip_event_destroy(void* event)
{
	if (event)
		ZH_HIP(hipEventDestroy((hipEvent_t)event));
	return ZEN_HIP_OK;
}

int zen_hip_malloc(void** dev, size_t bytes)
{
	ZH_HIP(zh_malloc(dev, bytes ? bytes : 1));
	return ZEN_HIP_OK;
}
int zen_hip_free(void* dev)
{
	if (dev)
		ZH_HIP(zh_free(dev));
	return ZEN_HIP_OK;
}
int zen_hip_memset(void* dev, int value, size_t bytes, void* stream)
{
	ZH_HIP(hipMemsetAsync(dev, value, bytes, (hipStream_t)stream));
	return ZEN_HIP_OK;
}
int zen_hip_memcpy_h2d(void* dev, const void* host, size_t bytes)
{
	ZH_HIP(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
	return ZEN_HIP_OK;
}
int zen_hip_memcpy_d2h(void* host, const void* dev, size_t bytes)
{
	ZH_HIP(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
	return ZEN_HIP_OK;
}
int zen_hip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream)
{
	ZH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
	return ZEN_HIP_OK;
}

int zen_hip_memcpy_h2d_async(void* dev, const void* host, size_t bytes, void* stream)
{
	ZH_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
	return ZEN_HIP_OK;
}
int zen_hip_memcpy_d2h_async(void* host, const void* dev, size_t bytes, void* stream)
{
	ZH_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
	return ZEN_HIP_OK;
}

// Input buffers that live in HBM (see zen_hip_host_alloc_mapped): remembered so that zen_hip_host_free can
// tell them from pinned host memory.
static std::mutex g_bar_mu;
static std::vector<void*> g_bar_bufs;

int zen_hip_host_alloc_mapped(size_t bytes, int write_combined, void** host, void** dev)
{
	if (!host || !dev)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "host_alloc_mapped: null argument");
	// io.h:24-66 : cudaHostAllocMapped | cudaHostAllocPortable (| WriteCombined for host_in).
	// write_combined marks the buffer the host only ever WRITES (IOGPU::host_in, io.h:33-35).  On a large-BAR
	// system the device's own memory can be written by the host directly, so that buffer is placed in HBM
	// (fine-grained): the host pushes a hop with posted writes (0.5 us per 4 KB measured) and the kernel reads
	// local memory, instead of pulling the hop over the host link (4-5 us of the 1024-hop call, measured).
	// Same contract for the caller: write through `host`, hand `dev` to the engine.
	// ZEN_HIP_INPUT_IN_HOST_MEMORY=1 keeps the reference's placement.
	if (write_combined && !getenv("ZEN_HIP_INPUT_IN_HOST_MEMORY")) {
		int devid = 0, large_bar = 0;
		if (hipGetDevice(&devid) == hipSuccess
		    && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, devid) == hipSuccess && large_bar) {
			void* p = nullptr;
			if (zh_ext_malloc(&p, bytes ? bytes : 1, hipDeviceMallocFinegrained) == hipSuccess) {
				std::lock_guard<std::mutex> lk(g_bar_mu);
				g_bar_bufs.push_back(p);
				*host = p;
				*dev = p;
				return ZEN_HIP_OK;
			}
		}
		(void)hipGetLastError();
	}
	unsigned flags = hipHostMallocMapped | hipHostMallocPortable;
	if (write_combined)
		flags |= hipHostMallocWriteCombined;
	ZH_HIP(zh_host_malloc(host, bytes ? bytes : 1, flags));
	ZH_HIP(zh_host_device_pointer(dev, *host));
	return ZEN_HIP_OK;
}
int zen_hip_host_free(void* host)
{
	if (!host)
		return ZEN_HIP_OK;
	g_host_free_gen.fetch_add(1, std::memory_order_relaxed); // engines drop what they remember about host buffers (hpr.hip host_alias_of)
	{
		std::lock_guard<std::mutex> lk(g_bar_mu);
		for (size_t i = 0; i < g_bar_bufs.size(); ++i) {
			if (g_bar_bufs[i] == host) {
				g_bar_bufs.erase(g_bar_bufs.begin() + (long)i);
				ZH_HIP(zh_free(host));
				return ZEN_HIP_OK;
			}
		}
	}
	ZH_HIP(zh_host_free(host));
	return ZEN_HIP_OK;
}

// ---- FFTC2CWrapperGPU ---------------------------------------------------------------------------
int zen_hip_fft_create(size_t nfft, zen_hip_fft_t* h)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "fft_create: null handle");
	if (!is_pow2(nfft))
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "fft_create: nfft %zu is not a power of two", nfft);
	if (nfft < 32 || nfft > 32768) // (fftw.bench.cu:231-252 sweeps 256..32768; the engine itself stops at 16384)
		ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "fft_create: nfft %zu outside 32..32768", nfft);
	zen_hip_fft* f = new zen_hip_fft;
	f->nfft = nfft;
	f->log2n = ilog2(nfft);
	f->tw = nullptr;
	std::vector<float> tw(nfft);
	make_twiddles(tw.data(), nfft);
	if (zh_malloc((void**)&f->tw, sizeof(float) * nfft) != hipSuccess
	    || hipMemcpy(f->tw, tw.data(), sizeof(float) * nfft, hipMemcpyHostToDevice) != hipSuccess) {
		(void)zh_free(f->tw);
		(void)hipGetLastError();
		delete f;
		ZH_FAIL(ZEN_HIP_E_HIP, "fft_create: device allocation failed");
	}
	*h = f;
	return ZEN_HIP_OK;
}

int zen_hip_fft_exec_batched(zen_hip_fft_t h, float* inout_dev, size_t batch, int inverse, void* stream)
{
	if (!h || !inout_dev)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "fft_exec: null argument");
	if (h->log2n > 14) { // one frame does not fit a workgroup's LDS: two steps through a scratch buffer
		if (batch > h->xch_batch) {
			// the buffer may still be in use by an earlier call on ANOTHER stream: wait for the device, not for `stream`
			// (a 32768-point handle has one scratch buffer: its calls must not overlap across streams, zen_hip.h)
			ZH_HIP(hipDeviceSynchronize());
			(void)zh_free(h->xch);
			h->xch = nullptr;
			h->xch_batch = 0;
			ZH_HIP(zh_malloc((void**)&h->xch, sizeof(float2) * h->nfft * batch));
			h->xch_batch = batch;
		}
		return launch_fft_big(h->log2n, (float2*)inout_dev, h->xch, h->tw, batch, inverse, (hipStream_t)stream);
	}
	return launch_fft(h->log2n, (float2*)inout_dev, h->tw, batch, inverse, (hipStream_t)stream);
}

int zen_hip_fft_exec(zen_hip_fft_t h, float* inout_dev, int inverse, void* stream)
{
	return zen_hip_fft_exec_batched(h, inout_dev, 1, inverse, stream);
}

int zen_hip_fft_destroy(zen_hip_fft_t h)
{
	if (h) {
		(void)zh_free(h->tw);
		(void)zh_free(h->xch);
		delete h;
	}
	return ZEN_HIP_OK;
}

// ---- MedianFilterGPU / BoxFilterGPU -------------------------------------------------------------
static int filter_create(int time, int frequency, int filter_len, int direction, bool is_box,
                         zen_hip_filter** h)
{
	if (!h)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "filter_create: null handle");
	int odd = 0;
	ZH_TRY(check_filter_len(time, frequency, filter_len, direction, &odd));
	zen_hip_filter* f = new zen_hip_filter;
	f->time = time;
	f->frequency = frequency;
	f->len = odd;
	f->direction = direction;
	f->is_box = is_box;
	f->assume_nonneg = false;
	*h = f;
	return ZEN_HIP_OK;
}

static int filter_run(zen_hip_filter* h, const float* src, float* dst, void* stream)
{
	if (!h || !src || !dst)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "filter_run: null argument");
	FilterArgs a;
	memset(&a, 0, sizeof(a));
	a.src = src;
	a.dst = dst;
	a.n_streams = 1;
	a.cols = h->frequency;
	a.ring_rows = h->time;
	a.first_row = 0;
	a.n_out_rows = h->time;
	a.clamp_lo = 0;
	a.clamp_hi = h->time - 1;
	a.len = h->len;
	a.direction = h->direction;
	a.nonneg = (!h->is_box && h->assume_nonneg) ? 1 : 0; // zen_hip_mfilt_assume_nonneg: this handle's promise (magnitudes)
	return h->is_box ? launch_box(a, (hipStream_t)stream) : launch_median(a, (hipStream_t)stream);
}

int zen_hip_mfilt_create(int time, int frequency, int filter_len, int direction, int copy_bord,
                         zen_hip_mfilt_t* h)
{
	(void)copy_bord; // CPU semantics (replicate border) always; see header
	return filter_create(time, frequency, filter_len, direction, false, h);
}
int zen_hip_mfilt_run(zen_hip_mfilt_t h, const float* src_dev, float* dst_dev, void* stream)
{
	return filter_run(h, src_dev, dst_dev, stream);
}
int zen_hip_mfilt_assume_nonneg(zen_hip_mfilt_t h, int nonneg)
{
	if (!h || h->is_box)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "mfilt_assume_nonneg: not a median filter handle");
	h->assume_nonneg = nonneg != 0;
	return ZEN_HIP_OK;
}
int zen_hip_mfilt_destroy(zen_hip_mfilt_t h)
{
	delete h;
	return ZEN_HIP_OK;
}

int zen_hip_box_create(int time, int frequency, int filter_len, int direction, zen_hip_box_t* h)
{
	return filter_create(time, frequency, filter_len, direction, true, h);
}
int zen_hip_box_run(zen_hip_box_t h, const float* src_dev, float* dst_dev, void* stream)
{
	return filter_run(h, src_dev, dst_dev, stream);
}
int zen_hip_box_destroy(zen_hip_box_t h)
{
	delete h;
	return ZEN_HIP_OK;
}

} // extern "C"
