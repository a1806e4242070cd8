// common.h -- shared host-side helpers of the gfx950 HPSS shim (error plumbing, small utilities).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/zen_hip.h"
#include "bounds.h" // ZH_CHK: the bounds checks of -DZEN_HIP_BOUNDS builds (nothing otherwise)

namespace zen_hip_impl {

void set_error(const char* fmt, ...);
// Process-wide tuning / debug switches (zen_hip_set_option).  Atomics: a host with one thread per GPU may
// flip one while another thread launches; every launch reads each switch once.
typedef std::atomic<int> opt_t;
extern opt_t g_opt_no_rt_fused;          // "no_rt_fused": single hops go through the 3-kernel path
extern opt_t g_opt_no_block_fused;       // "no_block_fused": blocks of causal hops use the 3-kernel path
extern opt_t g_opt_block_fused_minb;     // "block_fused_minb": fused block kernel built for n workgroups/CU
extern opt_t g_opt_no_istft_multi;       // "no_istft_multi": one workgroup per (frame, output) also for hard masks
extern opt_t g_opt_no_median47_neighbour; // "no_median47_neighbour": generic 47-tap kernel without its DPP block exchange
extern opt_t g_opt_median_general;       // "median_general": force the general wave kernel
extern opt_t g_opt_no_median47_dpp;      // "no_median47_dpp": 4096-bin rows / 47 taps through the generic kernel
extern opt_t g_opt_no_half_rows;         // "no_half_rows": the three-kernel path stores and filters whole magnitude rows
extern opt_t g_opt_no_direct_out;       // "no_direct_out": the fused block kernel leaves the overlap-add to finalize_kernel
extern opt_t g_opt_offline_range;       // "offline_range": samples per range of zen_hip_hpri_process's pipeline (0: default)
extern opt_t g_opt_offline_no_register; // "offline_no_register": zen_hip_hpri_process never hipHostRegisters the caller's buffers
extern opt_t g_opt_no_istft_xcd_map;    // "no_istft_xcd_map": several outputs: a grid of frames x outputs instead of same-XCD groups
extern opt_t g_opt_no_sse_block;        // "no_sse_block": blocks of frames on the SSE path through the four-launch path
extern opt_t g_opt_offline_chunk_hops; // "offline_chunk_hops": hops per chunk of the engines of HPRIOffline handles created from now on (0: sized by the device memory cap)
extern opt_t g_opt_no_istft_runs;      // "no_istft_runs": the offline small-hop pass writes Y rows and runs the synthesis + overlap-add launches
extern opt_t g_opt_istft_run_wide;     // "istft_run_wide": the same for istft_run_wide_kernel (0: default 12)
extern opt_t g_opt_istft_run;          // "istft_run": consecutive frames per wavefront of istft_run_kernel (0: default 16)
extern opt_t g_opt_no_median_tf;        // "no_median_tf": time median and frequency median + mask bits as two launches
extern opt_t g_opt_no_median_bits;      // "no_median_bits": the mask bits always come from mask_bits_kernel, never from a median kernel
extern opt_t g_opt_no_hop_lat;         // "no_hop_lat": single hops of the median path through rt_fused.hip's builds (rounds 1-4)
extern opt_t g_opt_no_sse_lat;         // "no_sse_lat": single hops of the SSE path through rt_sse.hip's two-wavefront kernels (rounds 2-4)
extern opt_t g_opt_no_rfft;            // "no_rfft": the analysis kernel runs the full complex transform on its real frames (rounds 1-4)
extern opt_t g_opt_publish_release;   // "publish_release": single hops are published with the system-scope release form (fence + release store)
extern opt_t g_opt_offline_sink_register; // "offline_sink_register": zen_hip_hpri_process_sink pins the caller's clip (hipHostRegister) like zen_hip_hpri_process does
extern opt_t g_opt_host_block_hops;    // "host_block_hops": hops per piece of zen_hip_hpr_process_host's pipeline (0: ~8 MiB of input)
extern opt_t g_opt_no_mask_bits;        // "no_mask_bits": the synthesis kernels compare H and P themselves (no mask_bits_kernel)
// Diagnostics whose results are not the reference's (timing experiments) or that only exist to cross-check a formulation
// are compiled into -DZEN_HIP_DIAG builds only (ZEN_HIP_EXTRA_FLAGS=-DZEN_HIP_DIAG python zen_amd/build.py --force); the
// shipped library refuses their names (zen_hip_set_option) and reads them as 0.
#ifdef ZEN_HIP_DIAG
#define ZH_DIAG_OPT(x) ((int)(x))
#else
#define ZH_DIAG_OPT(x) 0
#endif
extern opt_t g_opt_mask_divide;          // "mask_divide": the lean fused kernel forms its hard mask with the IEEE divide
extern opt_t g_opt_rt_fused_diag;         // "rt_fused_diag": 1 = fused kernel without its median stage, 2 = without synthesis (timing only)
extern opt_t g_opt_median47_variant;    // "median47_variant": median47_dpp_kernel build (0 default, 1 direct stores, 2/3 diagnostics)

extern std::atomic<unsigned> g_host_free_gen; // bumped by zen_hip_host_free: cached host/device alias lookups are stale

// Evaluate a HIP call; on failure record file:line + hipGetErrorString and return ZEN_HIP_E_HIP.
#define ZH_HIP(call)                                                                                  \
	do {                                                                                              \
		hipError_t e__ = (call);                                                                      \
		if (e__ != hipSuccess) {                                                                      \
			::zen_hip_impl::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call,             \
			                          hipGetErrorString(e__));                                        \
			return ZEN_HIP_E_HIP;                                                                     \
		}                                                                                             \
	} while (0)

#define ZH_TRY(expr)                \
	do {                            \
		int rc__ = (expr);          \
		if (rc__ != ZEN_HIP_OK)     \
			return rc__;            \
	} while (0)

#define ZH_FAIL(code, ...)                        \
	do {                                          \
		::zen_hip_impl::set_error(__VA_ARGS__);   \
		return (code);                            \
	} while (0)

static inline int ilog2(size_t n)
{
	int l = 0;
	while ((size_t(1) << l) < n)
		++l;
	return l;
}
static inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
static inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

// mfilt.h:78-89 / box.h:69-80 : too-big check on the caller's length, then make it odd.
int check_filter_len(int time, int frequency, int filter_len, int direction, int* odd_len);

// Host-generated tables shared bit-for-bit with the oracle (same formulas, same libm).
void make_window_sqrt_hann(float* w, size_t n);      // libzen/win.h:21-51
void make_twiddles(float* tw_interleaved, size_t n); // see oracle/zen_oracle.h (octant-symmetric)

} // namespace zen_hip_impl
