/*
 * zen_hip.h -- C-ABI of the MI355X (gfx950) HPSS engine: the drop-in boundary for Zen's GPU backend.
 *
 * Each entry point names the reference interface (sevagh/Zen, file:line) it replaces.  The reference
 * reaches its GPU through three thin C++ wrappers (libzen/fftw.h, libzen/mfilt.h, libzen/box.h), the
 * zero-copy buffer class (libzen/libzen/io.h) and the algorithm object HPR<Backend::GPU>
 * (libzen/hps.h:152-322, libzen/hps.cu:429-652).  A maintainer re-points those at this library; the
 * C++ that does so is in zen_amd/libzen/ and the binding stubs are shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: opaque handles, raw device/host pointers, sizes; no C++ or torch types.
 *   - every function returns 0 (ZEN_HIP_OK) or a ZEN_HIP_E_* code; zen_hip_last_error() gives the text.
 *     Codes marked [ZgException] are the conditions on which the reference throws zen::ZgException.
 *   - work is enqueued on the handle's stream (default: the null stream) and is asynchronous unless the
 *     function says "synchronises".  `stream` arguments are a hipStream_t passed as void*.
 *   - filter semantics are those of the reference CPU (IPP) path, the parity target: centred odd mask,
 *     replicate border (libzen/mfilt.h:270-342, libzen/box.h:217-288).  `copy_bord` is accepted and, as
 *     on the reference CPU backend (mfilt.h:289), has no effect.
 *   - matrices are `time` rows x `frequency` columns, row-major, frequency contiguous, tightly packed
 *     (mfilt.h:76-79: "expect 1D linear memory layout e.g. i*y + j").
 *   - arithmetic is IEEE binary32 without FMA contraction, bit-identical to oracle/zen_oracle.c.
 */
#ifndef ZEN_HIP_H
#define ZEN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
	ZEN_HIP_OK = 0,
	ZEN_HIP_E_FILTER_TOO_BIG = 1,     /* [ZgException] mfilt.h:85 / box.h:76 */
	ZEN_HIP_E_BAD_ARG = 2,
	ZEN_HIP_E_HOPS_NOT_DIVISIBLE = 3, /* [ZgException] hps.cu:33-36 */
	ZEN_HIP_E_HIP = 4,                /* a HIP runtime call failed (reference: std::exit, io.h:37-66) */
	ZEN_HIP_E_UNSUPPORTED = 5         /* size outside what the gfx950 kernels cover (engine: nfft 32..16384; FFT wrapper: ..32768) */
};

/* libzen/mfilt.h:27-31  enum MedianFilterDirection */
enum { ZEN_HIP_TIME_CAUSAL = 0, ZEN_HIP_TIME_ANTICAUSAL = 1, ZEN_HIP_FREQUENCY = 2 };
/* libzen/libzen/hps.h:25-27 */
enum { ZEN_HIP_OUTPUT_HARMONIC = 1, ZEN_HIP_OUTPUT_PERCUSSIVE = 2, ZEN_HIP_OUTPUT_RESIDUAL = 4 };

typedef struct zen_hip_fft* zen_hip_fft_t;
typedef struct zen_hip_filter* zen_hip_mfilt_t;
typedef struct zen_hip_filter* zen_hip_box_t;
typedef struct zen_hip_hpr* zen_hip_hpr_t;
typedef struct zen_hip_hpri* zen_hip_hpri_t;

/* ---------------------------------------------------------------------------------------------
 * runtime   (replaces libzen/core.cu:4-6, the cudaSetDeviceFlags(cudaDeviceMapHost) constructor)
 * ------------------------------------------------------------------------------------------- */
int zen_hip_init(int device);          /* hipSetDevice + mapped-host flag; idempotent */
int zen_hip_device_count(int* n);      /* GPUs visible to this process (one process per GPU: `zen batch --gpus`) */
const char* zen_hip_last_error(void);  /* thread-local text of the last failure */
const char* zen_hip_version(void);
int zen_hip_device_name(char* buf, size_t n);
int zen_hip_synchronize(void* stream); /* hipStreamSynchronize; synchronises */
/* a hipStream_t for the `stream` arguments and *_set_stream (a host without HIP headers: one stream per host thread,
 * the reference's intended --default-stream-per-thread, CMakeLists.txt:43); non-blocking with respect to the null stream */
int zen_hip_stream_create(void** stream);
int zen_hip_stream_destroy(void* stream);
/* hipEvent_t timing on a stream (the harness's clock for a single launch: bench.py, tools/bench_median.py; the
 * reference times with std::chrono around synchronising thrust calls).  elapsed_ms synchronises on `stop`. */
int zen_hip_event_create(void** event);
int zen_hip_event_record(void* event, void* stream);
int zen_hip_event_elapsed_ms(void* start, void* stop, float* ms);
int zen_hip_event_destroy(void* event);
/* process-wide tuning switches (no reference counterpart; atomics, may be set from any thread).  Every one of them
 * selects between implementations with IDENTICAL results (the tests run the alternatives against each other and against
 * the oracle); none changes what a call returns.
 * "median_general" = 1 forces the general wave-cooperative median kernel even where the sorting-network
 * fast paths apply; "no_rt_fused" = 1 sends causal calls through the three-kernel path instead of the fused
 * kernel of rt_fused.hip, "no_block_fused" = 1 does so only for calls of more than one hop;
 * "block_fused_minb" = 1..3 picks the occupancy the fused block kernel is compiled for; "no_istft_multi" = 1
 * synthesises hard-mask outputs in separate workgroups instead of one per frame; "no_median47_dpp" = 1 sends
 * 47-tap frequency masks on 4096-bin rows through the generic sorting-network kernel instead of
 * median47_dpp_kernel, "no_median47_neighbour" = 1 additionally switches off that kernel's DPP exchange of
 * sorted blocks; "no_half_rows" = 1 makes the three-kernel path store and filter whole magnitude rows instead of the
 * non-redundant half (bins 0..nfft/2); "no_mask_bits" = 1: the synthesis kernels of blocks of frames compare H and P
 * themselves instead of loading two mask bits per bin, "no_median_bits" = 1: those bits always come from a launch of
 * their own, never from the frequency-direction median kernel; "no_direct_out" = 1: the fused block kernel
 * of the headline configuration leaves the overlap-add to a launch of its own; "median47_variant" = 1 lets
 * median47_dpp_kernel store results without the LDS transpose; "offline_range" / "offline_no_register": see
 * zen_hip_hpri_process; "no_istft_runs" = 1: the passes of HPRIOffline with hard masks write synthesis rows and add the
 * overlapping halves in a launch of its own instead of walking runs of frames with the carry in registers (2: only the
 * large-hop pass does), "istft_run" / "istft_run_wide" = frames per run of the two kernels (0: the library's choice);
 * "offline_chunk_hops" = n: HPRIOffline handles created from now on process at most n hops per launch in either pass
 * (a bound on their device buffers; 0: sized by the device memory cap).  "no_sse_block", "no_median_tf",
 * "no_istft_xcd_map": the launch-per-stage alternatives of the fused kernels of round 4.
 * "publish_release" = 1 (default: the environment variable ZEN_HIP_PUBLISH_RELEASE, else 0): the single-hop kernels
 * publish a finished hop to its host-mapped buffer with a system-scope release fence + release store instead of
 * write-through sample stores followed by a relaxed flag store.  The default form relies on gfx950's write-through
 * system-scope stores and on posted writes reaching host memory in order; set this on a host where either is in doubt
 * (PCIe relaxed ordering).  Costs about a microsecond per hop.
 * "no_rfft" = 1: the analysis kernels of blocks of frames run the full complex transform on their real frames (rounds
 * 1-4) instead of the Hermitian half; "no_sse_lat" = 1: single hops of the causal SSE path run the two-wavefront kernels
 * of rounds 2-4 (rt_sse.hip) instead of the layout that spreads the frame over all four SIMDs of a CU (rt_sse_lat.hip).
 * "no_hop_lat" = 1: the same for single hops of the median path (rt_fused.hip's single-hop builds instead of rt_hop_lat.hip).
 * Timing diagnostics whose outputs are NOT the reference's ("median47_variant" 2..4, "rt_fused_diag") and the
 * divide-based cross-check of the hard masks ("mask_divide") exist in -DZEN_HIP_DIAG builds of the library only; the
 * shipped build answers ZEN_HIP_E_UNSUPPORTED. */
int zen_hip_set_option(const char* name, int value);

/* Memory checking (no reference counterpart in the API: the reference runs its tests under cuda-memcheck,
 * libzen/CMakeLists.txt:56-73).  With ZEN_HIP_REDZONE=<bytes> in the environment before the first allocation, every
 * device and mapped-host allocation of the library (zen_hip_malloc and zen_hip_host_alloc_mapped included) carries that
 * many bytes of NaN-patterned red zone on either side (ZEN_HIP_POISON=1: its interior starts as NaNs too);
 * zen_hip_memcheck synchronises the device, compares the zones of every live allocation and returns the cumulative
 * findings (every free checks as well).  In -DZEN_HIP_BOUNDS builds of the library the kernels additionally look up
 * every instrumented load / store in a table of the live allocations: `bounds_violations` counts the misses
 * (ZEN_HIP_BOUNDS_TRAP=1: the offending wavefront traps).  Without either, the call reports zeros. */
typedef struct zen_hip_memcheck_report {
	unsigned long long redzone_bytes;       /* 0: red zones are off */
	unsigned long long allocations;         /* allocations made so far */
	unsigned long long live_allocations;    /* tracked allocations alive now */
	unsigned long long corrupt_words;       /* 4-byte words of red zone found overwritten, cumulative */
	unsigned long long corrupt_allocations; /* zone sides they belonged to */
	unsigned long long bounds_violations;   /* out-of-bounds accesses the bounds build recorded, cumulative */
	int bounds_build;                       /* 1: this library was built with -DZEN_HIP_BOUNDS */
	char first_message[256];                /* the first red-zone finding (else the first bounds violation), as text */
	char first_violation[160];              /* bounds build: the first out-of-bounds access: bytes, address, source file and line */
} zen_hip_memcheck_report;
int zen_hip_memcheck(zen_hip_memcheck_report* out);
/* test hook: one thread stores `value` at dev + byte_offset (through the instrumented store path) and the call synchronises */
int zen_hip_debug_poke(void* dev, long long byte_offset, unsigned value);

/* device memory + copies: what thrust::device_vector / thrust::copy are to the reference
 * (core.h:26-27; used by every wrapper and test). */
int zen_hip_malloc(void** dev, size_t bytes);
int zen_hip_free(void* dev);
int zen_hip_memset(void* dev, int value, size_t bytes, void* stream);
int zen_hip_memcpy_h2d(void* dev, const void* host, size_t bytes); /* synchronises */
int zen_hip_memcpy_d2h(void* host, const void* dev, size_t bytes); /* synchronises */
int zen_hip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
/* on a stream: asynchronous where the host range is pinned (zen_hip_host_alloc_mapped), else returns when the bytes moved */
int zen_hip_memcpy_h2d_async(void* dev, const void* host, size_t bytes, void* stream);
int zen_hip_memcpy_d2h_async(void* host, const void* dev, size_t bytes, void* stream);

/* zen::io::IOGPU (libzen/libzen/io.h:16-81): pinned, mapped, portable host buffer and its device
 * alias; write_combined != 0 for host_in (io.h:33-35), the buffer the host only writes: on a large-BAR system
 * that one is placed in device memory the host can write directly (*host == *dev), so the kernel reads the hop
 * locally instead of pulling it over the host link (ZEN_HIP_INPUT_IN_HOST_MEMORY=1: always pinned host memory). */
int zen_hip_host_alloc_mapped(size_t bytes, int write_combined, void** host, void** dev);
int zen_hip_host_free(void* host);

/* ---------------------------------------------------------------------------------------------
 * FFTC2CWrapperGPU   (libzen/fftw.h:20-49)
 * In-place unnormalised complex DFT of `nfft` points (interleaved re,im floats), batch of 1 as in
 * cufftPlan1d(&plan, nfft, CUFFT_C2C, 1) (fftw.h:32).  inverse != 0 <=> CUFFT_INVERSE (fftw.h:40-43).
 * zen_hip_fft_exec_batched transforms `batch` consecutive nfft-point rows (the offline STFT).
 * nfft <= 16384 runs in place and a handle may be used from several streams at once; a 32768-point handle owns one
 * scratch buffer (two kernels exchange through it): its calls must be ordered on one stream, or otherwise not overlap.
 * ------------------------------------------------------------------------------------------- */
int zen_hip_fft_create(size_t nfft, zen_hip_fft_t* h);
int zen_hip_fft_exec(zen_hip_fft_t h, float* inout_dev, int inverse, void* stream);
int zen_hip_fft_exec_batched(zen_hip_fft_t h, float* inout_dev, size_t batch, int inverse, void* stream);
int zen_hip_fft_destroy(zen_hip_fft_t h);

/* ---------------------------------------------------------------------------------------------
 * MedianFilterGPU   (libzen/mfilt.h:33-268; ctor :61-66, filter() :227-267)
 * filter_len > the filtered dimension => ZEN_HIP_E_FILTER_TOO_BIG (mfilt.h:78-86), checked before
 * the length is made odd (mfilt.h:89).  src and dst must not alias (reference: distinct vectors).
 * Every length the reference accepts is accepted (mfilt.h:296-305: any filter_len <= the dimension): sorting-network /
 * block-merge kernels up to 63 taps and for the engine's long masks, the general sliding-window kernels (csrc/median.hip)
 * for every other odd length -- register windows up to 2047 taps, the sorted window in LDS up to 38 400 taps, in device
 * memory beyond (that launch allocates a scratch buffer and synchronises the stream before it returns).
 * ------------------------------------------------------------------------------------------- */
int zen_hip_mfilt_create(int time, int frequency, int filter_len, int direction, int copy_bord,
                         zen_hip_mfilt_t* h);
int zen_hip_mfilt_run(zen_hip_mfilt_t h, const float* src_dev, float* dst_dev, void* stream);
/* Per-handle promise (no reference counterpart, default off): every sample this handle will be given is >= +0 -- a
 * magnitude matrix -- so its kernels may order by the raw bit patterns as the engine's own launches do.  A handle that
 * breaks the promise gets wrong medians; no other handle is affected.  The 47-tap kernel on 4096-bin rows (BASELINE's
 * median metric) does not need it: it looks at the sign bits of every row it stages and re-keys only rows that hold a
 * negative sample. */
int zen_hip_mfilt_assume_nonneg(zen_hip_mfilt_t h, int nonneg);
int zen_hip_mfilt_destroy(zen_hip_mfilt_t h);

/* BoxFilterGPU   (libzen/box.h:30-215; ctor :55-58, filter() :182-214): mean over the mask. */
int zen_hip_box_create(int time, int frequency, int filter_len, int direction, zen_hip_box_t* h);
int zen_hip_box_run(zen_hip_box_t h, const float* src_dev, float* dst_dev, void* stream);
int zen_hip_box_destroy(zen_hip_box_t h);

/* ---------------------------------------------------------------------------------------------
 * HPR<Backend::GPU>   (libzen/hps.h:152-322) and what HPRRealtime<GPU> forwards to it
 * (libzen/hps.cu:282-427).
 *
 * The engine is a chunked streaming engine: one call takes n_hops >= 1 consecutive hops of
 * `n_streams` independent streams.  n_hops == 1 is the reference's process_next_hop; larger blocks give
 * bit-identical samples (same arithmetic per frame) at far higher throughput.  State carried between
 * calls: the previous hop of input, a ring of the last stft_width-1 spectra/magnitudes, and the
 * overlap-add carry per enabled output.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
	size_t hop, nwin, nfft, stft_width; /* hps.h:222-230 */
	int l_harm, l_perc, lag;            /* hps.h:227-229, :265-268 */
	int time_len, freq_len;             /* mask lengths after mfilt.h:89 made them odd */
	float cola_factor;                  /* hps.h:270-274 */
	size_t n_streams, max_hops_per_chunk;
} zen_hip_hpr_params;

/* hps.h:216-285.  causality: ZEN_HIP_TIME_CAUSAL (HPRRealtime, hps.cu:287-295) or
 * ZEN_HIP_TIME_ANTICAUSAL (HPRIOffline passes, hps.cu:38-48).
 * n_streams >= 1 independent mono streams processed in lock step (1 for the reference API).
 * max_hops_per_chunk bounds device memory: longer calls are processed in chunks of that many hops
 * (0 = default: 2^30 / (n_streams * nfft) hops, clamped to 1..65536: 12 bytes of ring per element plus the rows of
 * the outputs in use -- about 2.7 GB for one hop-1024 stream, 14 GB for 64 offline clips). */
int zen_hip_hpr_create(float fs, size_t hop, float beta, unsigned output_flags, int causality,
                       int copy_bord, size_t n_streams, size_t max_hops_per_chunk, zen_hip_hpr_t* h);
int zen_hip_hpr_destroy(zen_hip_hpr_t h);
int zen_hip_hpr_get_params(zen_hip_hpr_t h, zen_hip_hpr_params* p);
int zen_hip_hpr_set_stream(zen_hip_hpr_t h, void* stream);
int zen_hip_hpr_use_sse_filter(zen_hip_hpr_t h); /* hps.h:289 */
int zen_hip_hpr_use_soft_mask(zen_hip_hpr_t h);  /* hps.h:291 */
int zen_hip_hpr_reset_buffers(zen_hip_hpr_t h);  /* hps.h:296-321 */

/* HPR<GPU>::process_next_hop (hps.cu:429-486): one hop of `hop` floats at in_dev (device or mapped
 * host memory; n_streams == 1).  Asynchronous. */
int zen_hip_hpr_process_next_hop(zen_hip_hpr_t h, const float* in_dev);
/* HPRRealtime<GPU>::copy_{harmonic,percussive,residual} (hps.cu:341-363): writes the first `hop`
 * floats of the accumulator for the hop(s) of the LAST process call to out_dev and returns when they are
 * readable there (the reference relies on thrust's implicit sync, SURVEY 8(b) "Threading"): by
 * synchronising the stream, or -- single stream, single hop, mapped host destination -- by polling the
 * sequence word the kernel publishes behind the finished hop (no stream synchronise: other work the caller
 * queued on the stream is NOT waited for, except earlier zen_hip_hpr_copy_output_async calls of this
 * engine).  A hop of the cooperative long-hop kernel whose grid barrier timed out is reported as
 * ZEN_HIP_E_HIP by every synchronous copy until zen_hip_hpr_reset_buffers.  After a block call it writes n_hops*hop floats per stream
 * (valid only if that call fitted in one chunk).  `which` is one ZEN_HIP_OUTPUT_* flag. */
int zen_hip_hpr_copy_output(zen_hip_hpr_t h, unsigned which, float* out_dev);
int zen_hip_hpr_copy_output_async(zen_hip_hpr_t h, unsigned which, float* out_dev);

/* Resident kernel for the per-hop path (an MI355X extension, off by default).  idle_ms > 0: single-hop calls of a causal
 * one-stream, one-output engine (median path or SSE path) whose hop sizes the one-workgroup kernels cover (hops 128..1024)
 * no longer cost a launch each: ONE workgroup is started by the first zen_hip_hpr_process_next_hop and stays on its CU, taking every
 * further hop from a mailbox in device-visible memory and publishing it behind the same sequence word
 * zen_hip_hpr_copy_output polls; it leaves by itself after idle_ms without a hop (at most 2000), when anything else is
 * asked of the engine (block calls, use_*, reset, set_stream, profile, destroy), and is started again by the next hop.
 * Same samples as the per-launch path, hop for hop.  While it is resident it occupies one CU, and device-wide
 * synchronising calls of the process (hipFree, hipDeviceSynchronize) wait for it to leave -- at most idle_ms.
 * idle_ms == 0 switches the mode off (and sends a resident kernel home). */
int zen_hip_hpr_set_resident(zen_hip_hpr_t h, int idle_ms);
/* launches of the resident kernel so far, hops processed by those that have ended, whether one is resident now */
int zen_hip_hpr_resident_stats(zen_hip_hpr_t h, unsigned long long* launches, unsigned long long* hops, int* active);

/* Block form.  in_dev: n_streams rows of n_hops*hop floats, `in_stride` floats apart.  Each non-NULL
 * out_*_dev receives n_streams rows of n_hops*hop floats, `out_stride` apart: exactly what n_hops
 * calls of process_next_hop + copy_* would have produced.  Asynchronous.  An output may BE the input (same pointer
 * and stride: in place; the engine then leaves that call's outputs to a launch that runs after every read of the chunk);
 * any other overlap of an output range with the input range is undefined, as are overlapping outputs. */
int zen_hip_hpr_process(zen_hip_hpr_t h, const float* in_dev, size_t n_hops, size_t in_stride,
                        float* out_harm_dev, float* out_perc_dev, float* out_resid_dev, size_t out_stride);
/* The block form on HOST buffers (one stream per engine): n_hops*hop floats in, n_hops*hop floats into each non-NULL
 * output -- the region zen/fakert.h:221-247 times per hop (host hop in, process_next_hop, copy_*, host hop out) for a whole
 * block: pieces of the block go up, through zen_hip_hpr_process and back down on three streams.  Buffers from
 * zen_hip_host_alloc_mapped are copied asynchronously; pageable ones are registered for the duration of the call.  The
 * outputs must not overlap the input or each other.  Synchronous: returns when the outputs are in the caller's buffers.
 * Same samples as the per-hop API, hop for hop. */
int zen_hip_hpr_process_host(zen_hip_hpr_t h, const float* in_host, size_t n_hops, float* out_harm_host, float* out_perc_host,
                             float* out_resid_host);

/* profiling hook for bench.py: HIP events around every kernel launch on the engine's stream.  get() synchronises and returns the summed kernel time and launch count. */
int zen_hip_hpr_profile(zen_hip_hpr_t h, int enable);
int zen_hip_hpr_profile_get(zen_hip_hpr_t h, double* median_ms, unsigned long long* median_launches,
                            unsigned long long* median_elements);
/* diagnostic: after this call single-hop launches of the fused causal kernel leave s_memrealtime stamps (100 MHz)
 * of their phases in the returned mapped host array of 8 words (tools/rt_latency.cpp --stamps). */
int zen_hip_hpr_debug_stamps(zen_hip_hpr_t h, unsigned long long** host_stamps);
/* summed milliseconds / launch counts per kernel class:
 * [0] STFT, [1] frequency filter, [2] time filter, [3] iSTFT, [4] overlap-add/copy-out,
 * [5] fused causal kernel (STFT + median + masks + iSTFT of a hop in one workgroup) */
int zen_hip_hpr_profile_get_all(zen_hip_hpr_t h, double ms[6], unsigned long long launches[6]);

/* ---------------------------------------------------------------------------------------------
 * HPRIOffline<Backend::GPU>   (libzen/hps.cu:21-221): two cascaded HPR passes ("HPR-I").
 * Returns what HPRIOffline<GPU>::process returns (hps.cu:219-220) computed with CPU filter semantics:
 * harm = pass-1 harmonic, perc = pass-2 percussive, resid = pass-2 residual_out, which the reference
 * never writes (pass 2 is built with OUTPUT_PERCUSSIVE only, hps.cu:45-48) and is therefore zeros.
 * ------------------------------------------------------------------------------------------- */
int zen_hip_hpri_create(float fs, size_t hop_h, size_t hop_p, float beta_h, float beta_p, int nocopybord,
                        size_t n_clips, zen_hip_hpri_t* h);
int zen_hip_hpri_destroy(zen_hip_hpri_t h);
int zen_hip_hpri_set_stream(zen_hip_hpri_t h, void* stream);
int zen_hip_hpri_use_sse_filter(zen_hip_hpri_t h); /* hps.cu:95-100 */
int zen_hip_hpri_use_soft_mask(zen_hip_hpri_t h);  /* hps.cu:102-107 */
/* HPRIOffline::process(std::vector<float>) (hps.cu:128-221; the call zen/offline.h:141-147 times): host buffers, n
 * samples each; any output may be NULL; the buffers must not overlap (ZEN_HIP_E_BAD_ARG).  Synchronises.  Clips of 8 Mi samples and more run as a pipeline over time
 * ranges (zen_hip_hpri_process_range): the upload of range k+1 and the download of range k-1 under the kernels of range
 * k, bit-identical to the whole clip.  Buffers the runtime does not know (plain malloc / std::vector) are registered
 * with hipHostRegister for the duration of the call so that their copies are asynchronous; where that fails the copies
 * block and only overlap the kernels.  resid_host is filled with zeros by host threads (the reference never
 * writes pass 2's residual), nothing is copied for it.  Options: "offline_range" = samples per range (0: default, 4 or 8 Mi),
 * "offline_no_register" = 1: never register. */
int zen_hip_hpri_process(zen_hip_hpri_t h, const float* audio_host, size_t n, float* harm_host,
                         float* perc_host, float* resid_host);
/* what the last zen_hip_hpri_process call of this handle did, for the harness (bench.py offline_host) */
typedef struct {
	size_t n_ranges, range_samples;
	int input_pinned, outputs_pinned; /* known to the runtime, or registered for the call: asynchronous copies */
	double setup_ms;                  /* staging buffers + registration */
	double enqueue_ms;                /* the host loop that feeds the three streams */
	double total_ms;                  /* the whole call */
} zen_hip_hpri_host_stats;
int zen_hip_hpri_host_stats_get(zen_hip_hpri_t h, zen_hip_hpri_host_stats* out);
/* The same call for a caller that BUILDS its result buffers (HPRIOffline::process returns three new std::vector<float>,
 * hps.cu:131, :219-220: fresh memory, which is slow to pin and which a value-initialising constructor writes once for
 * nothing): the ranges of the clip come down into pinned staging memory of the handle and are handed to `sink` from there --
 * sink(user, output, begin, samples, count) with output 0 = harmonic, 1 = percussive; for each output the ranges arrive in
 * ascending order, each exactly once, together covering [0, n); the two outputs' calls come from two threads of the library
 * and may overlap each other; `samples` is valid for the duration of the call.  The residual is all zeros:
 * nothing is delivered for it.  Returns when every range has been handed over. */
typedef void (*zen_hip_hpri_sink_fn)(void* user, int output, size_t begin, const float* samples, size_t count);
int zen_hip_hpri_process_sink(zen_hip_hpri_t h, const float* audio_host, size_t n, int want_harm, int want_perc,
                              zen_hip_hpri_sink_fn sink, void* user);
/* Device-resident batch of n_clips equal-length clips (rows `stride` floats apart, n samples used).
 * Outputs may be NULL.  Asynchronous. */
int zen_hip_hpri_process_device(zen_hip_hpri_t h, const float* audio_dev, size_t n, size_t stride,
                                float* harm_dev, float* perc_dev, float* resid_dev, size_t out_stride);
/* Time-sharding one long clip across GPUs (no reference counterpart).  Writes output
 * samples [begin, end) of the n-sample clip at audio_dev (indexed from sample 0 of the clip; only
 * [in_begin, in_end) as reported by zen_hip_hpri_range_halo is read) to harm_dev / perc_dev (end - begin
 * floats each, either may be NULL).  Bit-identical to the same range of zen_hip_hpri_process_device: each
 * shard re-runs a short warm-up (2*stft_width + 2 hops per pass) instead of exchanging state.
 * Handle created with n_clips == 1.  Asynchronous. */
int zen_hip_hpri_range_halo(zen_hip_hpri_t h, size_t n, size_t begin, size_t end, size_t* in_begin, size_t* in_end);
int zen_hip_hpri_process_range(zen_hip_hpri_t h, const float* audio_dev, size_t n, size_t begin, size_t end,
                               float* harm_dev, float* perc_dev);
/* profiling hooks for bench.py, as zen_hip_hpr_profile / _get_all, per pass (1: hop_h, 2: hop_p) */
int zen_hip_hpri_profile(zen_hip_hpri_t h, int enable);
int zen_hip_hpri_profile_get_all(zen_hip_hpri_t h, int pass, double ms[6], unsigned long long launches[6]);
/* How a pass of `frames` frames per stream at transform size nfft (2048..16384) would be synthesised in runs (no
 * reference counterpart; host arithmetic only, no device needed): group_outputs[g] = 1 or 2 outputs summed into destination
 * g (pass 1 of HPRIOffline with hard masks: {2, 1} = P + R, H).  *run = frames per run, or 0 where the long workgroups
 * of such runs would leave the device less than 92 % busy and the pass keeps one launch per frame; *busy (may be NULL) =
 * the simulated share of the device's workgroup slots doing useful transforms. */
int zen_hip_run_plan(size_t frames, size_t streams, size_t nfft, const int* group_outputs, int n_groups, int* run, double* busy);
/* hops the two passes run for an n-sample clip (hps.cu:109-126), for throughput accounting */
int zen_hip_hpri_hop_counts(zen_hip_hpri_t h, size_t n, size_t* n_hops_h, size_t* n_hops_p);

#ifdef __cplusplus
}
#endif
#endif /* ZEN_HIP_H */
