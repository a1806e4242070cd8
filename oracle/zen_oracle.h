/*
 * zen_oracle.h -- CPU restatement of the sevagh/Zen HPSS hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X engine in zen_amd/.  It restates, in plain C99, the
 * reference's *CPU backend* (`HPR<Backend::CPU>`, libzen/hps.h + libzen/hps.cu) whose arithmetic lives
 * in Intel IPP (closed source, not vendored, version unpinned -- cmake/FindIPP.cmake:55-69).  IPP's
 * documented contracts are restated here:
 *   - ippiFilterMedianBorder_32f_C1R / ippiFilterBoxBorder_32f_C1R with ippBorderRepl: centred odd mask,
 *     out-of-image taps replicate the nearest edge sample          (libzen/mfilt.h:270-342, box.h:217-288)
 *   - ippsFFT{Fwd,Inv}_CToC_32fc_I with IPP_FFT_NODIV_BY_ANY: unnormalised DFT both ways
 *                                                                  (libzen/fftw.h:51-129)
 *
 * Who may use it: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the
 * product path (zen_amd/, include/) links, imports or executes this file.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - median filter: PINNED by the reference's own known-answer tests (libzen/mfilt.test.cu:315-591,
 *     stripe matrices 9x9/f3, 10x20/f5, 1024x128/f5, and the filter-too-big throw :525-534), carried in
 *     tests/test_oracle_golden.py; cross-checked against scipy.ndimage.median_filter(mode="nearest").
 *   - FFT: pinned only to the tolerance the reference itself uses (2e-4 abs vs another FFT,
 *     libzen/fftw.test.cu:16,83-101); checked here against numpy float64 FFTs at n = 64/1024/16384.
 *   - box filter: PARITY UNPINNED (reference suite disabled, libzen/CMakeLists.txt:82, and mostly commented
 *     out); restated from the IPP contract, summation order is this file's choice (ascending tap index,
 *     then divide).  Its three enabled cases (box.test.cu:124-199, a constant column) are carried over but
 *     cannot distinguish summation orders.
 *   - separated waveforms (HPRRealtime / HPRIOffline): PARITY UNPINNED -- the reference tests hold only
 *     EXPECT_NE / EXPECT_EQ properties (libzen/hps.test.cu:160-372), which are carried over as tests.
 *     Cross-checked against an independent float64 numpy model of the causal path (tests/test_oracle.py)
 *     and frozen as regression vectors in tests/golden/hpr_waveforms.npz.
 *
 * Arithmetic is IEEE binary32, round-to-nearest-even, no FMA contraction (compile with
 * -ffp-contract=off).  Every float operation is written out so that the HIP engine can reproduce it
 * bit for bit; where the reference calls libm (hypotf, powf) the exact formula used is stated below.
 */
#ifndef ZEN_ORACLE_H
#define ZEN_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libzen/mfilt.h:27-31 */
enum { ZO_TIME_CAUSAL = 0, ZO_TIME_ANTICAUSAL = 1, ZO_FREQUENCY = 2 };
/* libzen/libzen/hps.h:25-27 */
enum { ZO_OUTPUT_HARMONIC = 1, ZO_OUTPUT_PERCUSSIVE = 2, ZO_OUTPUT_RESIDUAL = 4 };
enum { ZO_OK = 0, ZO_E_FILTER_TOO_BIG = 1, ZO_E_BAD_ARG = 2, ZO_E_HOPS_NOT_DIVISIBLE = 3 };

/* ---- win.h:21-51 : periodic sqrt-von-Hann window, float math, PI = 3.14159265359F ---- */
void zo_window_sqrt_hann(float* w, size_t n);
void zo_window_hann(float* w, size_t n);

/* ---- fftw.h:51-129 : unnormalised complex DFT, power-of-two length ----
 * Textbook recursive radix-2 decimation-in-time.  Twiddles: one table tw[j] = exp(-2*pi*i*j/nfft),
 * j < nfft/2, evaluated in double for the first octant, rounded to float, completed by exact octant
 * symmetry so that tw[j + nfft/4] == -i * tw[j] bit for bit.  Butterfly at sub-size L, index k:
 *   t = tw[k*nfft/L] (x) odd[k];  out[k] = even[k] + t;  out[k+L/2] = even[k] - t
 *   (a+ib)(x)(c+id) = (a*c - b*d) + i(a*d + b*c), each product and sum rounded separately.
 * inverse != 0 uses conj(tw); no 1/n scaling either way (IPP_FFT_NODIV_BY_ANY). */
void zo_twiddles(float* tw_interleaved /* nfft/2 complex = nfft floats */, size_t nfft);
int zo_fft_c2c(float* data_interleaved /* nfft complex, in place */, size_t nfft, int inverse);

/* |z| as the reference's thrust::abs(complex<float>) = hypotf(re, im) (libzen/hps.h:82-89).
 * Written as (float)sqrt((double)re*re + (double)im*im): exact products, one rounding in the sum,
 * correctly rounded sqrt, one rounding to float -- the formula glibc >= 2.35 hypotf uses for finite
 * arguments (tests check equality with libm hypotf). */
float zo_cabs(float re, float im);
void zo_cabs_array(const float* z_interleaved, float* out, size_t n);

/* ---- mfilt.h:270-342 : MedianFilterCPU ----
 * src/dst: `time` rows x `frequency` columns, row-major, frequency contiguous (data[t*frequency + k]).
 * filter_len is made odd (len += 1 - len%2) after the too-big check, exactly as mfilt.h:296-305.
 * TimeCausal == TimeAnticausal on the CPU: mask {1, len} centred; Frequency: mask {len, 1} centred.
 * Returns ZO_E_FILTER_TOO_BIG where the reference throws ZgException. */
int zo_median_filter(const float* src, float* dst, int time, int frequency, int filter_len, int dir);
/* same semantics, O(len^2) per output copy-and-sort; used to check the sliding implementation */
int zo_median_filter_bruteforce(const float* src, float* dst, int time, int frequency, int filter_len,
                                int dir);

/* ---- box.h:217-288 : BoxFilterCPU (mean over the centred mask, replicate border) ----
 * dst = (sum of the len taps, ascending tap index, float accumulation starting from the first tap)
 *       / (float)len.   PARITY UNPINNED (see header). */
int zo_box_filter(const float* src, float* dst, int time, int frequency, int filter_len, int dir);

/* ---- hps.h:152-322 + hps.cu:429-652 : HPR<Backend::CPU> ---- */
typedef struct zo_hpr zo_hpr;

/* hps.h:216-285.  Returns NULL and sets *err on ZgException conditions (filter bigger than matrix). */
zo_hpr* zo_hpr_create(float fs, size_t hop, float beta, unsigned output_flags, int causality,
                      int copy_bord, int* err);
void zo_hpr_destroy(zo_hpr* h);
void zo_hpr_use_sse_filter(zo_hpr* h); /* hps.h:289 */
void zo_hpr_use_soft_mask(zo_hpr* h);  /* hps.h:291 */
void zo_hpr_reset_buffers(zo_hpr* h);  /* hps.h:296-321 */
/* hps.cu:429-486 (+ apply_median_filter :488-580, apply_sse_filter :582-652) */
void zo_hpr_process_next_hop(zo_hpr* h, const float* in_hop);
/* accumulators, nwin floats each; the first `hop` are what copy_* hands out (hps.cu:365-390) */
const float* zo_hpr_percussive_out(const zo_hpr* h);
const float* zo_hpr_harmonic_out(const zo_hpr* h);
const float* zo_hpr_residual_out(const zo_hpr* h);

typedef struct {
	size_t hop, nwin, nfft, stft_width;
	int l_harm, l_perc, lag;
	float cola_factor;
} zo_hpr_params;
void zo_hpr_get_params(const zo_hpr* h, zo_hpr_params* p);
/* stage taps for parity tests (pointers into the object's state, valid until the next call) */
const float* zo_hpr_window(const zo_hpr* h);            /* nwin */
const float* zo_hpr_sliding_stft(const zo_hpr* h);      /* stft_width*nfft complex interleaved */
const float* zo_hpr_s_mag(const zo_hpr* h);             /* stft_width*nfft */
const float* zo_hpr_harmonic_matrix(const zo_hpr* h);   /* stft_width*nfft */
const float* zo_hpr_percussive_matrix(const zo_hpr* h); /* stft_width*nfft */
const float* zo_hpr_percussive_mask(const zo_hpr* h);   /* stft_width*nfft (only lag row written) */
const float* zo_hpr_harmonic_mask(const zo_hpr* h);
const float* zo_hpr_residual_mask(const zo_hpr* h);

/* ---- hps.cu:282-427 : HPRRealtime<CPU> helpers ----
 * warmup (hps.cu:410-427): 1000 hops of iota data, then reset_buffers. */
void zo_hpr_warmup(zo_hpr* h);

/* ---- hps.cu:21-280 : HPRIOffline ---- */
typedef struct zo_hpri zo_hpri;
zo_hpri* zo_hpri_create(float fs, size_t hop_h, size_t hop_p, float beta_h, float beta_p, int nocopybord,
                        int* err);
void zo_hpri_destroy(zo_hpri* h);
void zo_hpri_use_sse_filter(zo_hpri* h);
void zo_hpri_use_soft_mask(zo_hpri* h);
/* hps.cu:109-126 : number of hops after padding; *padded_size = that many hops */
int zo_hpss_chunk_padder(size_t audio_size, size_t hop, size_t lag, size_t* padded_size);
/* HPRIOffline<CPU>::process (hps.cu:223-280) literally: the CPU backend returns
 * {percussive, percussive, percussive}; `perc` receives n floats. */
int zo_hpri_process_cpu(zo_hpri* h, const float* audio, size_t n, float* perc);
/* Same CPU filter semantics, but capturing what HPRIOffline<GPU>::process captures (hps.cu:128-221):
 * harm = pass-1 harmonic_out shifted by lag_h*hop_h; perc = pass-2 percussive; resid = pass-2
 * residual_out, which pass 2 never writes (constructed with OUTPUT_PERCUSSIVE only, hps.cu:45-48) and
 * is therefore all zeros.  Any of harm/perc/resid may be NULL. */
int zo_hpri_process(zo_hpri* h, const float* audio, size_t n, float* harm, float* perc, float* resid);

#ifdef __cplusplus
}
#endif
#endif /* ZEN_ORACLE_H */
