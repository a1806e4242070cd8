/*
 * zen_oracle.c -- CPU restatement of the sevagh/Zen HPSS hot path.  TEST INFRASTRUCTURE ONLY:
 * see zen_oracle.h for who may use it, what is pinned and what is "parity unpinned".
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * Citations are into /root/reference (sevagh/Zen).
 */
#include "zen_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* libzen/hps.h:22 and libzen/libzen/zen.h:16 : Eps = std::numeric_limits<float>::epsilon() */
#define ZO_EPS FLT_EPSILON

/* ------------------------------------------------------------------------------------------------
 * win.h:21-51
 * ---------------------------------------------------------------------------------------------- */
static const float ZO_PI = 3.14159265359F; /* win.h:13 */

void zo_window_sqrt_hann(float* w, size_t n)
{
	float N = (float)n; /* win.h:34 : periodic, divides by window_size not window_size-1 */
	for (size_t i = 0; i < n; ++i)
		w[i] = sqrtf(0.5F * (1.0F - cosf(2.0F * ZO_PI * (float)i / N))); /* win.h:37-38 */
}

void zo_window_hann(float* w, size_t n)
{
	float N = (float)n;
	for (size_t i = 0; i < n; ++i)
		w[i] = 0.5F * (1.0F - cosf(2.0F * ZO_PI * (float)i / N)); /* win.h:45-46 */
}

/* ------------------------------------------------------------------------------------------------
 * fftw.h:51-129  (IPP C2C FFT, IPP_FFT_NODIV_BY_ANY) restated as textbook radix-2 DIT
 * ---------------------------------------------------------------------------------------------- */
static int is_pow2(size_t n) { return n && !(n & (n - 1)); }

void zo_twiddles(float* tw, size_t nfft)
{
	/* tw[j] = (cos(2 pi j/n), -sin(2 pi j/n)), j < n/2; first octant from double libm, the rest by
	 * exact symmetry so that tw[j + n/4] == -i*tw[j] and tw[n/8] has equal |re| and |im|. */
	size_t half = nfft / 2;
	if (half == 0)
		return;
	if (nfft < 8) { /* n = 2, 4: only trivial twiddles */
		tw[0] = 1.0F;
		tw[1] = 0.0F;
		if (nfft == 4) {
			tw[2] = 0.0F;
			tw[3] = -1.0F;
		}
		return;
	}
	size_t Q = nfft / 4, O = nfft / 8;
	float* c = (float*)malloc(sizeof(float) * (Q + 1));
	float* s = (float*)malloc(sizeof(float) * (Q + 1));
	const double two_pi = 6.283185307179586476925286766559;
	for (size_t j = 0; j <= O; ++j) {
		double th = two_pi * (double)j / (double)nfft;
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
		float cj, sj;
		if (j <= Q) {
			cj = c[j];
			sj = s[j];
		}
		else { /* cos(t + pi/2) = -sin t ; sin(t + pi/2) = cos t */
			cj = -s[j - Q];
			sj = c[j - Q];
		}
		tw[2 * j] = cj;
		tw[2 * j + 1] = -sj;
	}
	free(c);
	free(s);
}

/* out[0..n) = DFT_n of x[0], x[stride], x[2 stride], ...   (complex interleaved)
 * tw: table for the full transform; twstep = nfft / n is the table stride for this sub-size. */
static void fft_rec(const float* x, size_t stride, float* out, size_t n, const float* tw, size_t twstep,
                    int inverse)
{
	if (n == 1) {
		out[0] = x[0];
		out[1] = x[1];
		return;
	}
	size_t h = n / 2;
	fft_rec(x, 2 * stride, out, h, tw, 2 * twstep, inverse);                  /* even samples */
	fft_rec(x + 2 * stride, 2 * stride, out + 2 * h, h, tw, 2 * twstep, inverse); /* odd samples */
	for (size_t k = 0; k < h; ++k) {
		float wr = tw[2 * (k * twstep)];
		float wi = tw[2 * (k * twstep) + 1];
		if (inverse)
			wi = -wi;
		float br = out[2 * (k + h)], bi = out[2 * (k + h) + 1];
		float tr = wr * br - wi * bi;
		float ti = wr * bi + wi * br;
		float ar = out[2 * k], ai = out[2 * k + 1];
		out[2 * k] = ar + tr;
		out[2 * k + 1] = ai + ti;
		out[2 * (k + h)] = ar - tr;
		out[2 * (k + h) + 1] = ai - ti;
	}
}

typedef struct {
	size_t nfft;
	float* tw;      /* nfft/2 complex */
	float* scratch; /* nfft complex */
} zo_fft_plan;

static int fft_plan_init(zo_fft_plan* p, size_t nfft)
{
	if (!is_pow2(nfft))
		return ZO_E_BAD_ARG; /* fftw.h:59 : fft_order = (int)log2(nfft) -- power of two only */
	p->nfft = nfft;
	p->tw = (float*)malloc(sizeof(float) * (nfft < 2 ? 2 : nfft));
	p->scratch = (float*)malloc(sizeof(float) * 2 * nfft);
	zo_twiddles(p->tw, nfft);
	return ZO_OK;
}

static void fft_plan_free(zo_fft_plan* p)
{
	free(p->tw);
	free(p->scratch);
}

static void fft_exec(zo_fft_plan* p, float* data, int inverse)
{
	memcpy(p->scratch, data, sizeof(float) * 2 * p->nfft);
	fft_rec(p->scratch, 1, data, p->nfft, p->tw, 1, inverse);
}

int zo_fft_c2c(float* data, size_t nfft, int inverse)
{
	zo_fft_plan p;
	int rc = fft_plan_init(&p, nfft);
	if (rc)
		return rc;
	fft_exec(&p, data, inverse);
	fft_plan_free(&p);
	return ZO_OK;
}

float zo_cabs(float re, float im)
{
	double r = (double)re, i = (double)im;
	return (float)sqrt(r * r + i * i);
}

void zo_cabs_array(const float* z, float* out, size_t n)
{
	for (size_t i = 0; i < n; ++i)
		out[i] = zo_cabs(z[2 * i], z[2 * i + 1]);
}

/* ------------------------------------------------------------------------------------------------
 * mfilt.h:270-342  MedianFilterCPU  (ippiFilterMedianBorder_32f_C1R, ippBorderRepl)
 * ---------------------------------------------------------------------------------------------- */
static int filter_len_check(int time, int frequency, int filter_len, int dir)
{
	/* mfilt.h:296-303 / box.h:243-250 : checked BEFORE the length is made odd */
	if (time <= 0 || frequency <= 0 || filter_len <= 0)
		return ZO_E_BAD_ARG;
	if (((dir == ZO_TIME_CAUSAL || dir == ZO_TIME_ANTICAUSAL) && filter_len > time)
	    || (dir == ZO_FREQUENCY && filter_len > frequency))
		return ZO_E_FILTER_TOO_BIG;
	if (dir != ZO_TIME_CAUSAL && dir != ZO_TIME_ANTICAUSAL && dir != ZO_FREQUENCY)
		return ZO_E_BAD_ARG;
	return ZO_OK;
}

static inline long clampl(long v, long lo, long hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* sliding sorted window along one line of n samples spaced `stride` floats apart */
static void median_line(const float* src, float* dst, long n, long stride, int len, float* win)
{
	int mid = len / 2;
	/* window for i = 0 : taps clamp(-mid .. +mid) */
	int cnt = 0;
	for (long t = -mid; t <= mid; ++t) {
		float v = src[clampl(t, 0, n - 1) * stride];
		int p = cnt;
		while (p > 0 && win[p - 1] > v) {
			win[p] = win[p - 1];
			--p;
		}
		win[p] = v;
		++cnt;
	}
	dst[0] = win[mid];
	for (long i = 1; i < n; ++i) {
		float outv = src[clampl(i - 1 - mid, 0, n - 1) * stride];
		float inv = src[clampl(i + mid, 0, n - 1) * stride];
		/* remove one instance of outv */
		int p = 0;
		while (p < len && win[p] != outv)
			++p;
		/* (p < len always: outv was inserted) */
		for (; p < len - 1; ++p)
			win[p] = win[p + 1];
		/* insert inv */
		p = len - 1;
		while (p > 0 && win[p - 1] > inv) {
			win[p] = win[p - 1];
			--p;
		}
		win[p] = inv;
		dst[i * stride] = win[mid];
	}
}

int zo_median_filter(const float* src, float* dst, int time, int frequency, int filter_len, int dir)
{
	int rc = filter_len_check(time, frequency, filter_len, dir);
	if (rc)
		return rc;
	filter_len += (1 - (filter_len % 2)); /* mfilt.h:305 */
	float* win = (float*)malloc(sizeof(float) * (size_t)filter_len);
	if (dir == ZO_FREQUENCY) { /* mask {len, 1} : along the contiguous axis, mfilt.h:316 */
		for (long t = 0; t < time; ++t)
			median_line(src + t * (long)frequency, dst + t * (long)frequency, frequency, 1, filter_len, win);
	}
	else { /* mask {1, len} : along time; causal == anticausal on CPU, mfilt.h:311-314 */
		for (long k = 0; k < frequency; ++k)
			median_line(src + k, dst + k, time, frequency, filter_len, win);
	}
	free(win);
	return ZO_OK;
}

int zo_median_filter_bruteforce(const float* src, float* dst, int time, int frequency, int filter_len,
                                int dir)
{
	int rc = filter_len_check(time, frequency, filter_len, dir);
	if (rc)
		return rc;
	filter_len += (1 - (filter_len % 2));
	int mid = filter_len / 2;
	float* win = (float*)malloc(sizeof(float) * (size_t)filter_len);
	for (long t = 0; t < time; ++t) {
		for (long k = 0; k < frequency; ++k) {
			for (int j = 0; j < filter_len; ++j) {
				float v;
				if (dir == ZO_FREQUENCY)
					v = src[t * frequency + clampl(k - mid + j, 0, frequency - 1)];
				else
					v = src[clampl(t - mid + j, 0, time - 1) * frequency + k];
				int p = j;
				while (p > 0 && win[p - 1] > v) {
					win[p] = win[p - 1];
					--p;
				}
				win[p] = v;
			}
			dst[t * frequency + k] = win[mid];
		}
	}
	free(win);
	return ZO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * box.h:217-288  BoxFilterCPU  (ippiFilterBoxBorder_32f_C1R, ippBorderRepl)   -- parity unpinned
 * ---------------------------------------------------------------------------------------------- */
int zo_box_filter(const float* src, float* dst, int time, int frequency, int filter_len, int dir)
{
	int rc = filter_len_check(time, frequency, filter_len, dir);
	if (rc)
		return rc;
	filter_len += (1 - (filter_len % 2)); /* box.h:252 */
	int mid = filter_len / 2;
	float flen = (float)filter_len;
	for (long t = 0; t < time; ++t) {
		for (long k = 0; k < frequency; ++k) {
			float acc;
			if (dir == ZO_FREQUENCY) {
				acc = src[t * frequency + clampl(k - mid, 0, frequency - 1)];
				for (int j = 1; j < filter_len; ++j)
					acc = acc + src[t * frequency + clampl(k - mid + j, 0, frequency - 1)];
			}
			else {
				acc = src[clampl(t - mid, 0, time - 1) * frequency + k];
				for (int j = 1; j < filter_len; ++j)
					acc = acc + src[clampl(t - mid + j, 0, time - 1) * frequency + k];
			}
			dst[t * frequency + k] = acc / flen;
		}
	}
	return ZO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * hps.h:152-322  HPR<Backend::CPU>
 * ---------------------------------------------------------------------------------------------- */
struct zo_hpr {
	float fs;
	size_t hop, nwin, nfft;
	float beta;
	int l_harm, l_perc, lag;
	size_t stft_width;
	int causality;

	float* input;        /* nwin */
	float* window;       /* nwin */
	float* sliding_stft; /* stft_width*nfft complex */
	float* s_mag;
	float* reciprocal;
	float* harmonic_matrix;
	float* percussive_matrix;
	float* percussive_mask;
	float* harmonic_mask;
	float* residual_mask;
	float* percussive_out; /* nwin */
	float* harmonic_out;
	float* residual_out;
	float* fft_vec; /* nfft complex : fftw.h public member */
	float cola_factor;
	zo_fft_plan fft;
	int output_percussive, output_harmonic, output_residual, use_sse, soft_mask;
};

static float* zalloc(size_t n) { return (float*)calloc(n ? n : 1, sizeof(float)); }

zo_hpr* zo_hpr_create(float fs, size_t hop, float beta, unsigned output_flags, int causality,
                      int copy_bord, int* err)
{
	(void)copy_bord; /* mfilt.h:289 : "not used for CPU" */
	int e = ZO_OK;
	zo_hpr* h = (zo_hpr*)calloc(1, sizeof(zo_hpr));
	h->fs = fs;
	h->hop = hop;
	h->nwin = 2 * hop; /* hps.h:224 */
	h->nfft = 4 * hop; /* hps.h:225 */
	h->beta = beta;
	/* hps.h:227 : l_harm(roundf(0.2 / ((float)(nfft - hop) / fs))) -- double 0.2 over a float quotient */
	h->l_harm = (int)roundf((float)(0.2 / (double)((float)(h->nfft - hop) / fs)));
	h->lag = h->l_harm; /* hps.h:228 */
	/* hps.h:229 : l_perc(roundf(500 / (fs / (float)nfft))) -- all float */
	h->l_perc = (int)roundf(500.0F / (fs / (float)h->nfft));
	h->stft_width = (size_t)(2 * h->l_harm); /* hps.h:230 */
	h->causality = causality;
	if (!is_pow2(h->nfft) || h->l_harm < 1 || h->l_perc < 1
	    || (causality != ZO_TIME_CAUSAL && causality != ZO_TIME_ANTICAUSAL)) {
		e = ZO_E_BAD_ARG;
		goto fail;
	}
	/* hps.h:246-258 : the four filter objects are constructed here and may throw */
	e = filter_len_check((int)h->stft_width, (int)h->nfft, h->l_harm, causality);
	if (!e)
		e = filter_len_check((int)h->stft_width, (int)h->nfft, h->l_perc, ZO_FREQUENCY);
	if (e)
		goto fail;

	size_t mat = h->stft_width * h->nfft;
	h->input = zalloc(h->nwin);
	h->window = zalloc(h->nwin);
	zo_window_sqrt_hann(h->window, h->nwin); /* hps.h:232 */
	h->sliding_stft = zalloc(2 * mat);
	h->s_mag = zalloc(mat);
	h->reciprocal = zalloc(mat);
	h->harmonic_matrix = zalloc(mat);
	h->percussive_matrix = zalloc(mat);
	h->percussive_mask = zalloc(mat);
	h->harmonic_mask = zalloc(mat);
	h->residual_mask = zalloc(mat);
	h->percussive_out = zalloc(h->nwin);
	h->harmonic_out = zalloc(h->nwin);
	h->residual_out = zalloc(h->nwin);
	h->fft_vec = zalloc(2 * h->nfft);
	fft_plan_init(&h->fft, h->nfft);

	if (causality == ZO_TIME_CAUSAL)
		h->lag = 1; /* hps.h:265-268 */

	/* hps.h:270-274 : COLA = nfft / sum(win .* win), float accumulation in index order */
	h->cola_factor = 0.0f;
	for (size_t i = 0; i < h->nwin; ++i)
		h->cola_factor += h->window[i] * h->window[i];
	h->cola_factor = (float)h->nfft / h->cola_factor;

	h->output_harmonic = (output_flags & ZO_OUTPUT_HARMONIC) != 0;     /* hps.h:276-284 */
	h->output_percussive = (output_flags & ZO_OUTPUT_PERCUSSIVE) != 0;
	h->output_residual = (output_flags & ZO_OUTPUT_RESIDUAL) != 0;
	if (err)
		*err = ZO_OK;
	return h;
fail:
	free(h);
	if (err)
		*err = e;
	return NULL;
}

void zo_hpr_destroy(zo_hpr* h)
{
	if (!h)
		return;
	free(h->input);
	free(h->window);
	free(h->sliding_stft);
	free(h->s_mag);
	free(h->reciprocal);
	free(h->harmonic_matrix);
	free(h->percussive_matrix);
	free(h->percussive_mask);
	free(h->harmonic_mask);
	free(h->residual_mask);
	free(h->percussive_out);
	free(h->harmonic_out);
	free(h->residual_out);
	free(h->fft_vec);
	fft_plan_free(&h->fft);
	free(h);
}

void zo_hpr_use_sse_filter(zo_hpr* h) { h->use_sse = 1; }
void zo_hpr_use_soft_mask(zo_hpr* h) { h->soft_mask = 1; }

void zo_hpr_reset_buffers(zo_hpr* h)
{
	/* hps.h:296-321 : everything except the window and the filters' scratch */
	size_t mat = h->stft_width * h->nfft;
	memset(h->input, 0, sizeof(float) * h->nwin);
	memset(h->percussive_out, 0, sizeof(float) * h->nwin);
	memset(h->harmonic_out, 0, sizeof(float) * h->nwin);
	memset(h->residual_out, 0, sizeof(float) * h->nwin);
	memset(h->fft_vec, 0, sizeof(float) * 2 * h->nfft);
	memset(h->sliding_stft, 0, sizeof(float) * 2 * mat);
	memset(h->s_mag, 0, sizeof(float) * mat);
	memset(h->reciprocal, 0, sizeof(float) * mat);
	memset(h->harmonic_matrix, 0, sizeof(float) * mat);
	memset(h->percussive_matrix, 0, sizeof(float) * mat);
	memset(h->harmonic_mask, 0, sizeof(float) * mat);
	memset(h->percussive_mask, 0, sizeof(float) * mat);
	memset(h->residual_mask, 0, sizeof(float) * mat);
}

/* hps.h:100-113 hard_mask_functor */
static inline float hard_mask(float x, float y, float beta) { return (float)((x / (y + ZO_EPS)) >= beta); }

/* x^p for the integer power the reference passes to powf (hps.h:116-129, power = (int)beta).
 * Written as repeated multiplication (p = 2: x*x) so that the HIP engine can reproduce it exactly;
 * glibc powf(x, 2.0f) agrees to within 1 ulp (tests/test_oracle.py checks). */
static inline float powi(float x, int p)
{
#ifdef ZO_LITERAL_POWF /* tests/test_oracle.py builds this variant to measure what the substitution costs */
	return powf(x, (float)p);
#endif
	if (p <= 0)
		return 1.0F; /* powf(x, 0) = 1 */
	float r = x;
	for (int i = 1; i < p; ++i)
		r = r * x;
	return r;
}

/* hps.h:116-129 soft_mask_functor */
static inline float soft_mask_f(float x, float y, int power)
{
	float xp = powi(x, power), yp = powi(y, power);
	return xp / (xp + yp + ZO_EPS);
}

/* hps.h:132-140 sse_mask_functor */
static inline float sse_mask_f(float x, float y) { return x * x / (x * x + y * y + ZO_EPS); }
/* (hps.h:132-140 writes x*x itself; only complex_abs_squared_functor below calls powf) */

/* mask the lag row, inverse FFT, overlap-add: hps.cu:517-528 (and :550-560, :569-579) */
static void apply_ifft_ola(zo_hpr* h, const float* mask_row, float* out)
{
	size_t nfft = h->nfft, r = h->stft_width - (size_t)h->lag;
	const float* srow = h->sliding_stft + 2 * r * nfft;
	for (size_t k = 0; k < nfft; ++k) { /* apply_mask_functor hps.h:58-66 : complex * real */
		h->fft_vec[2 * k] = srow[2 * k] * mask_row[k];
		h->fft_vec[2 * k + 1] = srow[2 * k + 1] * mask_row[k];
	}
	fft_exec(&h->fft, h->fft_vec, 1); /* fft.backward(), unnormalised */
	for (size_t i = 0; i < h->nwin; ++i) /* overlap_add_functor hps.h:68-80 : y + x.real()*cola */
		out[i] = out[i] + h->fft_vec[2 * i] * h->cola_factor;
}

static void apply_median_filter(zo_hpr* h) /* hps.cu:488-580 */
{
	size_t W = h->stft_width, nfft = h->nfft, mat = W * nfft;
	size_t r = W - (size_t)h->lag; /* `X.end() - lag*nfft` */
	for (size_t i = 0; i < mat; ++i) /* hps.cu:492-493 complex_abs_functor over the whole matrix */
		h->s_mag[i] = zo_cabs(h->sliding_stft[2 * i], h->sliding_stft[2 * i + 1]);

	zo_median_filter(h->s_mag, h->harmonic_matrix, (int)W, (int)nfft, h->l_harm, h->causality); /* :495 */
	zo_median_filter(h->s_mag, h->percussive_matrix, (int)W, (int)nfft, h->l_perc, ZO_FREQUENCY); /* :496 */

	const float* Hr = h->harmonic_matrix + r * nfft;
	const float* Pr = h->percussive_matrix + r * nfft;
	if (h->output_percussive) { /* hps.cu:498-529 */
		float* m = h->percussive_mask + r * nfft;
		for (size_t k = 0; k < nfft; ++k)
			m[k] = h->soft_mask ? soft_mask_f(Pr[k], Hr[k], (int)h->beta) : hard_mask(Pr[k], Hr[k], h->beta);
		apply_ifft_ola(h, m, h->percussive_out);
	}
	if (h->output_harmonic) { /* hps.cu:531-560 ; hard mask threshold is beta - Eps */
		float* m = h->harmonic_mask + r * nfft;
		float beta_h = h->beta - ZO_EPS;
		for (size_t k = 0; k < nfft; ++k)
			m[k] = h->soft_mask ? soft_mask_f(Hr[k], Pr[k], (int)h->beta) : hard_mask(Hr[k], Pr[k], beta_h);
		apply_ifft_ola(h, m, h->harmonic_out);
	}
	if (h->output_residual && !h->soft_mask) { /* hps.cu:562-579 */
		for (size_t i = 0; i < mat; ++i) /* residual_mask_functor hps.h:35-43 over the whole matrix */
			h->residual_mask[i] = 1 - (h->harmonic_mask[i] + h->percussive_mask[i]);
		apply_ifft_ola(h, h->residual_mask + r * nfft, h->residual_out);
	}
}

static void apply_sse_filter(zo_hpr* h) /* hps.cu:582-652 */
{
	size_t W = h->stft_width, nfft = h->nfft, mat = W * nfft;
	size_t r = W - (size_t)h->lag;
	for (size_t i = 0; i < mat; ++i) { /* complex_abs_squared_functor hps.h:91-98 : powf(abs(z), 2) */
		float a = zo_cabs(h->sliding_stft[2 * i], h->sliding_stft[2 * i + 1]);
#ifdef ZO_LITERAL_POWF
		h->s_mag[i] = powf(a, 2);
#else
		h->s_mag[i] = a * a;
#endif
	}
	for (size_t i = 0; i < mat; ++i) /* reciprocal_functor(1.0F) hps.h:45-56 : (1/x)*factor */
		h->reciprocal[i] = (1.0f / h->s_mag[i]) * 1.0F;

	zo_box_filter(h->reciprocal, h->harmonic_matrix, (int)W, (int)nfft, h->l_harm, h->causality); /* :596 */
	zo_box_filter(h->reciprocal, h->percussive_matrix, (int)W, (int)nfft, h->l_perc, ZO_FREQUENCY); /* :597 */

	float fp = (float)h->l_perc + 1.0F, fh = (float)h->l_harm + 1.0F; /* hps.cu:599-604 */
	for (size_t i = 0; i < mat; ++i)
		h->percussive_matrix[i] = (1.0f / h->percussive_matrix[i]) * fp;
	for (size_t i = 0; i < mat; ++i)
		h->harmonic_matrix[i] = (1.0f / h->harmonic_matrix[i]) * fh;

	const float* Hr = h->harmonic_matrix + r * nfft;
	const float* Pr = h->percussive_matrix + r * nfft;
	if (h->output_percussive) { /* hps.cu:607-628 */
		float* m = h->percussive_mask + r * nfft;
		for (size_t k = 0; k < nfft; ++k)
			m[k] = sse_mask_f(Pr[k], Hr[k]);
		apply_ifft_ola(h, m, h->percussive_out);
	}
	if (h->output_harmonic) { /* hps.cu:630-651 */
		float* m = h->harmonic_mask + r * nfft;
		for (size_t k = 0; k < nfft; ++k)
			m[k] = sse_mask_f(Hr[k], Pr[k]);
		apply_ifft_ola(h, m, h->harmonic_out);
	}
}

void zo_hpr_process_next_hop(zo_hpr* h, const float* in_hop) /* hps.cu:429-486 */
{
	size_t hop = h->hop, nwin = h->nwin, nfft = h->nfft, W = h->stft_width;
	/* :435-449 shift the overlap-add accumulators left by hop, zero the tail */
	if (h->output_percussive) {
		memmove(h->percussive_out, h->percussive_out + hop, sizeof(float) * (nwin - hop));
		memset(h->percussive_out + hop, 0, sizeof(float) * (nwin - hop));
	}
	if (h->output_harmonic) {
		memmove(h->harmonic_out, h->harmonic_out + hop, sizeof(float) * (nwin - hop));
		memset(h->harmonic_out + hop, 0, sizeof(float) * (nwin - hop));
	}
	if (h->output_residual) {
		memmove(h->residual_out, h->residual_out + hop, sizeof(float) * (nwin - hop));
		memset(h->residual_out + hop, 0, sizeof(float) * (nwin - hop));
	}
	/* :452-453 input = input[hop:] ++ in_hop */
	memmove(h->input, h->input + hop, sizeof(float) * (nwin - hop));
	memcpy(h->input + hop, in_hop, sizeof(float) * hop);
	/* :456-462 window, zero-pad to nfft */
	for (size_t i = 0; i < nwin; ++i) { /* window_functor hps.h:24-33 : complex{x*y, 0} */
		h->fft_vec[2 * i] = h->input[i] * h->window[i];
		h->fft_vec[2 * i + 1] = 0.0F;
	}
	memset(h->fft_vec + 2 * nwin, 0, sizeof(float) * 2 * (nfft - nwin));
	fft_exec(&h->fft, h->fft_vec, 0); /* :465 */
	/* :469-472 rotate the sliding STFT up one row, append */
	memmove(h->sliding_stft, h->sliding_stft + 2 * nfft, sizeof(float) * 2 * (W - 1) * nfft);
	memcpy(h->sliding_stft + 2 * (W - 1) * nfft, h->fft_vec, sizeof(float) * 2 * nfft);
	if (!h->use_sse)
		apply_median_filter(h); /* :476-480 */
	else
		apply_sse_filter(h); /* :481-485 */
}

const float* zo_hpr_percussive_out(const zo_hpr* h) { return h->percussive_out; }
const float* zo_hpr_harmonic_out(const zo_hpr* h) { return h->harmonic_out; }
const float* zo_hpr_residual_out(const zo_hpr* h) { return h->residual_out; }
const float* zo_hpr_window(const zo_hpr* h) { return h->window; }
const float* zo_hpr_sliding_stft(const zo_hpr* h) { return h->sliding_stft; }
const float* zo_hpr_s_mag(const zo_hpr* h) { return h->s_mag; }
const float* zo_hpr_harmonic_matrix(const zo_hpr* h) { return h->harmonic_matrix; }
const float* zo_hpr_percussive_matrix(const zo_hpr* h) { return h->percussive_matrix; }
const float* zo_hpr_percussive_mask(const zo_hpr* h) { return h->percussive_mask; }
const float* zo_hpr_harmonic_mask(const zo_hpr* h) { return h->harmonic_mask; }
const float* zo_hpr_residual_mask(const zo_hpr* h) { return h->residual_mask; }

void zo_hpr_get_params(const zo_hpr* h, zo_hpr_params* p)
{
	p->hop = h->hop;
	p->nwin = h->nwin;
	p->nfft = h->nfft;
	p->stft_width = h->stft_width;
	p->l_harm = h->l_harm;
	p->l_perc = h->l_perc;
	p->lag = h->lag;
	p->cola_factor = h->cola_factor;
}

void zo_hpr_warmup(zo_hpr* h) /* hps.cu:410-427 */
{
	int test_iters = 1000;
	float* testdata = (float*)malloc(sizeof(float) * (size_t)test_iters * h->hop);
	float v = 0.0F; /* std::iota(..., 0.0F) : repeated ++ on a float */
	for (size_t i = 0; i < (size_t)test_iters * h->hop; ++i) {
		testdata[i] = v;
		v = v + 1.0F;
	}
	for (int i = 0; i < test_iters; ++i)
		zo_hpr_process_next_hop(h, testdata + (size_t)i * h->hop);
	zo_hpr_reset_buffers(h);
	free(testdata);
}

/* ------------------------------------------------------------------------------------------------
 * hps.cu:21-280  HPRIOffline
 * ---------------------------------------------------------------------------------------------- */
struct zo_hpri {
	zo_hpr* h; /* p_impl_h */
	zo_hpr* p; /* p_impl_p */
	size_t hop_h, hop_p;
};

zo_hpri* zo_hpri_create(float fs, size_t hop_h, size_t hop_p, float beta_h, float beta_p, int nocopybord,
                        int* err)
{
	if (hop_p == 0 || hop_h % hop_p != 0) { /* hps.cu:33-36 */
		if (err)
			*err = ZO_E_HOPS_NOT_DIVISIBLE;
		return NULL;
	}
	int e = ZO_OK;
	zo_hpri* o = (zo_hpri*)calloc(1, sizeof(zo_hpri));
	o->hop_h = hop_h;
	o->hop_p = hop_p;
	/* hps.cu:38-48 */
	o->h = zo_hpr_create(fs, hop_h, beta_h, ZO_OUTPUT_HARMONIC | ZO_OUTPUT_PERCUSSIVE | ZO_OUTPUT_RESIDUAL,
	                     ZO_TIME_ANTICAUSAL, !nocopybord, &e);
	if (o->h)
		o->p = zo_hpr_create(fs, hop_p, beta_p, ZO_OUTPUT_PERCUSSIVE, ZO_TIME_ANTICAUSAL, !nocopybord, &e);
	if (!o->h || !o->p) {
		zo_hpr_destroy(o->h);
		free(o);
		if (err)
			*err = e;
		return NULL;
	}
	if (err)
		*err = ZO_OK;
	return o;
}

void zo_hpri_destroy(zo_hpri* o)
{
	if (!o)
		return;
	zo_hpr_destroy(o->h);
	zo_hpr_destroy(o->p);
	free(o);
}

void zo_hpri_use_sse_filter(zo_hpri* o) /* hps.cu:95-100 */
{
	zo_hpr_use_sse_filter(o->h);
	zo_hpr_use_sse_filter(o->p);
}

void zo_hpri_use_soft_mask(zo_hpri* o) /* hps.cu:102-107 */
{
	zo_hpr_use_soft_mask(o->h);
	zo_hpr_use_soft_mask(o->p);
}

int zo_hpss_chunk_padder(size_t audio_size, size_t hop, size_t lag, size_t* padded_size)
{
	/* hps.cu:109-126 : float ceil of a float quotient, then + lag chunks; the vector is resized to
	 * exactly n_chunks*hop samples (zero filled when it grows). */
	int n_chunks = (int)(ceilf((float)audio_size / (float)hop));
	n_chunks += (int)lag;
	if (padded_size)
		*padded_size = (size_t)n_chunks * hop;
	return n_chunks;
}

int zo_hpri_process(zo_hpri* o, const float* audio, size_t n, float* harm, float* perc, float* resid)
{
	zo_hpr *H = o->h, *P = o->p;
	size_t hop_h = o->hop_h, hop_p = o->hop_p;
	zo_hpr_reset_buffers(H); /* a fresh HPRIOffline starts from zero state */
	zo_hpr_reset_buffers(P);

	size_t padded1;
	int n1 = zo_hpss_chunk_padder(n, hop_h, (size_t)H->lag, &padded1); /* hps.cu:229-230 */
	float* a = zalloc(padded1);
	memcpy(a, audio, sizeof(float) * (n < padded1 ? n : padded1));
	float* intermediate = zalloc(padded1); /* hps.cu:233 */
	float* harm_full = zalloc(padded1);    /* hps.cu:139 (GPU variant) */

	for (int i = 0; i < n1; ++i) { /* hps.cu:235-246 (and :142-167) */
		zo_hpr_process_next_hop(H, a + (size_t)i * hop_h);
		for (size_t j = 0; j < hop_h; ++j) { /* sum_vectors_functor hps.h:142-150 */
			intermediate[(size_t)i * hop_h + j] = H->percussive_out[j] + H->residual_out[j];
			harm_full[(size_t)i * hop_h + j] = H->harmonic_out[j];
		}
	}
	/* hps.cu:250-254 : shift left by lag_h*hop_h; the vector's tail keeps its old contents (Q9) */
	size_t sh1 = (size_t)H->lag * hop_h;
	memmove(intermediate, intermediate + sh1, sizeof(float) * (padded1 - sh1));
	memmove(harm_full, harm_full + sh1, sizeof(float) * (padded1 - sh1));

	size_t padded2;
	int n2 = zo_hpss_chunk_padder(n, hop_p, (size_t)P->lag, &padded2); /* hps.cu:256-257 */
	float* perc_full = zalloc(padded2);
	float* tmp = zalloc(hop_p);
	for (int i = 0; i < n2; ++i) { /* hps.cu:260-268 : reads `intermediate` past size(), inside capacity */
		size_t off = (size_t)i * hop_p;
		const float* in;
		if (off + hop_p <= padded1)
			in = intermediate + off;
		else { /* past the allocation in the reference (undefined there): zeros here */
			for (size_t j = 0; j < hop_p; ++j)
				tmp[j] = (off + j < padded1) ? intermediate[off + j] : 0.0F;
			in = tmp;
		}
		zo_hpr_process_next_hop(P, in);
		memcpy(perc_full + off, P->percussive_out, sizeof(float) * hop_p);
	}
	/* hps.cu:272-276 */
	size_t sh2 = (size_t)P->lag * hop_p;
	memmove(perc_full, perc_full + sh2, sizeof(float) * (padded2 - sh2));

	for (size_t i = 0; i < n; ++i) {
		if (perc)
			perc[i] = i < padded2 ? perc_full[i] : 0.0F;
		if (harm)
			harm[i] = i < padded1 ? harm_full[i] : 0.0F;
		if (resid)
			resid[i] = 0.0F; /* pass-2 residual_out is never written (hps.cu:45-48, :200-204) */
	}
	free(a);
	free(intermediate);
	free(harm_full);
	free(perc_full);
	free(tmp);
	return ZO_OK;
}

int zo_hpri_process_cpu(zo_hpri* o, const float* audio, size_t n, float* perc)
{
	return zo_hpri_process(o, audio, n, NULL, perc, NULL); /* hps.cu:278-279 : {perc, perc, perc} */
}
