/*
 * ipp_check.c -- TEST INFRASTRUCTURE ONLY.  Built and run only on a host that HAS Intel IPP (oracle/ipp_probe.py
 * finds ipp.h + the libraries; neither the build container nor this pool's GPU boxes do, so here it never compiles).
 *
 * What it is for: the arithmetic of the reference's CPU path lives in IPP, which the oracle restates
 * from IPP's documented contract.  Where IPP exists this program calls the very functions the reference calls, with
 * the reference's arguments, next to the restatement, on the same seeded data, and prints one JSON line:
 *   median: ippiFilterMedianBorder_32f_C1R, ippBorderRepl, mask {1,len} / {len,1}     (libzen/mfilt.h:310-340)
 *   box:    ippiFilterBoxBorder_32f_C1R, ippBorderRepl                                (libzen/box.h:251-286)
 *   fft:    ippsFFT{Fwd,Inv}_CToC_32fc_I, IPP_FFT_NODIV_BY_ANY, ippAlgHintNone        (libzen/fftw.h:69-113)
 * with the number of differing outputs / the largest differences, and both sides' timings (the literal-IPP CPU
 * baseline BASELINE.md section 3 asks for when IPP is there).  The median must agree exactly (an order statistic);
 * box and FFT may differ in the last bits (summation order / butterfly order are IPP's own) -- that difference is
 * what "parity unpinned" in zen_oracle.h means, and this is the one place it could be measured.
 */
#include <ipp.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "zen_oracle.h"

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static float frand(unsigned* s)
{
	*s = *s * 1664525u + 1013904223u;
	return (float)((*s >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f; /* uniform(-1, 1) */
}

/* one filter comparison: time x frequency matrix, frequency contiguous (mfilt.h:76-79), step = frequency * 4 bytes */
static void check_filter(int is_box, int time, int frequency, int len, int dir, const char* tag)
{
	const size_t n = (size_t)time * (size_t)frequency;
	float *src = malloc(4 * n), *a = malloc(4 * n), *b = malloc(4 * n);
	unsigned seed = 12345u + (unsigned)len;
	for (size_t i = 0; i < n; ++i)
		src[i] = is_box ? fabsf(frand(&seed)) + 0.01f : frand(&seed);
	const int odd = len + (1 - (len % 2)); /* mfilt.h:305 / box.h:246 */
	IppiSize roi = {frequency, time};
	IppiSize mask = (dir == ZO_FREQUENCY) ? (IppiSize){odd, 1} : (IppiSize){1, odd};
	int bufsize = 0;
	IppStatus st = is_box ? ippiFilterBoxBorderGetBufferSize(roi, mask, ipp32f, 1, &bufsize)
	                      : ippiFilterMedianBorderGetBufferSize(roi, mask, ipp32f, 1, &bufsize);
	Ipp8u* buf = ippsMalloc_8u(bufsize > 0 ? bufsize : 1);
	const int step = frequency * (int)sizeof(float);
	double t0 = now_s();
	if (st >= 0)
		st = is_box ? ippiFilterBoxBorder_32f_C1R(src, step, a, step, roi, mask, ippBorderRepl, 0, buf)
		            : ippiFilterMedianBorder_32f_C1R(src, step, a, step, roi, mask, ippBorderRepl, 0, buf);
	double t_ipp = now_s() - t0;
	t0 = now_s();
	int rc = is_box ? zo_box_filter(src, b, time, frequency, len, dir) : zo_median_filter(src, b, time, frequency, len, dir);
	double t_port = now_s() - t0;
	size_t differ = 0;
	double max_rel = 0.0;
	for (size_t i = 0; i < n; ++i) {
		if (memcmp(&a[i], &b[i], 4) != 0 && !(a[i] == 0.0f && b[i] == 0.0f)) {
			++differ;
			const double r = fabs((double)a[i] - (double)b[i]) / (fabs((double)b[i]) + 1e-30);
			if (r > max_rel)
				max_rel = r;
		}
	}
	printf("{\"case\": \"%s\", \"time\": %d, \"frequency\": %d, \"len\": %d, \"ipp_status\": %d, \"port_status\": %d, "
	       "\"differing\": %zu, \"of\": %zu, \"max_rel_diff\": %.3g, \"ipp_s\": %.6f, \"port_s\": %.6f},\n",
	       tag, time, frequency, odd, (int)st, rc, differ, n, max_rel, t_ipp, t_port);
	ippsFree(buf);
	free(src);
	free(a);
	free(b);
}

static void check_fft(int order)
{
	const size_t n = (size_t)1 << order;
	float *x = malloc(8 * n), *a = malloc(8 * n), *b = malloc(8 * n);
	unsigned seed = 777u + (unsigned)order;
	for (size_t i = 0; i < 2 * n; ++i)
		x[i] = frand(&seed);
	int size_spec = 0, size_init = 0, size_buffer = 0;
	IppStatus st = ippsFFTGetSize_C_32fc(order, IPP_FFT_NODIV_BY_ANY, ippAlgHintNone, &size_spec, &size_init, &size_buffer);
	Ipp8u* m_spec = size_spec > 0 ? (Ipp8u*)ippMalloc(size_spec) : NULL;
	Ipp8u* m_init = size_init > 0 ? (Ipp8u*)ippMalloc(size_init) : NULL;
	Ipp8u* m_buf = size_buffer > 0 ? (Ipp8u*)ippMalloc(size_buffer) : NULL;
	IppsFFTSpec_C_32fc* spec = NULL;
	if (st == ippStsNoErr)
		st = ippsFFTInit_C_32fc(&spec, order, IPP_FFT_NODIV_BY_ANY, ippAlgHintNone, m_spec, m_init);
	memcpy(a, x, 8 * n);
	memcpy(b, x, 8 * n);
	double t0 = now_s();
	if (st == ippStsNoErr)
		st = ippsFFTFwd_CToC_32fc_I((Ipp32fc*)a, spec, m_buf);
	double t_ipp = now_s() - t0;
	t0 = now_s();
	int rc = zo_fft_c2c(b, n, 0);
	double t_port = now_s() - t0;
	double max_abs = 0.0;
	size_t differ = 0;
	for (size_t i = 0; i < 2 * n; ++i) {
		const double d = fabs((double)a[i] - (double)b[i]);
		if (d > max_abs)
			max_abs = d;
		differ += memcmp(&a[i], &b[i], 4) != 0;
	}
	printf("{\"case\": \"fft_forward\", \"n\": %zu, \"ipp_status\": %d, \"port_status\": %d, \"differing\": %zu, \"of\": %zu, "
	       "\"max_abs_diff\": %.3g, \"reference_tolerance\": 2e-4, \"ipp_s\": %.6f, \"port_s\": %.6f},\n",
	       n, (int)st, rc, differ, 2 * n, max_abs, t_ipp, t_port);
	if (m_init)
		ippFree(m_init);
	if (m_buf)
		ippFree(m_buf);
	if (m_spec)
		ippFree(m_spec);
	free(x);
	free(a);
	free(b);
}

int main(void)
{
	ippInit();
	const IppLibraryVersion* v = ippGetLibVersion();
	printf("{\"ipp_version\": \"%s %s\", \"cases\": [\n", v ? v->Name : "?", v ? v->Version : "?");
	/* the path shapes of BASELINE.md (a slice of rows each) and the reference's stripe-test shapes */
	check_filter(0, 512, 4096, 47, ZO_FREQUENCY, "median_freq_47");
	check_filter(0, 512, 4096, 3, ZO_TIME_ANTICAUSAL, "median_time_3");
	check_filter(0, 2048, 1024, 13, ZO_FREQUENCY, "median_freq_13");
	check_filter(0, 2048, 1024, 11, ZO_TIME_ANTICAUSAL, "median_time_11");
	check_filter(0, 128, 16384, 187, ZO_FREQUENCY, "median_freq_187");
	check_filter(0, 1024, 128, 5, ZO_TIME_CAUSAL, "median_stripe_shape");
	check_filter(1, 1024, 2048, 23, ZO_FREQUENCY, "box_freq_23");
	check_filter(1, 1024, 2048, 7, ZO_TIME_ANTICAUSAL, "box_time_7");
	check_fft(6);
	check_fft(10);
	check_fft(12);
	check_fft(14);
	printf("{\"case\": \"end\"}]}\n");
	return 0;
}
