/* san_driver.c -- drives every entry point of the CPU restatement (zen_oracle.c) for the sanitizer builds
 * (`make -C oracle san_driver_asan san_driver_ubsan`; tests/test_sanitizers.py).  TEST INFRASTRUCTURE ONLY.
 * The reference has the same two opt-in builds (libzen/CMakeLists.txt:108-133, README.md:140-147).  The
 * restatement reproduces two pieces of reference behaviour that deserve the check: the in-place shifts of
 * the offline driver and the read past size() but inside capacity of pass 2 (hps.cu:171-190).
 * Prints a checksum so that the three builds (plain, ASAN+UBSAN, UBSAN) can be compared. */
#include "zen_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned lcg = 12345u;
static float rnd(void)
{
	lcg = lcg * 1664525u + 1013904223u;
	return (float)(lcg >> 8) / 8388608.0f - 1.0f;
}
static double acc = 0;
static void fold(const float* x, size_t n)
{
	for (size_t i = 0; i < n; ++i)
		if (isfinite(x[i]))
			acc += fabs((double)x[i]) * (double)(1 + (i % 7));
}

static void run_realtime(float fs, size_t hop, unsigned flags, int causality, int soft, int sse, int hops)
{
	int err = 0;
	zo_hpr* h = zo_hpr_create(fs, hop, 2.0f, flags, causality, 1, &err);
	if (!h) {
		printf("hpr_create(%g, %zu) -> %d\n", fs, hop, err);
		return;
	}
	if (soft)
		zo_hpr_use_soft_mask(h);
	if (sse)
		zo_hpr_use_sse_filter(h);
	float* in = (float*)malloc(sizeof(float) * hop);
	for (int rep = 0; rep < 2; ++rep) {
		for (int i = 0; i < hops; ++i) {
			for (size_t k = 0; k < hop; ++k)
				in[k] = (i == 3 && k == 5) ? 1e30f : rnd(); /* one huge sample: inf/NaN paths of the masks */
			zo_hpr_process_next_hop(h, in);
			fold(zo_hpr_percussive_out(h), hop);
			fold(zo_hpr_harmonic_out(h), hop);
			fold(zo_hpr_residual_out(h), hop);
		}
		zo_hpr_reset_buffers(h);
	}
	zo_hpr_params p;
	zo_hpr_get_params(h, &p);
	fold(zo_hpr_window(h), p.nwin);
	fold(zo_hpr_s_mag(h), p.stft_width * p.nfft);
	free(in);
	zo_hpr_destroy(h);
}

static void run_offline(float fs, size_t hop_h, size_t hop_p, size_t n, int soft, int sse)
{
	int err = 0;
	zo_hpri* h = zo_hpri_create(fs, hop_h, hop_p, 2.5f, 2.5f, 0, &err);
	if (!h) {
		printf("hpri_create(%zu, %zu) -> %d\n", hop_h, hop_p, err);
		return;
	}
	if (soft)
		zo_hpri_use_soft_mask(h);
	if (sse)
		zo_hpri_use_sse_filter(h);
	float* x = (float*)malloc(sizeof(float) * (n ? n : 1));
	float* o[3];
	for (int k = 0; k < 3; ++k)
		o[k] = (float*)malloc(sizeof(float) * (n ? n : 1));
	for (size_t i = 0; i < n; ++i)
		x[i] = rnd();
	zo_hpri_process(h, x, n, o[0], o[1], o[2]);
	for (int k = 0; k < 3; ++k)
		fold(o[k], n);
	zo_hpri_process_cpu(h, x, n, o[1]);
	fold(o[1], n);
	zo_hpri_process(h, x, n, NULL, o[1], NULL);
	for (int k = 0; k < 3; ++k)
		free(o[k]);
	free(x);
	zo_hpri_destroy(h);
}

static void run_filters(void)
{
	static const int shapes[][3] = {{9, 9, 3},   {10, 20, 5}, {1, 1, 1},  {1, 64, 47}, {64, 1, 47}, {3, 200, 187},
	                                {22, 70, 13}, {5, 5, 5},   {2, 8, 8},  {7, 33, 32}, {300, 4, 255}};
	for (size_t s = 0; s < sizeof(shapes) / sizeof(shapes[0]); ++s) {
		const int t = shapes[s][0], f = shapes[s][1], len = shapes[s][2];
		float* a = (float*)malloc(sizeof(float) * t * f);
		float* b = (float*)malloc(sizeof(float) * t * f);
		float* c = (float*)malloc(sizeof(float) * t * f);
		for (int i = 0; i < t * f; ++i)
			a[i] = rnd();
		for (int dir = 0; dir < 3; ++dir) {
			const int rc = zo_median_filter(a, b, t, f, len, dir);
			const int rb = zo_median_filter_bruteforce(a, c, t, f, len, dir);
			if (rc != rb || (rc == ZO_OK && memcmp(b, c, sizeof(float) * t * f))) {
				printf("median mismatch %dx%d len %d dir %d\n", t, f, len, dir);
				exit(2);
			}
			if (rc == ZO_OK)
				fold(b, (size_t)t * f);
			if (zo_box_filter(a, b, t, f, len, dir) == ZO_OK)
				fold(b, (size_t)t * f);
		}
		free(a);
		free(b);
		free(c);
	}
}

int main(void)
{
	for (size_t n = 2; n <= 16384; n *= 2) {
		float* z = (float*)malloc(sizeof(float) * 2 * n);
		float* m = (float*)malloc(sizeof(float) * n);
		for (size_t i = 0; i < 2 * n; ++i)
			z[i] = rnd();
		zo_fft_c2c(z, n, 0);
		zo_cabs_array(z, m, n);
		fold(m, n);
		zo_fft_c2c(z, n, 1);
		fold(z, 2 * n);
		free(z);
		free(m);
	}
	printf("fft_c2c(48) -> %d\n", zo_fft_c2c((float[96]){0}, 48, 0));
	run_filters();
	const unsigned ALL = ZO_OUTPUT_HARMONIC | ZO_OUTPUT_PERCUSSIVE | ZO_OUTPUT_RESIDUAL;
	run_realtime(44100.f, 1024, ZO_OUTPUT_PERCUSSIVE, ZO_TIME_CAUSAL, 0, 0, 12);
	run_realtime(44100.f, 1024, ALL, ZO_TIME_CAUSAL, 0, 0, 8);
	run_realtime(48000.f, 256, ALL, ZO_TIME_ANTICAUSAL, 0, 0, 40);
	run_realtime(48000.f, 256, ALL, ZO_TIME_CAUSAL, 1, 0, 30);
	run_realtime(44100.f, 512, ZO_OUTPUT_PERCUSSIVE | ZO_OUTPUT_HARMONIC, ZO_TIME_CAUSAL, 0, 1, 30);
	run_realtime(44100.f, 4096, ALL, ZO_TIME_ANTICAUSAL, 0, 0, 4);
	run_realtime(44100.f, 8, ALL, ZO_TIME_CAUSAL, 0, 0, 50);
	run_realtime(44100.f, 100, ALL, ZO_TIME_CAUSAL, 0, 0, 1);   /* not a power of two: refused */
	run_realtime(8000.f, 4096, ALL, ZO_TIME_ANTICAUSAL, 0, 0, 1); /* l_harm 0: refused */
	static const size_t lens[] = {0, 1, 7, 255, 256, 257, 4095, 4096, 4097, 10000, 20 * 4096 + 11};
	for (size_t i = 0; i < sizeof(lens) / sizeof(lens[0]); ++i)
		run_offline(44100.f, 1024, 256, lens[i], 0, 0);
	run_offline(48000.f, 4096, 256, 20 * 4096 + 11, 0, 0); /* hps_cpu_public.test.cu:63-101 */
	run_offline(44100.f, 4096, 256, 30000, 1, 0);
	run_offline(44100.f, 2048, 512, 30000, 0, 1);
	run_offline(44100.f, 4096, 300, 100, 0, 0); /* hops not divisible: refused */
	zo_hpr* w = zo_hpr_create(44100.f, 256, 2.0f, ZO_OUTPUT_PERCUSSIVE, ZO_TIME_CAUSAL, 1, NULL);
	if (w) {
		zo_hpr_warmup(w);
		zo_hpr_destroy(w);
	}
	printf("checksum %.10e\n", acc);
	return 0;
}
