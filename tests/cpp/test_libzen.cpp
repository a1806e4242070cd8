// test_libzen.cpp -- the reference's gtest suites re-expressed against the C++ host mirror
// (zen_amd/libzen) on a real GPU, with the CPU oracle as the checker.  Runs under `pytest -m gpu`
// (tests/test_cpp_host.py).  Sources of the cases:
//   libzen/mfilt.test.cu  : stripe known answers (CPU / GPU-copybord "everywhere" form), filter-too-big
//   libzen/fftw.test.cu   : forward / inverse within 2e-4 of another FFT, n = 64 / 1024 / 16384
//   libzen/hps.test.cu    : ProcessingModifiesInput, PercOnlyOutputsPerc, ResettingDoesTheRightThing
//   libzen/hps_gpu_public.test.cu : HPRIOffline Basic / WithPadding, HPRRealtime
//   demos/pitch-tracking/main.cu:90-118, demos/beat-tracking/main.cu:92-127 : the HPR front ends of the demos
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include <core.h>
#include <hps.h>
#include <libzen/hps.h>
#include <libzen/io.h>

#include "../../oracle/zen_oracle.h"

using namespace zen;
using namespace zen::internal;
using namespace zen::internal::hps;
using namespace zen::internal::hps::mfilt;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                              \
	do {                                                                         \
		++g_checks;                                                              \
		if (!(cond)) {                                                           \
			if (++g_fail <= 20)                                                  \
				std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
		}                                                                        \
	} while (0)

static std::vector<float> generate_data_normalized(std::size_t size) // hps.test.cu:24-36
{
	static std::uniform_real_distribution<float> distribution(-1.0F, 1.0F);
	static std::default_random_engine generator;
	std::vector<float> data(size);
	for (auto& v : data)
		v = distribution(generator);
	return data;
}

static void test_mfilt_stripes(int x, int y, int f)
{
	std::vector<float> testdata(x * y, 0.0F);
	for (int i = 0; i < x; ++i)
		for (int j = 0; j < y; ++j) { // mfilt.test.cu:32-39
			if (i == x / 2)
				testdata[i * y + j] = 5;
			if (j == y / 2)
				testdata[i * y + j] = 8;
		}
	device_vector<float> src(testdata), dst(testdata.size());
	for (int variant = 0; variant < 2; ++variant) {
		const bool copy_bord = variant == 1;
		MedianFilterGPU causal(x, y, f, TimeCausal, copy_bord), anticausal(x, y, f, TimeAnticausal, copy_bord),
		    freq(x, y, f, Frequency, copy_bord);
		causal.filter(src, dst);
		auto r = dst.to_host();
		for (int i = 0; i < x; ++i)
			for (int j = 0; j < y; ++j)
				CHECK(r[i * y + j] == (j == y / 2 ? 8.0F : 0.0F)); // mfilt.test.cu:701-718 everywhere form
		anticausal.filter(src, dst);
		r = dst.to_host();
		for (int i = 0; i < x; ++i)
			for (int j = 0; j < y; ++j)
				CHECK(r[i * y + j] == (j == y / 2 ? 8.0F : 0.0F));
		freq.filter(src, dst);
		r = dst.to_host();
		for (int i = 0; i < x; ++i)
			for (int j = 0; j < y; ++j)
				CHECK(r[i * y + j] == (i == x / 2 ? 5.0F : 0.0F));
	}
}

static void test_mfilt_degenerate() // mfilt.test.cu:290-300, :525-534
{
	for (auto dir : {Frequency, TimeCausal, TimeAnticausal}) {
		bool thrown = false;
		try {
			MedianFilterGPU m(9, 9, 171, dir);
		}
		catch (const ZgException&) {
			thrown = true;
		}
		CHECK(thrown);
		thrown = false;
		try {
			box::BoxFilterGPU b(9, 9, 171, dir);
		}
		catch (const ZgException& e) {
			thrown = std::strstr(e.what(), "box filter") != nullptr;
		}
		CHECK(thrown);
	}
}

static void test_fft(std::size_t nfft) // fftw.test.cu:83-101, :132-286
{
	const float AllowableFFTError = 0.0002;
	std::vector<std::complex<float>> x(nfft);
	auto re = generate_data_normalized(nfft), im = generate_data_normalized(nfft);
	for (std::size_t i = 0; i < nfft; ++i)
		x[i] = {re[i], im[i]};
	fftw::FFTC2CWrapperGPU fftg(nfft);
	fftg.fft_vec.assign(x);
	fftg.forward();
	auto g = fftg.fft_vec.to_host();
	std::vector<std::complex<float>> c = x;
	zo_fft_c2c(reinterpret_cast<float*>(c.data()), nfft, 0);
	for (std::size_t i = 0; i < nfft; ++i) {
		CHECK(std::isfinite(g[i].real()) == std::isfinite(c[i].real()));
		CHECK(std::fabs(g[i].real() - c[i].real()) <= AllowableFFTError);
		CHECK(std::fabs(g[i].imag() - c[i].imag()) <= AllowableFFTError);
		CHECK(g[i] == c[i]); // and in fact bit-identical to the oracle
	}
	fftg.backward();
	g = fftg.fft_vec.to_host();
	zo_fft_c2c(reinterpret_cast<float*>(c.data()), nfft, 1);
	for (std::size_t i = 0; i < nfft; ++i)
		CHECK(g[i] == c[i]);
}

static void test_hpr_properties() // hps.test.cu:160-372 (GPU variants), hop 256, 100 hops, fs 48000
{
	const std::size_t hop = 256, n_hops = 100;
	auto data = generate_data_normalized(n_hops * hop);
	zen::io::IOGPU io(8192);
	const unsigned all = zen::hps::OUTPUT_HARMONIC | zen::hps::OUTPUT_PERCUSSIVE | zen::hps::OUTPUT_RESIDUAL;
	for (auto caus : {TimeCausal, TimeAnticausal}) {
		HPR<Backend::GPU> g_all(48000.0F, hop, 2.0, all, caus, true), g_all_ncb(48000.0F, hop, 2.0, all, caus, false),
		    g_perc(48000.0F, hop, 2.0, zen::hps::OUTPUT_PERCUSSIVE, caus, true);
		int err = 0;
		zo_hpr* o_all = zo_hpr_create(48000.0F, hop, 2.0, all, caus == TimeCausal ? ZO_TIME_CAUSAL : ZO_TIME_ANTICAUSAL, 1, &err);
		CHECK(o_all != nullptr);
		CHECK(g_all.l_harm == 12 && g_all.stft_width == 24 && g_all.l_perc == 11 && g_all.nfft == 1024);
		for (std::size_t i = 0; i < n_hops; ++i) {
			std::copy(data.begin() + i * hop, data.begin() + (i + 1) * hop, io.host_in);
			g_all.process_next_hop(io.device_in);
			g_all_ncb.process_next_hop(io.device_in);
			g_perc.process_next_hop(io.device_in);
			zo_hpr_process_next_hop(o_all, data.data() + i * hop);
			auto p = g_all.percussive_out(), h = g_all.harmonic_out(), r = g_all.residual_out();
			auto pn = g_all_ncb.percussive_out(), pp = g_perc.percussive_out();
			auto ph = g_perc.harmonic_out(), pr = g_perc.residual_out();
			for (std::size_t j = 0; j < hop; ++j) {
				if (caus == TimeCausal || i >= 12) { // anticausal output lags l_harm hops (all zero before)
					CHECK(data[i * hop + j] != p[j]); // ProcessingModifiesInput
					CHECK(data[i * hop + j] != h[j]);
					CHECK(data[i * hop + j] != r[j]);
				}
				CHECK(p[j] == pn[j]);             // CPU semantics: copybord == nocopybord (:260-261)
				CHECK(ph[j] == 0 && pr[j] == 0);  // PercOnlyOutputsPerc (:321-343)
				CHECK(p[j] == pp[j]);
				CHECK(p[j] == zo_hpr_percussive_out(o_all)[j]); // bit-exact vs the oracle
				CHECK(h[j] == zo_hpr_harmonic_out(o_all)[j]);
				CHECK(r[j] == zo_hpr_residual_out(o_all)[j]);
			}
		}
		zo_hpr_destroy(o_all);
	}
	// ResettingDoesTheRightThing (:345-372)
	HPR<Backend::GPU> g(48000.0F, hop, 2.0, zen::hps::OUTPUT_PERCUSSIVE, TimeCausal, true);
	std::copy(data.begin(), data.begin() + hop, io.host_in);
	g.process_next_hop(io.device_in);
	auto im = g.percussive_out();
	g.reset_buffers();
	g.process_next_hop(io.device_in);
	auto again = g.percussive_out();
	for (std::size_t j = 0; j < hop; ++j)
		CHECK(again[j] == im[j]);
}

static void test_public_api(std::size_t extra) // hps_gpu_public.test.cu (Basic / WithPadding)
{
	const std::size_t big_hop = 4096, small_hop = 256, n_big_hops = 20;
	auto testdata = generate_data_normalized(n_big_hops * big_hop);
	testdata.resize(testdata.size() + extra);
	zen::hps::HPRIOffline<Backend::GPU> hpri_offline(48000.0F, big_hop, small_hop, 2.0, 2.0);
	auto ret = hpri_offline.process(testdata);
	CHECK(ret[1].size() == testdata.size());
	for (std::size_t i = 0; i < big_hop * n_big_hops; ++i)
		CHECK(ret[1][i] != testdata[i]);
	int err = 0;
	zo_hpri* o = zo_hpri_create(48000.0F, big_hop, small_hop, 2.0, 2.0, 0, &err);
	std::vector<float> oh(testdata.size()), op(testdata.size()), orr(testdata.size());
	zo_hpri_process(o, testdata.data(), testdata.size(), oh.data(), op.data(), orr.data());
	zo_hpri_destroy(o);
	for (std::size_t i = 0; i < testdata.size(); ++i) {
		CHECK(ret[0][i] == oh[i]);
		CHECK(ret[1][i] == op[i]);
		CHECK(ret[2][i] == 0.0F);
	}
	// HPRRealtime through mapped buffers, warm-up first (zen/fakert.h:145-247)
	zen::hps::HPRRealtime<Backend::GPU> p_rt(48000.0F, small_hop, 2.0, zen::hps::OUTPUT_PERCUSSIVE);
	zen::io::IOGPU io(small_hop);
	p_rt.warmup(io);
	zo_hpr* ort = zo_hpr_create(48000.0F, small_hop, 2.0, ZO_OUTPUT_PERCUSSIVE, ZO_TIME_CAUSAL, 1, &err);
	const std::size_t n_small_hops = 60;
	for (std::size_t i = 0; i < n_small_hops; ++i) {
		std::copy(testdata.begin() + i * small_hop, testdata.begin() + (i + 1) * small_hop, io.host_in);
		p_rt.process_next_hop(io.device_in);
		p_rt.copy_percussive(io.device_out);
		zo_hpr_process_next_hop(ort, testdata.data() + i * small_hop);
		for (std::size_t j = 0; j < small_hop; ++j) {
			CHECK(testdata[i * small_hop + j] != io.host_out[j]);
			CHECK(io.host_out[j] == zo_hpr_percussive_out(ort)[j]);
		}
	}
	zo_hpr_destroy(ort);
	auto none = hpri_offline.process(std::vector<float>()); // an empty clip gives three empty vectors
	CHECK(none[0].empty() && none[1].empty() && none[2].empty());
	bool thrown = false;
	try {
		zen::hps::HPRIOffline<Backend::GPU> bad(48000.0F, 4096, 300, 2.0, 2.0);
	}
	catch (const ZgException& e) {
		thrown = std::string(e.what()) == "hop_h and hop_p should be evenly divisible";
	}
	CHECK(thrown);
}

// The two downstream demos drive HPRRealtime<GPU> exactly like this: pitch tracking takes
// the harmonic output at hop 4096 / beta 2.5 (demos/pitch-tracking/main.cu:90-118), beat tracking the
// percussive output at hop 256 / beta 2.5 (demos/beat-tracking/main.cu:92-127), both through IOGPU's
// mapped buffers, one hop per call.  Only the HPR front half is in scope; it must equal the CPU path.
static void test_demo_front_end(float fs, std::size_t hop, float beta, unsigned what, std::size_t n_hops, int resident_ms = 0)
{
	auto x = generate_data_normalized(n_hops * hop);
	zen::hps::HPRRealtime<Backend::GPU> rt(fs, hop, beta, what);
	zen::io::IOGPU io(hop);
	rt.warmup(io);
	if (resident_ms > 0) // the MI355X extension: the same loop served by a resident kernel (no launch per hop)
		rt.use_resident_kernel(resident_ms);
	int err = 0;
	zo_hpr* o = zo_hpr_create(fs, hop, beta, what == zen::hps::OUTPUT_HARMONIC ? ZO_OUTPUT_HARMONIC : ZO_OUTPUT_PERCUSSIVE,
	                          ZO_TIME_CAUSAL, 1, &err);
	CHECK(o != nullptr);
	bool any = false;
	for (std::size_t i = 0; i < n_hops; ++i) {
		std::copy(x.begin() + i * hop, x.begin() + (i + 1) * hop, io.host_in);
		rt.process_next_hop(io.device_in);
		if (what == zen::hps::OUTPUT_HARMONIC)
			rt.copy_harmonic(io.device_out);
		else
			rt.copy_percussive(io.device_out);
		zo_hpr_process_next_hop(o, x.data() + i * hop);
		const float* ref = what == zen::hps::OUTPUT_HARMONIC ? zo_hpr_harmonic_out(o) : zo_hpr_percussive_out(o);
		for (std::size_t j = 0; j < hop; ++j) {
			CHECK(io.host_out[j] == ref[j]);
			any = any || io.host_out[j] != 0.0F;
		}
	}
	CHECK(any);
	zo_hpr_destroy(o);
}

// The block of hops from host buffers (HPRRealtime::process_hops_host, an MI355X extension: the loop of zen/fakert.h:221-247
// for a whole block): pageable vectors and IOGPU's pinned buffer, against the per-hop CPU path.
static void test_block_from_host(std::size_t hop, std::size_t n_hops)
{
	auto x = generate_data_normalized(n_hops * hop);
	zen::hps::HPRRealtime<Backend::GPU> rt(44100.0F, hop, 2.0F, zen::hps::OUTPUT_PERCUSSIVE | zen::hps::OUTPUT_HARMONIC);
	std::vector<float> perc(x.size(), -1.0F);
	zen::io::IOGPU pinned(x.size());
	rt.process_hops_host(x.data(), n_hops, pinned.host_out, perc.data(), nullptr);
	int err = 0;
	zo_hpr* o = zo_hpr_create(44100.0F, hop, 2.0F, ZO_OUTPUT_PERCUSSIVE | ZO_OUTPUT_HARMONIC, ZO_TIME_CAUSAL, 1, &err);
	CHECK(o != nullptr);
	bool same = true, any = false;
	for (std::size_t i = 0; i < n_hops; ++i) {
		zo_hpr_process_next_hop(o, x.data() + i * hop);
		for (std::size_t j = 0; j < hop; ++j) {
			same = same && perc[i * hop + j] == zo_hpr_percussive_out(o)[j] && pinned.host_out[i * hop + j] == zo_hpr_harmonic_out(o)[j];
			any = any || perc[i * hop + j] != 0.0F;
		}
	}
	CHECK(same);
	CHECK(any);
	zo_hpr_destroy(o);
}

// HPRIOffline<GPU>::process on a clip long enough for the host-vector pipeline (ranges of 4 Mi samples, result vectors
// populated by threads): equal, sample for sample, to the same call on its first part run alone wherever the two must
// agree -- an offline output sample depends on a few hops around it -- and the residual is zeros (hps.cu:219-220, Q8).
static void test_offline_long_clip()
{
	const std::size_t n = ((std::size_t)9 << 20) + 77, m = (std::size_t)1 << 20;
	auto base = generate_data_normalized(m);
	std::vector<float> x(n);
	for (std::size_t i = 0; i < n; ++i)
		x[i] = base[i % m] * (0.5F + 0.5F * (float)((i / m) & 1));
	zen::hps::HPRIOffline<Backend::GPU> hpss(44100.0F, 4096, 256, 2.0, 2.0);
	auto all = hpss.process(x);
	CHECK(all[0].size() == n && all[1].size() == n && all[2].size() == n);
	std::vector<float> head(x.begin(), x.begin() + 3 * m);
	auto part = hpss.process(head); // under 8 Mi samples: one range, no pipeline
	bool same = true, any = false, zeros = true;
	for (std::size_t i = 0; i < 2 * m; ++i) { // (the last Mi samples of `head` see its end; the first two do not)
		same = same && all[0][i] == part[0][i] && all[1][i] == part[1][i];
		any = any || all[1][i] != 0.0F;
	}
	for (std::size_t i = 0; i < n; i += 997)
		zeros = zeros && all[2][i] == 0.0F;
	CHECK(same);
	CHECK(any);
	CHECK(zeros);
}

int main()
{
	if (zen_hip_init(0) != ZEN_HIP_OK) {
		std::printf("no GPU: %s\n", zen_hip_last_error());
		return 2;
	}
	test_mfilt_stripes(9, 9, 3);      // MedianFilterSmallSquare
	test_mfilt_stripes(10, 20, 5);    // SmallRectangle
	test_mfilt_stripes(1024, 17, 5);  // LargeRectangle (GPU fixture)
	test_mfilt_stripes(1024, 128, 5); // LargeRectangle (CPU fixture)
	test_mfilt_degenerate();
	test_fft(64);
	test_fft(1024);
	test_fft(16384);
	test_hpr_properties();
	test_public_api(0);
	test_public_api(11);
	test_demo_front_end(48000.0F, 4096, 2.5F, zen::hps::OUTPUT_HARMONIC, 12);   // pitch-tracking front end
	test_demo_front_end(44100.0F, 4096, 2.5F, zen::hps::OUTPUT_HARMONIC, 12);
	test_demo_front_end(44100.0F, 256, 2.5F, zen::hps::OUTPUT_PERCUSSIVE, 80);  // beat-tracking front end
	test_demo_front_end(48000.0F, 256, 2.5F, zen::hps::OUTPUT_PERCUSSIVE, 80);
	test_demo_front_end(44100.0F, 256, 2.5F, zen::hps::OUTPUT_PERCUSSIVE, 80, 50);  // ... through the resident kernel
	test_demo_front_end(44100.0F, 1024, 2.0F, zen::hps::OUTPUT_HARMONIC, 40, 50);
	test_block_from_host(1024, 37);
	test_block_from_host(256, 4500); // (several pieces of the pipeline)
	test_offline_long_clip();
	std::printf("%d checks, %d failures\n", g_checks, g_fail);
	return g_fail ? 1 : 0;
}
