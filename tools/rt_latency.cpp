// rt_latency.cpp -- per-hop latency of the realtime call path through the C-ABI, timed as zen/fakert.h:221-247
// does (copy the hop into the mapped input buffer, process_next_hop, copy_percussive, copy the hop out),
// without an interpreter in the loop.  Prints one JSON line per hop size.
// Arguments: [hops per configuration, default 2000] [--stamps: also print the phase stamps of one single-hop kernel
// of every kind].  Environment: ZEN_RT_RESIDENT=<idle ms>: the per-hop calls go through the resident kernel
// (zen_hip_hpr_set_resident; every hop of the median path and SSE hop 512, "resident": 1 in their lines; with --stamps: the
// phase stamps are then those of a resident hop); ZEN_RT_OPT="name=value,..." (zen_hip_set_option), ZEN_RT_OUTPUT=H (the harmonic output), ZEN_RT_ONLY_SSE=1 (only the SSE configurations), ZEN_RT_DIAG=<n> (sets the library's
// "rt_fused_diag" option: timing diagnostics, results not valid; 4 = agent-scope grid barriers in rt_wide.hip).
//   g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/rt_latency -L zen_amd -lzen_hip -Wl,-rpath,$PWD/zen_amd
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "zen_hip.h"

#define CK(x)                                                                  \
	do {                                                                       \
		if ((x) != ZEN_HIP_OK) {                                               \
			std::fprintf(stderr, "%s: %s\n", #x, zen_hip_last_error());        \
			std::exit(1);                                                      \
		}                                                                      \
	} while (0)

static double now_us()
{
	return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
	const int n_hops = argc > 1 ? std::atoi(argv[1]) : 2000;
	// the two latency switches a realtime host opts into before its first HIP call; not overridden if set
	setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
	setenv("ZEN_HIP_SCHEDULE", "spin", 0);
	CK(zen_hip_init(0));
	if (const char* d = std::getenv("ZEN_RT_DIAG")) // timing diagnostics of the single-hop kernels (results not valid)
		CK(zen_hip_set_option("rt_fused_diag", std::atoi(d)));
	if (const char* o = std::getenv("ZEN_RT_OPT")) { // "name=value,name=value": zen_hip_set_option (e.g. no_sse_lat=1)
		std::string all(o);
		size_t pos = 0;
		while (pos < all.size()) {
			const size_t end = all.find(',', pos) == std::string::npos ? all.size() : all.find(',', pos);
			const std::string kv = all.substr(pos, end - pos);
			const size_t eq = kv.find('=');
			if (eq != std::string::npos)
				CK(zen_hip_set_option(kv.substr(0, eq).c_str(), std::atoi(kv.c_str() + eq + 1)));
			pos = end + 1;
		}
	}
	// ZEN_RT_OUTPUT=H: the harmonic output instead of the percussive one (the single-hop builds that carry every mask variant)
	const int out_flag = (std::getenv("ZEN_RT_OUTPUT") && std::getenv("ZEN_RT_OUTPUT")[0] == 'H') ? ZEN_HIP_OUTPUT_HARMONIC : ZEN_HIP_OUTPUT_PERCUSSIVE;
	const bool only_sse = std::getenv("ZEN_RT_ONLY_SSE") != nullptr;
	for (int sse = only_sse ? 1 : 0; sse < 2; ++sse) {
		for (size_t hop : {256, 512, 1024, 2048, 4096}) {
			if (sse && hop != 512 && hop != 1024 && hop != 2048)
				continue;
			zen_hip_hpr_t h = nullptr;
			CK(zen_hip_hpr_create(44100.f, hop, 2.0f, out_flag, ZEN_HIP_TIME_CAUSAL, 1, 1, 64, &h));
			if (sse)
				CK(zen_hip_hpr_use_sse_filter(h));
			const int resident_ms = std::getenv("ZEN_RT_RESIDENT") ? std::atoi(std::getenv("ZEN_RT_RESIDENT")) : 0;
			// (median path: every hop -- the one-workgroup kernel up to 1024, the cooperative one at 2048 / 4096; SSE path: the hops the
			// one-workgroup SSE kernel covers)
			const bool resident = resident_ms > 0 && (!sse || hop <= 1024);
			if (resident)
				CK(zen_hip_hpr_set_resident(h, resident_ms));
			if (argc > 2) { // (before the first hop: a resident kernel keeps the arguments of its launch)
				unsigned long long* st0 = nullptr;
				CK(zen_hip_hpr_debug_stamps(h, &st0));
			}
			void *hin, *din, *hout, *dout;
			CK(zen_hip_host_alloc_mapped(hop * 4, 1, &hin, &din));
			CK(zen_hip_host_alloc_mapped(hop * 4, 0, &hout, &dout));
			std::vector<float> x(hop * 64), y(hop);
			for (size_t i = 0; i < x.size(); ++i)
				x[i] = (float)((i * 2654435761u) % 20001) / 10000.f - 1.f;
			double t_proc = 0, t_copy = 0, t_all = 0;
			for (int i = -200; i < n_hops; ++i) { // 200 warm-up hops
				const float* src = x.data() + (size_t)(i & 63) * hop;
				const double t0 = now_us();
				std::memcpy(hin, src, hop * 4);
				CK(zen_hip_hpr_process_next_hop(h, (const float*)din));
				const double t1 = now_us();
				CK(zen_hip_hpr_copy_output(h, out_flag, (float*)dout));
				const double t2 = now_us();
				std::memcpy(y.data(), hout, hop * 4);
				const double t3 = now_us();
				if (i >= 0) {
					t_proc += t1 - t0;
					t_copy += t2 - t1;
					t_all += t3 - t0;
				}
			}
			if (argc > 2 && sse && hop <= 1024) { // the single-launch SSE kernel (rt_sse.hip)
				unsigned long long* st = nullptr;
				CK(zen_hip_hpr_debug_stamps(h, &st));
				std::memcpy(hin, x.data(), hop * 4);
				for (int rep = 0; rep < 3; ++rep) {
					CK(zen_hip_hpr_process_next_hop(h, (const float*)din));
					CK(zen_hip_hpr_copy_output(h, out_flag, (float*)dout));
				}
				{ // the stamps are written after the flag the copy call waits for
					const double t0 = now_us();
					while (now_us() - t0 < 2000.0) {}
				}
				std::printf("{\"hop\": %zu, \"sse\": 1, \"phase_us\": {\"carry_history\": %.2f, \"forward_fft_abs\": %.2f, "
				            "\"box_sums\": %.2f, \"mask_inverse_fft_store\": %.2f}, \"shader_clock_MHz\": %.0f}\n",
				            hop, (st[1] - st[0]) / 100.0, (st[2] - st[1]) / 100.0, (st[3] - st[2]) / 100.0,
				            (st[4] - st[3]) / 100.0, (double)st[5] / ((st[4] - st[0]) / 100.0));
			}
			if (argc > 2 && !sse && hop > 1024) { // the cooperative single-hop kernel (rt_wide.hip): phases and grid barriers
				unsigned long long* st = nullptr;
				CK(zen_hip_hpr_debug_stamps(h, &st));
				std::memcpy(hin, x.data(), hop * 4);
				for (int rep = 0; rep < 3; ++rep) {
					CK(zen_hip_hpr_process_next_hop(h, (const float*)din));
					CK(zen_hip_hpr_copy_output(h, out_flag, (float*)dout));
				}
				{
					const double t0 = now_us();
					while (now_us() - t0 < 2000.0) {}
				}
				auto d = [&](int i) { return (st[i + 1] - st[i]) / 100.0; };
				std::printf("{\"hop\": %zu, \"sse\": %d, \"phase_us\": {\"carries_analysis_a\": %.2f, \"barrier\": %.2f, \"analysis_b_abs\": %.2f, "
				            "\"barrier2\": %.2f, \"median\": %.2f, \"barrier3\": %.2f, \"mask_synthesis_a\": %.2f, \"barrier4\": %.2f, "
				            "\"synthesis_b_store\": %.2f, \"barrier5\": %.2f}, \"kernel_us\": %.2f, \"xcc_id_of_workgroups\": [%llu, %llu, %llu, %llu], \"light_barriers\": %llu}\n",
				            hop, sse, d(0), d(1), d(2), d(3), d(4), d(5), d(6), d(7), d(8), d(9), (st[10] - st[0]) / 100.0, st[12] & 15,
				            st[13] & 15, hop > 2048 ? st[14] & 15 : 99ull, hop > 2048 ? st[15] & 15 : 99ull, (st[12] >> 4) & 1);
			}
			if (argc > 2 && !sse && hop <= 1024) { // --stamps: phase times of the last single-hop launch
				unsigned long long* st = nullptr;
				CK(zen_hip_hpr_debug_stamps(h, &st));
				std::memcpy(hin, x.data(), hop * 4);
				for (int rep = 0; rep < 3; ++rep) {
					CK(zen_hip_hpr_process_next_hop(h, (const float*)din));
					CK(zen_hip_hpr_copy_output(h, out_flag, (float*)dout));
				}
				{ // the stamps are written after the flag the copy call waits for
					const double t0 = now_us();
					while (now_us() - t0 < 2000.0) {}
				}
				std::printf("{\"hop\": %zu, \"phase_us\": {\"housekeeping\": %.2f, \"forward_fft_abs\": %.2f, \"border_median\": %.2f, "
				            "\"mask_inverse_fft_store_publish\": %.2f}, \"kernel_us\": %.2f",
				            hop, (st[1] - st[0]) / 100.0, (st[2] - st[1]) / 100.0, (st[3] - st[2]) / 100.0,
				            (st[5] - st[3]) / 100.0, (st[5] - st[0]) / 100.0);
				if (st[6] > st[3] && st[8] >= st[7]) // rt_hop_lat.hip: the synthesis in three parts
					std::printf(", \"synthesis_us\": {\"masks\": %.2f, \"inverse_fft_stores\": %.2f, \"publish\": %.2f}", (st[6] - st[3]) / 100.0,
					            (st[7] - st[6]) / 100.0, (st[8] - st[7]) / 100.0);
				std::printf("}\n");
			}
			unsigned long long res_launches = 0;
			if (resident)
				CK(zen_hip_hpr_resident_stats(h, &res_launches, nullptr, nullptr));
			std::printf("{\"hop\": %zu, \"sse\": %d, \"resident\": %d, \"resident_launches\": %llu, \"us_per_hop\": %.2f, "
			            "\"us_process_call\": %.2f, \"us_copy_call\": %.2f, \"hops\": %d, \"pct_of_hop_period\": %.4f}\n",
			            hop, sse, resident ? 1 : 0, res_launches, t_all / n_hops, t_proc / n_hops, t_copy / n_hops, n_hops,
			            100.0 * (t_all / n_hops) / (1e6 * hop / 44100.0));
			zen_hip_host_free(hin);
			zen_hip_host_free(hout);
			zen_hip_hpr_destroy(h);
		}
	}
	return 0;
}
