// probe_pcie.cpp -- what bounds HPRIOffline::process(std::vector<float>) (libzen/hps.cu:128-221) once the compute
// is 13 ms per hour of audio: the host link and the host's own memory system.  Measures, on the box it runs on,
//   * pinned <-> device copy rates, one direction and both at once (the roof of the offline_host bench leg:
//     4 B in + 8 B out per sample)
//   * pageable hipMemcpy rates (what the serial round-1 path got)
//   * hipHostRegister / Unregister cost of a caller's pageable buffer
//   * k-thread memcpy pageable <-> pinned (the staging alternative), k-thread memset
//   * first-touch cost of a fresh std::vector<float>(n) (what the C++ signature forces on every process() call)
//   hipcc --offload-arch=gfx950 -O2 -pthread tools/probe_pcie.cpp -o tools/bin/probe_pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at %s:%d\"}\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <class F> static double best_of(int reps, F f)
{
	double best = 1e30;
	for (int i = 0; i < reps; ++i) {
		const double t0 = now_ms();
		f();
		const double t = now_ms() - t0;
		if (t < best)
			best = t;
	}
	return best;
}

static void par(int k, size_t bytes, const std::function<void(size_t, size_t)>& f)
{
	std::vector<std::thread> th;
	const size_t per = ((bytes / k) + 4095) & ~(size_t)4095;
	for (int i = 0; i < k; ++i) {
		const size_t a = (size_t)i * per, b = a + per < bytes ? a + per : bytes;
		if (a < b)
			th.emplace_back([=, &f] { f(a, b - a); });
	}
	for (auto& t : th)
		t.join();
}

int main(int argc, char** argv)
{
	const size_t MB = argc > 1 ? (size_t)atol(argv[1]) : 512;
	const size_t bytes = MB << 20;
	CK(hipSetDevice(0));
	char *pin_a, *pin_b, *dev_a, *dev_b;
	CK(hipHostMalloc((void**)&pin_a, bytes, hipHostMallocDefault));
	CK(hipHostMalloc((void**)&pin_b, bytes, hipHostMallocDefault));
	CK(hipMalloc((void**)&dev_a, bytes));
	CK(hipMalloc((void**)&dev_b, bytes));
	memset(pin_a, 1, bytes);
	memset(pin_b, 2, bytes);
	hipStream_t s0, s1;
	CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
	const double gb = (double)bytes / 1e9;
	printf("{\"probe\": \"pcie\", \"mb\": %zu, \"host_threads\": %u", MB, std::thread::hardware_concurrency());
	{
		FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
		char buf[128] = "?";
		if (f) {
			if (!fgets(buf, sizeof buf, f))
				buf[0] = 0;
			fclose(f);
			buf[strcspn(buf, "\n")] = 0;
		}
		printf(", \"thp\": \"%s\"", buf);
	}
	double t;
	t = best_of(5, [&] { CK(hipMemcpyAsync(dev_a, pin_a, bytes, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); });
	printf(", \"pinned_h2d_GBps\": %.2f", gb / t * 1e3);
	t = best_of(5, [&] { CK(hipMemcpyAsync(pin_b, dev_b, bytes, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); });
	printf(", \"pinned_d2h_GBps\": %.2f", gb / t * 1e3);
	t = best_of(5, [&] {
		CK(hipMemcpyAsync(dev_a, pin_a, bytes, hipMemcpyHostToDevice, s0));
		CK(hipMemcpyAsync(pin_b, dev_b, bytes, hipMemcpyDeviceToHost, s1));
		CK(hipStreamSynchronize(s0));
		CK(hipStreamSynchronize(s1));
	});
	printf(", \"pinned_bidir_ms\": %.3f, \"pinned_bidir_each_GBps\": %.2f", t, gb / t * 1e3);
	// the offline shape: x bytes in, 2x bytes out, at once
	t = best_of(5, [&] {
		CK(hipMemcpyAsync(dev_a, pin_a, bytes / 2, hipMemcpyHostToDevice, s0));
		CK(hipMemcpyAsync(pin_b, dev_b, bytes, hipMemcpyDeviceToHost, s1));
		CK(hipStreamSynchronize(s0));
		CK(hipStreamSynchronize(s1));
	});
	printf(", \"pinned_in1_out2_ms\": %.3f, \"pinned_in1_out2_total_GBps\": %.2f", t, 1.5 * gb / t * 1e3);
	// small pinned copies: the per-chunk overhead of a pipeline
	for (size_t kb : {256, 1024, 4096, 16384}) {
		const size_t b = kb << 10;
		t = best_of(5, [&] {
			for (int i = 0; i < 16; ++i)
				CK(hipMemcpyAsync(dev_a + i * b, pin_a + i * b, b, hipMemcpyHostToDevice, s0));
			CK(hipStreamSynchronize(s0));
		});
		printf(", \"pinned_h2d_%zuKB_GBps\": %.2f", kb, 16.0 * b / 1e9 / t * 1e3);
	}
	// pageable
	char* pg_a = (char*)malloc(bytes);
	char* pg_b = (char*)malloc(bytes);
	memset(pg_a, 3, bytes);
	memset(pg_b, 4, bytes);
	t = best_of(3, [&] { CK(hipMemcpy(dev_a, pg_a, bytes, hipMemcpyHostToDevice)); });
	printf(", \"pageable_h2d_GBps\": %.2f", gb / t * 1e3);
	t = best_of(3, [&] { CK(hipMemcpy(pg_b, dev_b, bytes, hipMemcpyDeviceToHost)); });
	printf(", \"pageable_d2h_GBps\": %.2f", gb / t * 1e3);
	// register a caller's buffer
	{
		double t_reg = 1e30, t_unreg = 1e30, t_cp = 1e30;
		for (int i = 0; i < 3; ++i) {
			double t0 = now_ms();
			hipError_t e = hipHostRegister(pg_a, bytes, hipHostRegisterDefault);
			double t1 = now_ms();
			if (e != hipSuccess) {
				printf(", \"host_register\": \"%s\"", hipGetErrorString(e));
				break;
			}
			CK(hipMemcpyAsync(dev_a, pg_a, bytes, hipMemcpyHostToDevice, s0));
			CK(hipStreamSynchronize(s0));
			double t2 = now_ms();
			CK(hipHostUnregister(pg_a));
			double t3 = now_ms();
			if (t1 - t0 < t_reg) t_reg = t1 - t0;
			if (t2 - t1 < t_cp) t_cp = t2 - t1;
			if (t3 - t2 < t_unreg) t_unreg = t3 - t2;
		}
		printf(", \"host_register_GBps\": %.2f, \"host_unregister_GBps\": %.2f, \"registered_h2d_GBps\": %.2f", gb / t_reg * 1e3, gb / t_unreg * 1e3, gb / t_cp * 1e3);
		// is a copy from registered memory asynchronous?  time to enqueue vs time to complete
		if (hipHostRegister(pg_a, bytes, hipHostRegisterDefault) == hipSuccess) {
			double t0 = now_ms();
			CK(hipMemcpyAsync(dev_a, pg_a, bytes, hipMemcpyHostToDevice, s0));
			double t1 = now_ms();
			CK(hipStreamSynchronize(s0));
			double t2 = now_ms();
			printf(", \"registered_h2d_enqueue_ms\": %.3f, \"registered_h2d_complete_ms\": %.3f", t1 - t0, t2 - t0);
			CK(hipHostUnregister(pg_a));
		}
		{
			double t0 = now_ms();
			CK(hipMemcpyAsync(dev_a, pg_a, bytes, hipMemcpyHostToDevice, s0));
			double t1 = now_ms();
			CK(hipStreamSynchronize(s0));
			double t2 = now_ms();
			printf(", \"pageable_h2d_async_enqueue_ms\": %.3f, \"pageable_h2d_async_complete_ms\": %.3f", t1 - t0, t2 - t0);
		}
	}
	// threaded staging copies
	for (int k : {1, 2, 4, 8, 16}) {
		t = best_of(3, [&] { par(k, bytes, [&](size_t o, size_t n) { memcpy(pin_a + o, pg_a + o, n); }); });
		printf(", \"memcpy_page2pin_%dthr_GBps\": %.2f", k, gb / t * 1e3);
		t = best_of(3, [&] { par(k, bytes, [&](size_t o, size_t n) { memcpy(pg_b + o, pin_b + o, n); }); });
		printf(", \"memcpy_pin2page_%dthr_GBps\": %.2f", k, gb / t * 1e3);
	}
	for (int k : {1, 4, 8}) {
		t = best_of(3, [&] { par(k, bytes, [&](size_t o, size_t n) { memset(pg_b + o, 0, n); }); });
		printf(", \"memset_%dthr_GBps\": %.2f", k, gb / t * 1e3);
	}
	// first touch: what std::vector<float>(n) costs
	t = best_of(3, [&] { std::vector<float> v(bytes / 4); asm volatile("" : : "r"(v.data()) : "memory"); });
	printf(", \"fresh_vector_GBps\": %.2f", gb / t * 1e3);
	t = best_of(3, [&] { std::vector<float> v((const float*)pin_b, (const float*)pin_b + bytes / 4); asm volatile("" : : "r"(v.data()) : "memory"); });
	printf(", \"fresh_vector_from_pinned_GBps\": %.2f", gb / t * 1e3);
	// D2H straight into never-touched pageable memory
	t = best_of(3, [&] {
		char* f = (char*)malloc(bytes);
		CK(hipMemcpy(f, dev_b, bytes, hipMemcpyDeviceToHost));
		free(f);
	});
	printf(", \"pageable_d2h_untouched_GBps\": %.2f", gb / t * 1e3);
	printf("}\n");
	return 0;
}
