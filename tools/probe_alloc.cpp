// probe_alloc.cpp -- how fast can a process get three fresh zeroed n-float vectors?  The variants behind
// zen_amd/libzen/hps.cpp fresh_zeros(): plain value-initialisation, MADV_POPULATE_WRITE by k threads with and without
// transparent huge pages, then the zero fill.   g++ -O2 -std=c++17 -pthread tools/probe_alloc.cpp -o /tmp/probe_alloc
#include <sys/mman.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int populate(void* p, size_t bytes, unsigned k, bool thp)
{
	const uintptr_t page = 4096, huge = (uintptr_t)2 << 20;
	uintptr_t a = ((uintptr_t)p + page - 1) & ~(page - 1), e = ((uintptr_t)p + bytes) & ~(page - 1);
	int rc_thp = thp ? madvise((void*)a, e - a, MADV_HUGEPAGE) : 0;
	const uintptr_t per = (((e - a) / k) + huge - 1) & ~(huge - 1);
	std::vector<std::thread> th;
	std::vector<int> rcs(k, 0);
	for (unsigned i = 0; i < k; ++i) {
		const uintptr_t b0 = a + i * per, b1 = b0 + per < e ? b0 + per : e;
		if (b0 < b1)
			th.emplace_back([=, &rcs] { rcs[i] = madvise((void*)b0, b1 - b0, MADV_POPULATE_WRITE); });
	}
	for (auto& t : th)
		t.join();
	int bad = rc_thp;
	for (int r : rcs)
		bad |= r;
	return bad;
}

int main(int argc, char** argv)
{
	const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 158760000;
	{
		double t0 = now_ms();
		std::vector<float> a(n), b(n), c(n);
		asm volatile("" : : "r"(a.data()), "r"(b.data()), "r"(c.data()) : "memory");
		printf("{\"variant\": \"plain x3 serial\", \"ms\": %.1f}\n", now_ms() - t0);
	}
	{
		double t0 = now_ms();
		std::vector<float> v[3];
		std::thread t1([&] { v[0].resize(n); }), t2([&] { v[1].resize(n); });
		v[2].resize(n);
		t1.join();
		t2.join();
		printf("{\"variant\": \"plain x3, one thread each\", \"ms\": %.1f}\n", now_ms() - t0);
	}
	for (int thp = 0; thp < 2; ++thp)
		for (unsigned k : {1u, 4u, 8u, 16u, 32u}) {
			double t0 = now_ms();
			std::vector<float> v[3];
			double tp[3], tz[3];
			int rc[3];
			auto mk = [&](int i) {
				double a0 = now_ms();
				v[i].reserve(n);
				rc[i] = populate(v[i].data(), n * 4, k, thp);
				double a1 = now_ms();
				v[i].resize(n);
				tp[i] = a1 - a0;
				tz[i] = now_ms() - a1;
			};
			std::thread t1(mk, 0), t2(mk, 1);
			mk(2);
			t1.join();
			t2.join();
			double t_all = now_ms() - t0;
			double t1f = now_ms();
			for (auto& x : v)
				std::vector<float>().swap(x);
			printf("{\"variant\": \"populate\", \"thp\": %d, \"threads_per_vector\": %u, \"ms\": %.1f, \"populate_ms\": %.1f, \"zero_fill_ms\": %.1f, "
			       "\"madvise_failed\": %d, \"free_ms\": %.1f}\n", thp, k, t_all, tp[2], tz[2], rc[0] | rc[1] | rc[2], now_ms() - t1f);
		}
	return 0;
}
