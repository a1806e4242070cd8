// probe_bar.cpp -- can the host write device memory directly (large BAR), and how fast?  Decides whether
// zen::io::IOGPU::host_in can live in HBM (the host pushes the hop with posted writes, the kernel reads
// local memory) instead of pinned host memory that the kernel pulls over the host link.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_bar.cpp -o /tmp/probe_bar
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

__global__ void sum_k(const float* p, int n, float* out)
{
	float s = 0;
	for (int i = threadIdx.x; i < n; i += blockDim.x)
		s += p[i];
	atomicAdd(out, s);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void try_buf(const char* name, float* p, int n)
{
	signal(SIGSEGV, on_segv);
	signal(SIGBUS, on_segv);
	if (sigsetjmp(jb, 1)) {
		printf("%-28s host write FAULTED\n", name);
		return;
	}
	float src[4096];
	for (int i = 0; i < n; ++i)
		src[i] = 1.0f;
	double t0 = now_us();
	for (int rep = 0; rep < 1000; ++rep)
		memcpy(p, src, n * 4);
	double t1 = now_us();
	float *out, h = 0;
	hipMalloc(&out, 4);
	hipMemset(out, 0, 4);
	sum_k<<<1, 256>>>(p, n, out);
	hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
	printf("%-28s host write ok: %.2f us per %d-float memcpy; kernel sees sum %.0f (expect %d)\n", name, (t1 - t0) / 1000, n, h, n);
	hipFree(out);
}

int main()
{
	const int n = 1024;
	float* p = nullptr;
	if (hipMalloc(&p, n * 4) == hipSuccess)
		try_buf("hipMalloc", p, n);
	float* q = nullptr;
	if (hipExtMallocWithFlags((void**)&q, n * 4, hipDeviceMallocFinegrained) == hipSuccess)
		try_buf("hipExtMalloc(Finegrained)", q, n);
	float* u = nullptr;
	if (hipExtMallocWithFlags((void**)&u, n * 4, hipDeviceMallocUncached) == hipSuccess)
		try_buf("hipExtMalloc(Uncached)", u, n);
	float* m = nullptr;
	if (hipMallocManaged(&m, n * 4) == hipSuccess)
		try_buf("hipMallocManaged", m, n);
	float* hm = nullptr;
	if (hipHostMalloc(&hm, n * 4, hipHostMallocMapped | hipHostMallocWriteCombined) == hipSuccess)
		try_buf("hipHostMalloc(WC)", hm, n);
	return 0;
}
