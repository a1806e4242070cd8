// ubench_copy.hip -- practical HBM roof of the box: streaming copy / read / write kernels with 16-byte
// accesses over 1 GiB, several grid shapes.  hipcc --offload-arch=gfx950 -O3 tools/ubench_copy.hip -o tools/bin/ubench_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ __launch_bounds__(256) void copy_k(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
		b[i] = a[i];
}
template <int U>
__global__ __launch_bounds__(256) void copy_u(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
	const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
	float4 v[U];
#pragma unroll
	for (int u = 0; u < U; ++u)
		v[u] = a[base + (size_t)u * 256];
#pragma unroll
	for (int u = 0; u < U; ++u)
		b[base + (size_t)u * 256] = v[u];
}
// nontemporal variants: streaming data that will not be re-read should not displace cache lines
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_nt(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
	const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
	float4 v[U];
#pragma unroll
	for (int u = 0; u < U; ++u) {
		const float* p = reinterpret_cast<const float*>(a + base + (size_t)u * 256);
		if (NTL) {
			v[u].x = __builtin_nontemporal_load(p);
			v[u].y = __builtin_nontemporal_load(p + 1);
			v[u].z = __builtin_nontemporal_load(p + 2);
			v[u].w = __builtin_nontemporal_load(p + 3);
		}
		else
			v[u] = a[base + (size_t)u * 256];
	}
#pragma unroll
	for (int u = 0; u < U; ++u) {
		float* q = reinterpret_cast<float*>(b + base + (size_t)u * 256);
		if (NTS) {
			__builtin_nontemporal_store(v[u].x, q);
			__builtin_nontemporal_store(v[u].y, q + 1);
			__builtin_nontemporal_store(v[u].z, q + 2);
			__builtin_nontemporal_store(v[u].w, q + 3);
		}
		else
			b[base + (size_t)u * 256] = v[u];
	}
}
// persistent copy with the next tile's loads in flight while the current tile is stored
__global__ __launch_bounds__(256) void copy_pipe(const float4* __restrict__ a, float4* __restrict__ b, size_t ntiles)
{
	float4 v[4], w[4];
	size_t t = blockIdx.x;
	if (t >= ntiles)
		return;
#pragma unroll
	for (int u = 0; u < 4; ++u)
		v[u] = a[t * 1024 + threadIdx.x + u * 256];
	for (; t < ntiles; t += gridDim.x) {
		const size_t tn = t + gridDim.x;
		if (tn < ntiles) {
#pragma unroll
			for (int u = 0; u < 4; ++u)
				w[u] = a[tn * 1024 + threadIdx.x + u * 256];
		}
#pragma unroll
		for (int u = 0; u < 4; ++u)
			b[t * 1024 + threadIdx.x + u * 256] = v[u];
#pragma unroll
		for (int u = 0; u < 4; ++u)
			v[u] = w[u];
	}
}
__global__ __launch_bounds__(256) void read_k(const float4* __restrict__ a, float* out, size_t n)
{
	float s = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		const float4 v = a[i];
		s += v.x + v.y + v.z + v.w;
	}
	if (s == 123.456f)
		out[0] = s;
}
__global__ __launch_bounds__(256) void write_k(float4* __restrict__ b, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
		b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <class F>
float timeit(F f, int iters = 10)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	for (int i = 0; i < 3; ++i)
		f();
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	for (int i = 0; i < iters; ++i)
		f();
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	return ms / iters;
}

int main(int argc, char** argv)
{
	const size_t n = (size_t)1 << 26; // float4 elements: 1 GiB per buffer
	float4 *a, *b;
	float* o;
	(void)hipMalloc(&a, n * 16);
	(void)hipMalloc(&b, n * 16);
	(void)hipMalloc(&o, 4);
	(void)hipMemset(a, 0, n * 16);
	if (argc > 1 && !strcmp(argv[1], "--json")) {
		// bench.py's practical HBM denominator: the best streaming copy kernel here (4 x 16 B per thread, nontemporal loads and
		// stores) over the median kernel's working set (25 840 x 4096 floats in, the same out) and over 1 GiB, beside hipMemcpy
		const size_t m = (size_t)25840 * 1024;
		const float ms_m = timeit([&] { copy_nt<4, true, true><<<(unsigned)(m / (256 * 4)), 256>>>(a, b, m); }, 200);
		const float ms_g = timeit([&] { copy_nt<4, true, true><<<(unsigned)(n / (256 * 4)), 256>>>(a, b, n); }, 20);
		const float ms_c = timeit([&] { (void)hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0); }, 20);
		printf("{\"tuned_copy_median_shape_GBps\": %.1f, \"tuned_copy_1GiB_GBps\": %.1f, \"hipMemcpy_d2d_1GiB_GBps\": %.1f, "
		       "\"kernel\": \"copy_nt<4, true, true>: 4 x 16 B per thread, nontemporal loads and stores (tools/ubench_copy.hip)\"}\n",
		       2.0 * m * 16 / ms_m / 1e6, 2.0 * n * 16 / ms_g / 1e6, 2.0 * n * 16 / ms_c / 1e6);
		return 0;
	}
	for (int grid : {2048, 4096, 8192, 16384, 65536}) {
		float ms = timeit([&] { copy_k<<<grid, 256>>>(a, b, n); });
		printf("copy  grid-stride %6d blocks: %.3f ms  %.0f GB/s (read+write)\n", grid, ms, 2.0 * n * 16 / ms / 1e6);
	}
	{
		float ms = timeit([&] { copy_u<4><<<(unsigned)(n / (256 * 4)), 256>>>(a, b, n); });
		printf("copy  4 x 16 B per thread, one shot: %.3f ms  %.0f GB/s\n", ms, 2.0 * n * 16 / ms / 1e6);
		ms = timeit([&] { copy_u<8><<<(unsigned)(n / (256 * 8)), 256>>>(a, b, n); });
		printf("copy  8 x 16 B per thread, one shot: %.3f ms  %.0f GB/s\n", ms, 2.0 * n * 16 / ms / 1e6);
		ms = timeit([&] { copy_nt<4, true, false><<<(unsigned)(n / (256 * 4)), 256>>>(a, b, n); });
		printf("copy  4 x 16 B, nontemporal loads:   %.3f ms  %.0f GB/s\n", ms, 2.0 * n * 16 / ms / 1e6);
		ms = timeit([&] { copy_nt<4, false, true><<<(unsigned)(n / (256 * 4)), 256>>>(a, b, n); });
		printf("copy  4 x 16 B, nontemporal stores:  %.3f ms  %.0f GB/s\n", ms, 2.0 * n * 16 / ms / 1e6);
		ms = timeit([&] { copy_nt<4, true, true><<<(unsigned)(n / (256 * 4)), 256>>>(a, b, n); });
		printf("copy  4 x 16 B, nontemporal both:    %.3f ms  %.0f GB/s\n", ms, 2.0 * n * 16 / ms / 1e6);
		for (int g : {1024, 1792, 2048, 4096}) {
			ms = timeit([&] { copy_pipe<<<g, 256>>>(a, b, n / 1024); });
			printf("copy  persistent pipelined %5d WGs: %.3f ms  %.0f GB/s\n", g, ms, 2.0 * n * 16 / ms / 1e6);
		}
		// the median kernel's working set: 25 840 x 4096 floats in, the same out
		{
			const size_t m = (size_t)25840 * 1024;
			ms = timeit([&] { copy_u<4><<<(unsigned)(m / (256 * 4)), 256>>>(a, b, m); }, 50);
			printf("copy  4 x 16 B, 25840 x 4096 floats: %.3f ms  %.0f GB/s\n", ms, 2.0 * m * 16 / ms / 1e6);
			ms = timeit([&] { copy_nt<4, true, true><<<(unsigned)(m / (256 * 4)), 256>>>(a, b, m); }, 50);
			printf("copy  same, nontemporal both:        %.3f ms  %.0f GB/s\n", ms, 2.0 * m * 16 / ms / 1e6);
		}
		ms = timeit([&] { (void)hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0); });
		printf("hipMemcpy device-to-device:          %.3f ms  %.0f GB/s\n", ms, 2.0 * n * 16 / ms / 1e6);
	}
	for (int grid : {4096, 16384}) {
		float ms = timeit([&] { read_k<<<grid, 256>>>(a, o, n); });
		printf("read  grid-stride %6d blocks: %.3f ms  %.0f GB/s\n", grid, ms, 1.0 * n * 16 / ms / 1e6);
		ms = timeit([&] { write_k<<<grid, 256>>>(b, n); });
		printf("write grid-stride %6d blocks: %.3f ms  %.0f GB/s\n", grid, ms, 1.0 * n * 16 / ms / 1e6);
	}
	return 0;
}
