// check_sqrt.hip -- the double-precision square root of fft_dev.h (cabs_exact: the compiler's Goldschmidt sequence
// without its range scaling, which |S|^2 of two floats never needs) against sqrt() on the device, for every value
// class the path can produce: squares of random float pairs over the whole exponent range, exact squares,
// neighbours of exact squares, zeros, denormal inputs, the largest floats.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I zen_amd/csrc tools/check_sqrt.hip -o /tmp/check_sqrt && /tmp/check_sqrt
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "fft_dev.h"

__device__ unsigned long long rng(unsigned long long& s)
{
	s ^= s << 13;
	s ^= s >> 7;
	s ^= s << 17;
	return s;
}

__global__ void check(unsigned long long seed, int iters, unsigned long long* bad, float* first)
{
	unsigned long long s = seed ^ (0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1));
	unsigned long long n_bad = 0;
	for (int it = 0; it < iters; ++it) {
		const unsigned long long r = rng(s);
		unsigned a = (unsigned)r, b = (unsigned)(r >> 32);
		const int mode = it & 7;
		if (mode == 1)
			b = a; // equal magnitudes
		if (mode == 2)
			b = 0; // one component zero: exact squares
		if (mode == 3)
			a &= 0x807fffffu; // denormal / tiny
		if (mode == 4)
			a = (a & 0x80000000u) | 0x7f7fffffu - (a & 0xff); // near the largest float
		if (mode == 5)
			b = (b & 0x807fffffu) | (a & 0x7f800000u); // same exponent
		float re, im;
		memcpy(&re, &a, 4);
		memcpy(&im, &b, 4);
		if (re != re || im != im || re - re != 0.0f || im - im != 0.0f)
			continue; // (NaN and inf inputs are outside the contract)
		const float ref = (float)sqrt((double)re * (double)re + (double)im * (double)im);
		const float got = zfft::cabs_exact(re, im);
		if (__float_as_uint(ref) != __float_as_uint(got)) {
			if (n_bad == 0 && atomicAdd(bad, 0ull) == 0) {
				first[0] = re;
				first[1] = im;
			}
			++n_bad;
		}
	}
	if (n_bad)
		atomicAdd(bad, n_bad);
}

int main()
{
	unsigned long long* bad;
	float* first;
	(void)hipMalloc(&bad, 8);
	(void)hipMalloc(&first, 8);
	(void)hipMemset(bad, 0, 8);
	const int blocks = 4096, threads = 256, iters = 4096, rounds = 8;
	for (int r = 0; r < rounds; ++r)
		hipLaunchKernelGGL(check, dim3(blocks), dim3(threads), 0, 0, 0x1234567ull + r, iters, bad, first);
	unsigned long long h = 0;
	float f[2] = {0, 0};
	(void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
	(void)hipMemcpy(f, first, 8, hipMemcpyDeviceToHost);
	std::printf("{\"pairs\": %.3g, \"mismatches\": %llu, \"first\": [%a, %a]}\n", (double)blocks * threads * iters * rounds, h, f[0], f[1]);
	return h != 0;
}
