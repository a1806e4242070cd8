// check_div.hip -- exhaustive proof, on the GPU, of the two short division forms of exact_div.h against the IEEE division the
// compiler emits (-fno-fast-math: v_div_scale / v_rcp / five fma / v_div_fmas / v_div_fixup, ~15 issue slots):
//   recip_exact(x)    = 1.0f / x      for EVERY float x whose bits pass its range test (2^-126 <= |x| <= 2^126)
//   div_const_exact   = x / c         for EVERY float x that passes its range test and every integer c = 1..255
// Prints the number of mismatches (bit patterns compared); exit status 1 if there is one.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math tools/check_div.hip -o tools/bin/check_div
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../zen_amd/csrc/exact_div.h"

__global__ __launch_bounds__(256) void chk_recip(unsigned long long* bad, unsigned long long* tested)
{
	unsigned long long nb = 0, nt = 0;
	for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * 256) {
		const float x = __uint_as_float((unsigned)i);
		if (!zdiv::recip_in_range(x))
			continue;
		++nt;
		const float a = zdiv::recip_exact(x), b = 1.0f / x;
		if (__float_as_uint(a) != __float_as_uint(b))
			++nb;
	}
	atomicAdd(bad, nb);
	atomicAdd(tested, nt);
}
__global__ __launch_bounds__(256) void chk_const(unsigned long long* bad, unsigned long long* tested, int c)
{
	unsigned long long nb = 0, nt = 0;
	const float cf = (float)c, r = 1.0f / cf;
	for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * 256) {
		const float x = __uint_as_float((unsigned)i);
		if (!zdiv::div_const_in_range(x))
			continue;
		++nt;
		const float a = zdiv::div_const_exact(x, cf, r), b = x / cf;
		if (__float_as_uint(a) != __float_as_uint(b))
			++nb;
	}
	atomicAdd(bad, nb);
	atomicAdd(tested, nt);
}

int main(int argc, char** argv)
{
	unsigned long long *d, h[2];
	hipMalloc((void**)&d, 16);
	unsigned long long total_bad = 0;
	hipMemset(d, 0, 16);
	hipLaunchKernelGGL(chk_recip, dim3(16384), dim3(256), 0, 0, d, d + 1);
	hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
	printf("recip_exact: %llu of %llu in-range floats differ from 1.0f / x\n", h[0], h[1]);
	total_bad += h[0];
	const int cmax = argc > 1 ? atoi(argv[1]) : 255;
	unsigned long long cb = 0, ct = 0;
	for (int c = 1; c <= cmax; ++c) {
		hipMemset(d, 0, 16);
		hipLaunchKernelGGL(chk_const, dim3(16384), dim3(256), 0, 0, d, d + 1, c);
		hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
		if (h[0])
			printf("div_const_exact: c = %d: %llu of %llu differ\n", c, h[0], h[1]);
		cb += h[0];
		ct += h[1];
	}
	printf("div_const_exact: %llu of %llu (x, c) pairs differ from x / c, c = 1..%d\n", cb, ct, cmax);
	total_bad += cb;
	printf(total_bad ? "MISMATCHES\n" : "all identical\n");
	return total_bad ? 1 : 0;
}
