// ubench_valu2.hip -- issue rates of the float / double instructions the FFT and |z| code is made of
// (companion of ubench_valu.hip).  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu2.hip -o tools/bin/ubench_valu2
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int iters)
{
	int a[8], b[8];
	for (int i = 0; i < 8; ++i) {
		a[i] = threadIdx.x * 7 + i + 0x3f800000;
		b[i] = threadIdx.x * 3 + i * 5 + 0x3f000000;
	}
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int r = 0; r < 16; ++r) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				if (OP == 0)
					asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 1)
					asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 2)
					asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 3)
					asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 4)
					asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 5)
					asm volatile("v_rcp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
				else if (OP == 6)
					asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&b[i & 6]), "v"(*(double*)&b[(i + 2) & 6]));
				else if (OP == 7)
					asm volatile("v_mul_f64 %0, %1, %2" : "=v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[i & 6]), "v"(*(double*)&b[(i + 2) & 6]));
				else if (OP == 8)
					asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(*(double*)&a[i & 6]) : "v"(b[i]));
				else if (OP == 9)
					asm volatile("v_rsq_f64 %0, %1" : "=v"(*(double*)&a[i & 6]) : "v"(*(double*)&b[i & 6]));
				else if (OP == 10)
					asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(*(double*)&b[i & 6]));
				else if (OP == 11)
					asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 12)
					asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
				else if (OP == 13)
					asm volatile("v_div_scale_f32 %0, vcc, %1, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(a[i]) : "vcc");
				else if (OP == 14)
					asm volatile("v_sqrt_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
			}
		}
	}
	int s = 0;
	for (int i = 0; i < 8; ++i)
		s += a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int blocks, int iters)
{
	int* d;
	hipMalloc(&d, sizeof(int) * blocks * 256);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	k<OP><<<blocks, 256>>>(d, 10);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	k<OP><<<blocks, 256>>>(d, iters);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	double winstr = (double)blocks * 4 * iters * 128.0; // wave-instructions
	printf("%-16s blocks %5d: %.3f ms  %.3f T wave-instr/s  (%.2f ns per instr per SIMD)\n", name, blocks, ms,
	       winstr / ms / 1e9, 1024.0 * ms * 1e6 / winstr);
	hipFree(d);
}

int main()
{
	const int blocks = 4096, it = 1000;
	run<0>("v_add_f32", blocks, it);
	run<1>("v_mul_f32", blocks, it);
	run<2>("v_sub_f32", blocks, it);
	run<3>("v_add_u32", blocks, it);
	run<4>("v_and_b32", blocks, it);
	run<11>("v_xor_b32", blocks, it);
	run<5>("v_rcp_f32", blocks, it);
	run<14>("v_sqrt_f32", blocks, it);
	run<12>("v_cmp_lt_f32", blocks, it);
	run<13>("v_div_scale_f32", blocks, it);
	run<6>("v_fma_f64", blocks, it);
	run<7>("v_mul_f64", blocks, it);
	run<8>("v_cvt_f64_f32", blocks, it);
	run<10>("v_cvt_f32_f64", blocks, it);
	run<9>("v_rsq_f64", blocks, it);
	return 0;
}
