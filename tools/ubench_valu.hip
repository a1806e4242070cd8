// ubench_valu.hip -- VALU issue-rate probe for the instructions the median networks are made of.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int iters)
{
	int a[8], b[8];
	for (int i = 0; i < 8; ++i) {
		a[i] = threadIdx.x * 7 + i;
		b[i] = threadIdx.x * 3 + i * 5;
	}
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int r = 0; r < 16; ++r) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				if (OP == 0)
					asm volatile("v_min_i32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 1)
					asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]), "v"(b[(i + 1) & 7]));
				else if (OP == 2)
					asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]), "v"(b[(i + 1) & 7]));
				else if (OP == 3)
					asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 4)
					asm volatile("v_max_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 5)
					asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 6)
					asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]), "v"(b[(i + 1) & 7]));
				else if (OP == 7)
					asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
				else if (OP == 8)
					asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 9)
					asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&b[i & 6]), "v"(*(double*)&b[(i + 2) & 6]));
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
	printf("%-14s blocks %5d: %.3f ms  %.2f T wave-instr/s  = %.1f T lane-ops/s  (%.2f cycles/instr/SIMD at 2.4GHz)\n",
	       name, blocks, ms, winstr / ms / 1e9, winstr * 64 / ms / 1e9, 1024 * 2.4e9 / (winstr / (ms * 1e-3)));
	hipFree(d);
}

int main()
{
	for (int blocks : {1024, 2048, 4096}) {
		run<0>("v_min_i32", blocks, 2000);
		run<1>("v_max3_i32", blocks, 2000);
		run<2>("v_med3_i32", blocks, 2000);
		run<3>("v_pk_max_i16", blocks, 2000);
		run<4>("v_max_f32", blocks, 2000);
		run<5>("v_fma_f32", blocks, 2000);
		run<6>("v_med3_f32", blocks, 2000);
		run<7>("v_mov_b32", blocks, 2000);
		run<8>("v_cndmask_b32", blocks, 2000);
		run<9>("v_pk_fma_f32", blocks, 2000);
	}
	return 0;
}
