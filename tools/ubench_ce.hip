// ubench_ce.hip -- compare-exchange formulations on gfx950: v_min+v_max vs v_cmp+2*v_cndmask.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int iters)
{
	int a[16];
	for (int i = 0; i < 16; ++i)
		a[i] = (threadIdx.x * 2654435761u + i * 40503u) >> 3;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int r = 0; r < 8; ++r) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const int p = i, q = 15 - ((i + r) & 7);
				int x = a[p], y = a[q];
				if (OP == 0) {
					a[p] = min(x, y);
					a[q] = max(x, y);
				}
				else {
					int lo, hi;
					asm volatile("v_cmp_gt_i32 vcc, %2, %3\n\tv_cndmask_b32 %0, %2, %3, vcc\n\tv_cndmask_b32 %1, %3, %2, vcc"
					             : "=&v"(lo), "=&v"(hi)
					             : "v"(x), "v"(y)
					             : "vcc");
					a[p] = lo;
					a[q] = hi;
				}
			}
		}
	}
	int s = 0;
	for (int i = 0; i < 16; ++i)
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
	double ce = (double)blocks * 4 * iters * 64.0; // wave-level compare-exchanges
	printf("%-16s blocks %5d: %.3f ms  %.3f T wave-CE/s (%.2f cycles per CE per SIMD at 2.4 GHz)\n", name, blocks, ms,
	       ce / ms / 1e9, 1024 * 2.4e9 / (ce / (ms * 1e-3)));
	hipFree(d);
}

int main()
{
	for (int blocks : {2048, 4096}) {
		run<0>("min+max", blocks, 2000);
		run<1>("cmp+2cndmask", blocks, 2000);
	}
	return 0;
}
