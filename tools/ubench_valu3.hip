// ubench_valu3.hip -- round-2 issue-rate probes for the median networks (companion of ubench_valu.hip):
//   * the gfx950-only v_minimum3_f32 / v_maximum3_f32, unsigned and 16-bit min/max, v_max_f64
//   * cross-lane movement: v_mov_b32_dpp (wave_shl:1, row_shr:1), DPP fused into v_min_i32, ds_bpermute_b32
//   * MIXES: do half-rate min/max and full-rate fma/mov share one issue budget (time = sum) or overlap?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu3.hip -o tools/bin/ubench_valu3 ; run on the GPU box.
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
	const int bp = ((threadIdx.x + 2) & 63) * 4;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int r = 0; r < 16; ++r) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				if (OP == 0)
					asm volatile("v_min_i32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 1)
					asm volatile("v_minimum3_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]), "v"(b[(i + 1) & 7]));
				else if (OP == 2)
					asm volatile("v_maximum3_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]), "v"(b[(i + 1) & 7]));
				else if (OP == 3)
					asm volatile("v_min_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 4)
					asm volatile("v_min_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 5)
					asm volatile("v_max_f64 %0, %1, %2" : "=v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[i & 6]), "v"(*(double*)&b[i & 6]));
				else if (OP == 6)
					asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
				else if (OP == 7)
					asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
				else if (OP == 8)
					asm volatile("v_min_i32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
				else if (OP == 9)
					asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(a[i]) : "v"(bp), "v"(b[i]));
				else if (OP == 10) { // mix: min + fma alternating (4 + 4 per 8)
					if (i & 1)
						asm volatile("v_min_i32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
					else
						asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[i]), "v"(b[i]));
				}
				else if (OP == 11) { // mix: min + mov_dpp alternating
					if (i & 1)
						asm volatile("v_min_i32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
					else
						asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
				}
				else if (OP == 12) { // mix: 3 min : 1 fma
					if ((i & 3) != 3)
						asm volatile("v_min_i32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
					else
						asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[i]), "v"(b[i]));
				}
				else if (OP == 13)
					asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 14)
					asm volatile("v_cmp_gt_i32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b[i]) : "s20", "s21");
				else if (OP == 15)
					asm volatile("v_cndmask_b32 %0, %1, %2, s[20:21]" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 16)
					asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]), "v"(b[(i + 1) & 7]));
				else if (OP == 17)
					asm volatile("v_pk_min_i16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 18)
					asm volatile("v_sub_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
				else if (OP == 19)
					asm volatile("v_max_i32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "s"(it));
				else if (OP == 20) // 2 waves: LDS-pipe traffic next to min (overlap test): bpermute + min
				{
					if (i & 1)
						asm volatile("v_min_i32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
					else
						asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(a[i]) : "v"(bp), "v"(b[i]));
				}
			}
		}
		if (OP == 9 || OP == 20)
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
	printf("%-28s blocks %5d: %8.3f ms  %.3f T wave-instr/s  (%.2f ns per instr per SIMD)\n", name, blocks, ms,
	       winstr / ms / 1e9, 1024.0 * ms * 1e6 / winstr);
	hipFree(d);
}

int main()
{
	const int B = 4096, I = 1000;
	run<0>("v_min_i32", B, I);
	run<13>("v_fma_f32", B, I);
	run<1>("v_minimum3_f32", B, I);
	run<2>("v_maximum3_f32", B, I);
	run<3>("v_min_u32", B, I);
	run<4>("v_min_u16", B, I);
	run<5>("v_max_f64", B, I);
	run<16>("v_med3_i32", B, I);
	run<17>("v_pk_min_i16", B, I);
	run<18>("v_sub_u32", B, I);
	run<19>("v_max_i32 (sgpr operand)", B, I);
	run<6>("v_mov_b32_dpp wave_shl:1", B, I);
	run<7>("v_mov_b32_dpp row_shr:1", B, I);
	run<8>("v_min_i32_dpp wave_shl:1", B, I);
	run<9>("ds_bpermute_b32", B, I);
	run<14>("v_cmp_gt_i32 -> sgpr pair", B, I);
	run<15>("v_cndmask_b32 (sgpr mask)", B, I);
	run<10>("mix 1 min : 1 fma", B, I);
	run<12>("mix 3 min : 1 fma", B, I);
	run<11>("mix 1 min : 1 mov_dpp", B, I);
	run<20>("mix 1 min : 1 ds_bpermute", B, I);
	return 0;
}
