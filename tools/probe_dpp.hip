// probe_dpp.hip -- semantics of the gfx9 wave-wide DPP shifts and of v_cndmask with an SGPR-pair mask.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out)
{
	const int lane = threadIdx.x;
	const int x = 100 + lane;
	int shr = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, false); // wave_shr:1
	int shl = __builtin_amdgcn_update_dpp(-2, x, 0x130, 0xf, 0xf, false); // wave_shl:1
	unsigned long long m = 0xF0F0F0F0F0F0F0F0ull;
	int sel;
	asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(sel) : "v"(1), "v"(2), "s"(m));
	out[lane] = shr;
	out[64 + lane] = shl;
	out[128 + lane] = sel;
}
int main()
{
	int *d, h[192];
	hipMalloc(&d, sizeof(h));
	k<<<1, 64>>>(d);
	hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	printf("wave_shr:1 lanes 0,1,2,31,32,63: %d %d %d %d %d %d\n", h[0], h[1], h[2], h[31], h[32], h[63]);
	printf("wave_shl:1 lanes 0,1,30,31,62,63: %d %d %d %d %d %d\n", h[64], h[65], h[94], h[95], h[126], h[127]);
	printf("cndmask sgpr mask lanes 0..7: %d %d %d %d %d %d %d %d\n", h[128], h[129], h[130], h[131], h[132], h[133], h[134], h[135]);
	return 0;
}
