// ubench_rowwrite.hip -- the store pattern of the analysis kernel at nfft 1024 with nothing else in it: per frame a row of
// 513 float2 (pitch 520) and a half row of 513 floats (pitch 1024), written by 32 lanes with 8- and 4-byte stores (rfft mapping:
// lane k writes bins k + 128 c and 128 (c+1) - k), 8 frames per 256-thread workgroup.  What time would a kernel with NO
// arithmetic need for the analysis kernel's output?   hipcc --offload-arch=gfx950 -O3 tools/ubench_rowwrite.hip -o tools/bin/ubench_rowwrite
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE> // 0: the kernel's stores; 1: the same bytes as 16-byte stores, one row after the other
__global__ __launch_bounds__(256) void rows(float2* S, float* mag, int n_frames, long long s_pitch, float seed)
{
	const int tid = threadIdx.x, slot = tid >> 5, k = tid & 31;
	const long long f = (long long)blockIdx.x * 8 + slot;
	if (f >= n_frames)
		return;
	float2* Sr = S + f * s_pitch;
	float* mr = mag + f * 1024;
	if (MODE == 0) {
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const int kk = k + 32 * i;
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				Sr[kk + 128 * c] = make_float2(seed + kk, seed + c);
				mr[kk + 128 * c] = seed + c;
			}
#pragma unroll
			for (int c = 4; c < 8; ++c) {
				const int bin = 128 * (8 - c) - kk;
				if (bin != 512 - 0 || kk != 0) {
					Sr[bin] = make_float2(seed + kk, seed + c);
					mr[bin] = seed + c;
				}
			}
		}
		if (k == 0) {
			Sr[512] = make_float2(seed, 0.f);
			mr[512] = seed;
		}
	}
	else {
		float4* S4 = reinterpret_cast<float4*>(Sr);
		for (int i = k; i < 257; i += 32)
			S4[i] = make_float4(seed + i, seed, seed, seed);
		float4* m4 = reinterpret_cast<float4*>(mr);
		for (int i = k; i < 129; i += 32)
			m4[i] = make_float4(seed + i, seed, seed, seed);
	}
}

int main(int argc, char** argv)
{
	const int n_frames = argc > 1 ? atoi(argv[1]) : 331456;
	const long long s_pitch = 520;
	float2* S;
	float* mag;
	hipMalloc((void**)&S, sizeof(float2) * s_pitch * n_frames);
	hipMalloc((void**)&mag, sizeof(float) * 1024 * (size_t)n_frames);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const double bytes = (double)n_frames * (513 * 8 + 513 * 4);
	for (int mode = 0; mode < 2; ++mode) {
		for (int rep = 0; rep < 3; ++rep) {
			float ms = 0;
			const int it = 20;
			for (int w = 0; w < 3; ++w)
				hipLaunchKernelGGL(mode ? rows<1> : rows<0>, dim3((n_frames + 7) / 8), dim3(256), 0, 0, S, mag, n_frames, s_pitch, 1.0f);
			hipEventRecord(e0);
			for (int i = 0; i < it; ++i)
				hipLaunchKernelGGL(mode ? rows<1> : rows<0>, dim3((n_frames + 7) / 8), dim3(256), 0, 0, S, mag, n_frames, s_pitch, 1.0f + i);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			hipEventElapsedTime(&ms, e0, e1);
			printf("{\"mode\": \"%s\", \"frames\": %d, \"ms\": %.4f, \"GBps\": %.0f}\n", mode ? "16-byte stores" : "the kernel's 8- and 4-byte stores", n_frames,
			       ms / it, bytes / (ms / it * 1e-3) / 1e9);
		}
	}
	return 0;
}
