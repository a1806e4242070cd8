// exact_div.h -- IEEE-exact float divisions in a third of the instructions, for the two shapes the SSE path is made of.
//
// The reference's SSE filter divides all the time (hps.h:45-56 reciprocal_functor: (1 / x) * k; the box mean's sum / length,
// box.h:266-286; hps.cu:599-604), and the oracle's results are those of IEEE division.  The compiler's division (no fast math)
// is v_div_scale x2, v_rcp, five fma, v_div_fmas, v_div_fixup plus two mode switches: ~15 issue slots.  Two shapes need less:
//   * 1 / x:  z0 = v_rcp_f32(x) (1 ulp), e = fma(-x, z0, 1), z = fma(e, z0, z0)            -- Markstein's reciprocal refinement
//   * x / c, c a small integer constant (a mask length):  r = RN(1 / c) once; q0 = x r, e = fma(-q0, c, x), q = fma(e, r, q0)
// Both are correctly rounded wherever nothing under- or overflows on the way, which the range tests below guarantee -- and that
// they equal the compiler's division for EVERY float in range (and every c = 1..255) is not argued but checked, exhaustively,
// on the GPU: tools/check_div.hip, run by tests/test_gpu_round5.py.  Out-of-range operands (zeros, infinities, NaNs, values
// next to the denormals) take the compiler's division: the callers test a batch at once and branch wave-uniformly.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace zdiv {

// 2^-126 <= |x| <= 2^126: x and 1/x are normal, the refinement's products are too
__device__ __forceinline__ bool recip_in_range(float x)
{
	return ((__float_as_uint(x) & 0x7fffffffu) - 0x00800000u) <= (0x7e800000u - 0x00800000u);
}
__device__ __forceinline__ float recip_exact(float x)
{
	const float z0 = __builtin_amdgcn_rcpf(x);
	const float e = __builtin_fmaf(-x, z0, 1.0f);
	return __builtin_fmaf(e, z0, z0);
}

// 2^-100 <= |x| <= 2^126: x / c (c <= 255) and the remainder x - q c are normal
__device__ __forceinline__ bool div_const_in_range(float x)
{
	return ((__float_as_uint(x) & 0x7fffffffu) - 0x0d800000u) <= (0x7e800000u - 0x0d800000u);
}
// r = 1.0f / c (the caller computes it once per kernel)
__device__ __forceinline__ float div_const_exact(float x, float c, float r)
{
	const float q0 = x * r;
	const float e = __builtin_fmaf(-q0, c, x);
	return __builtin_fmaf(e, r, q0);
}

// K values at once.  The short form for all of them; if ANY lane of the wavefront holds an operand outside the range the short
// form is proven for, the whole batch is done again with the compiler's division (identical where the short form applies,
// IEEE everywhere else): a wave-uniform branch that silence, infinities and the edge of the denormals take, and music does not.
template <int K>
__device__ __forceinline__ void recip_batch(const float (&x)[K], float (&z)[K])
{
	bool bad = false;
#pragma unroll
	for (int i = 0; i < K; ++i) {
		z[i] = recip_exact(x[i]);
		bad = bad || !recip_in_range(x[i]);
	}
	if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
#pragma unroll
		for (int i = 0; i < K; ++i)
			z[i] = 1.0f / x[i];
	}
}
template <int K>
__device__ __forceinline__ void div_const_batch(const float (&x)[K], float c, float r, float (&q)[K])
{
	bool bad = false;
#pragma unroll
	for (int i = 0; i < K; ++i) {
		q[i] = div_const_exact(x[i], c, r);
		bad = bad || !div_const_in_range(x[i]);
	}
	if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
#pragma unroll
		for (int i = 0; i < K; ++i)
			q[i] = x[i] / c;
	}
}

} // namespace zdiv
