// masks.h -- the element-wise mask functors of the reference (libzen/hps.h:35-43, :100-140), shared by
// the synthesis kernel (stft.hip) and the fused realtime kernel (rt_fused.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstring>

#pragma clang fp contract(off)

namespace zen_hip_impl {

struct MaskCfg {
	float beta, beta_h;
	int soft, power, sse, out_h, out_p;
};

__device__ __forceinline__ float powi(float x, int p) // oracle powi(): repeated multiplication
{
	if (p <= 0)
		return 1.0F;
	float r = x;
	for (int i = 1; i < p; ++i)
		r = r * x;
	return r;
}

__device__ __forceinline__ float pmask_value(float h, float p, const MaskCfg& c)
{
	const float EPS = FLT_EPSILON;
	if (c.sse) // sse_mask_functor hps.h:132-140
		return p * p / (p * p + h * h + EPS);
	if (c.soft) { // soft_mask_functor hps.h:116-129
		const float xp = powi(p, c.power), yp = powi(h, c.power);
		return xp / (xp + yp + EPS);
	}
	return (float)((p / (h + EPS)) >= c.beta); // hard_mask_functor hps.h:100-113
}

__device__ __forceinline__ float hmask_value(float h, float p, const MaskCfg& c)
{
	const float EPS = FLT_EPSILON;
	if (c.sse)
		return h * h / (h * h + p * p + EPS);
	if (c.soft) {
		const float xp = powi(h, c.power), yp = powi(p, c.power);
		return xp / (xp + yp + EPS);
	}
	return (float)((h / (p + EPS)) >= c.beta_h);
}

// hard_mask_functor (hps.h:100-113) without the division.  With IEEE round-to-nearest-even division,
// fl(x / d) >= beta exactly when the real quotient x / d reaches the rounding boundary below beta: the
// midpoint m of pred(beta) and beta if beta's significand is even (a tie rounds up to beta), else anything
// above it.  m has 25 significant bits and d 24, so m * d is exact in double (49 bits) and (double)x >= m * (double)d
// resp. > is an exact test.  The inclusive case is folded into the threshold on the host (hard_mask_threshold):
// x and m*d are both multiples of the last of those 49 bits wherever they are close, so x >= m*d  <=>
// x > m*d*(1 - 2^-50), and the device compares strictly in both cases.  d = +inf gives fl(x / d) = 0 or NaN, never
// >= beta > 0: the threshold is +inf then, which no x exceeds under a STRICT comparison (while the comparison was
// inclusive an extra fma(t, 0, t) turned it into a NaN; gone with it: one double-precision instruction per mask).
// NaNs compare false on both sides.  Valid for normal positive beta (hard_mask_threshold() returns 0 otherwise: divide).
__device__ __forceinline__ float hard_mask_exact(float x, float d, double thr, bool /*unused*/ = false)
{
	const double t = thr * (double)d;
	return (double)x > t ? 1.0F : 0.0F;
}

// which: 0 percussive, 1 harmonic, 2 residual.  `which` and the cfg flags are wave-uniform, so only the
// division(s) the requested output needs are executed.
__device__ __forceinline__ float mask_value(int which, float h, float p, const MaskCfg& c)
{
	if (which == 0)
		return pmask_value(h, p, c);
	if (which == 1)
		return hmask_value(h, p, c);
	const float hm = c.out_h ? hmask_value(h, p, c) : 0.0f;
	const float pm = c.out_p ? pmask_value(h, p, c) : 0.0f;
	return 1 - (hm + pm); // residual_mask_functor hps.h:35-43
}

// Both hard masks by exact comparison (hard_mask_exact): thr_p for (P / (H + Eps)) >= beta, thr_h for
// (H / (P + Eps)) >= beta - Eps (hps.cu:501-505, :535-540).  A zero threshold keeps the division.
struct HardThr {
	double p, h;
	int p_inc, h_inc;
};

__device__ __forceinline__ float pmask_thr(float h, float p, const MaskCfg& c, const HardThr& t)
{
	if (!c.sse && !c.soft && t.p != 0.0)
		return hard_mask_exact(p, h + FLT_EPSILON, t.p, t.p_inc != 0);
	return pmask_value(h, p, c);
}
__device__ __forceinline__ float hmask_thr(float h, float p, const MaskCfg& c, const HardThr& t)
{
	if (!c.sse && !c.soft && t.h != 0.0)
		return hard_mask_exact(h, p + FLT_EPSILON, t.h, t.h_inc != 0);
	return hmask_value(h, p, c);
}
// mask_value with the divisions of the hard masks replaced where the thresholds allow
__device__ __forceinline__ float mask_value_thr(int which, float h, float p, const MaskCfg& c, const HardThr& t)
{
	if (which == 0)
		return pmask_thr(h, p, c, t);
	if (which == 1)
		return hmask_thr(h, p, c, t);
	const float hm = c.out_h ? hmask_thr(h, p, c, t) : 0.0f;
	const float pm = c.out_p ? pmask_thr(h, p, c, t) : 0.0f;
	return 1 - (hm + pm); // residual_mask_functor hps.h:35-43
}

// host side of hard_mask_exact: the boundary and whether it belongs to the "true" side
inline double hard_mask_threshold(float beta, int* inclusive)
{
	unsigned u;
	memcpy(&u, &beta, sizeof(u));
	const unsigned expo = (u >> 23) & 0xffu;
	if ((u >> 31) || expo == 0 || expo == 0xffu) // negative, zero, subnormal, inf, NaN: keep the divide
		return 0.0;
	const unsigned up = u - 1; // pred(beta): the next float below (beta > FLT_MIN's pred is subnormal: still exact)
	float pred;
	memcpy(&pred, &up, sizeof(pred));
	*inclusive = 0; // (kept for the callers' structs: the inclusive case is folded into the value below)
	const double m = ((double)pred + (double)beta) * 0.5;
	// a tie rounds to the even significand: beta even -> the boundary itself counts -> compare against a threshold
	// a quarter of the 49-bit product's last place lower (see hard_mask_exact)
	return (u & 1u) == 0 ? m * (1.0 - 0x1p-50) : m;
}
// both thresholds of an engine (beta for the percussive mask, beta - Eps for the harmonic one)
inline HardThr hard_mask_thresholds(float beta, float beta_h, bool divide)
{
	HardThr t{0.0, 0.0, 0, 0};
	if (!divide) {
		t.p = hard_mask_threshold(beta, &t.p_inc);
		t.h = hard_mask_threshold(beta_h, &t.h_inc);
	}
	return t;
}

} // namespace zen_hip_impl
