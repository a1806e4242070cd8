// row_load.h -- staging loads shared by the frequency-direction median kernels (median_net.hip, median_big.hip).
#pragma once
#include "median_net.h"

namespace zen_hip_impl {

using znet::to_key;

// Four consecutive logical columns vc..vc+3 (vc a multiple of 4, the row 16-byte aligned and a multiple of 4
// long) of a source row as ordering keys: replicate border (ippBorderRepl) outside [0, cols); `herm`: only
// columns 0..cols/2 are stored, column c > cols/2 is column cols - c (FilterArgs::hermitian).
template <bool NONNEG>
__device__ __forceinline__ int4 row_vec_keys(const float* __restrict__ srow, int vc, int cols, int herm)
{
	if (herm && vc >= (cols >> 1)) {
		if (vc >= cols) { // beyond the row: column cols-1, which is column 1
			const int b = to_key<NONNEG>(srow[1]);
			return make_int4(b, b, b, b);
		}
		// columns cols-vc, cols-vc-1, cols-vc-2, cols-vc-3: one aligned vector and the scalar above it
		const int mc = cols - vc;
		const float4 v = *reinterpret_cast<const float4*>(srow + mc - 4);
		return make_int4(to_key<NONNEG>(srow[mc]), to_key<NONNEG>(v.w), to_key<NONNEG>(v.z), to_key<NONNEG>(v.y));
	}
	const int vcl = vc < 0 ? 0 : (vc > cols - 4 ? cols - 4 : vc);
	const float4 x = *reinterpret_cast<const float4*>(srow + vcl);
	int4 k = make_int4(to_key<NONNEG>(x.x), to_key<NONNEG>(x.y), to_key<NONNEG>(x.z), to_key<NONNEG>(x.w));
	if (vc < 0)
		k = make_int4(k.x, k.x, k.x, k.x);
	else if (vc >= cols)
		k = make_int4(k.w, k.w, k.w, k.w);
	return k;
}


} // namespace zen_hip_impl
