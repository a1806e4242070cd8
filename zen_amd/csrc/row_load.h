// row_load.h -- staging loads shared by the frequency-direction median kernels (median_net.hip, median_big.hip).
#pragma once
#include "bounds.h"
#include "median_net.h"

namespace zen_hip_impl {

using znet::to_key;

// Four consecutive logical columns vc..vc+3 (vc a multiple of 4, the row 16-byte aligned and a multiple of 4
// long) of a source row as ordering keys: replicate border (ippBorderRepl) outside [0, cols); `herm`: only
// columns 0..cols/2 are stored, column c > cols/2 is column cols - c (FilterArgs::hermitian).
// Branch-free: ONE 16-byte load from a selected address, the vector assembled by selects.  (Written with a branch per
// case -- lower half / mirrored / beyond the row -- every call became a load with a full wait inside its branch, and a
// staging loop of six such calls six dependent trips to memory.)
typedef float float4_u __attribute__((ext_vector_type(4), aligned(4))); // (the mirrored vector starts at any column)
template <bool NONNEG>
__device__ __forceinline__ int4 row_vec_keys(const float* __restrict__ srow, int vc, int cols, int herm)
{
	const bool up = herm && vc >= (cols >> 1); // not stored: the mirror image, or beyond the row
	const bool mir = up && vc < cols;
	const int mc = cols - vc;                  // mirrored: columns mc, mc-1, mc-2, mc-3
	const int vcl = vc < 0 ? 0 : (vc > cols - 4 ? cols - 4 : vc);
	const int va = up ? (mir ? mc - 3 : 0) : vcl;
	// (streaming: every sample is staged by one workgroup and its halo by the next; nothing gains from staying in a cache)
	const float* q = srow + va;
	ZH_CHK(q, 4);
	const float x0 = __builtin_nontemporal_load(q), x1 = __builtin_nontemporal_load(q + 1), x2 = __builtin_nontemporal_load(q + 2),
	            x3 = __builtin_nontemporal_load(q + 3);
	const int k0 = to_key<NONNEG>(x0), k1 = to_key<NONNEG>(x1), k2 = to_key<NONNEG>(x2), k3 = to_key<NONNEG>(x3);
	// stored: (k0 k1 k2 k3), left of the row k0 x4, right of it k3 x4; mirrored: (k3 k2 k1 k0); Hermitian and beyond the
	// row: column cols-1, which is column 1: k1 x4
	const bool lo = vc < 0, hi = !up && vc >= cols, rep = lo || hi || (up && !mir);
	const int r = lo ? k0 : (hi ? k3 : k1); // the replicated key where there is one
	return make_int4(rep ? r : (mir ? k3 : k0), rep ? r : (mir ? k2 : k1), rep ? r : (mir ? k1 : k2), rep ? r : (mir ? k0 : k3));
}


} // namespace zen_hip_impl
