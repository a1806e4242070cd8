// median47_core.h -- the 47-tap block scheme shared by median47_dpp_kernel (median47.hip) and the fused
// causal kernel (rt_fused.hip).  See median47.hip for the derivation.
//
// Row image in LDS (`img`): 16-word chunks 20 words apart; chunk c holds block B(c-1) = x[16(c-1)-8 ..
// 16(c-1)+7], i.e. column k sits at word k + 24; the 24 words below column 0 and the 24 above column
// 16*n_blk - 1 hold the replicate border (ippBorderRepl).  The thread with block number `blk` produces
// the medians of columns 16*blk .. 16*blk+15.  Within a wave consecutive lanes must hold consecutive
// blocks (lanes whose neighbours are not their blocks' neighbours produce garbage: the fused kernel uses
// that to park unneeded lanes); across waves the pieces travel through `edge` (64 words per wave):
//   [0,16) pieces of the block left of lane 0; [16,32) sorted block right of lane 63, [32,48) its pieces;
//   [48,64) pieces of the block after that.
// Call m47_sort_and_publish, then __syncthreads(), then m47_select.
#pragma once
#include "median_net.h"

namespace zm47 {

constexpr int RSTR = 20; // 16 words + 4 pad per chunk (5c mod 16 distinct: ds_read_b128 is conflict free)

__device__ __forceinline__ int dpp_from_next(int old, int v) // lane i <- lane i+1; lane 63 keeps `old`
{
	return __builtin_amdgcn_update_dpp(old, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_from_prev(int old, int v) // lane i <- lane i-1; lane 0 keeps `old`
{
	return __builtin_amdgcn_update_dpp(old, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

struct Pieces {
	int s16[16]; // the block sorted
	int up[16];  // lower half, 1st and 3rd quarter sorted: wanted by the two lanes to the left
	int dn[16];  // upper half, 2nd and 4th quarter sorted: wanted by the lane to the right
};

// Sorts the thread's own block and writes the wave-edge records.  `wave` is the edge record of this wave,
// `first` / `last`: this thread holds the first / last block of the row (n_blk - 1), whose outer
// neighbours are border replicas.
__device__ __forceinline__ void m47_sort_and_publish(const int* img, int (*edge)[64], int blk, int lane, int wave,
                                                     bool first, bool last, Pieces& pc)
{
	int oct[16], quad[16];
	{
		int raw[16];
		znet::lds_load<16>(&img[(blk + 1) * RSTR], raw);
		znet::pyramid16(raw, pc.s16, oct, quad);
	}
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		pc.up[i] = oct[i];
		pc.dn[i] = oct[8 + i];
	}
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		pc.up[8 + i] = quad[i];
		pc.up[12 + i] = quad[8 + i];
		pc.dn[8 + i] = quad[4 + i];
		pc.dn[12 + i] = quad[12 + i];
	}
	if (wave > 0 && lane < 2) {
		znet::lds_store<16>(&edge[wave - 1][32 + 16 * lane], pc.up);
		if (lane == 0)
			znet::lds_store<16>(&edge[wave - 1][16], pc.s16);
	}
	if (wave < 3 && lane == 63)
		znet::lds_store<16>(&edge[wave + 1][0], pc.dn);
	if (first) { // B(-1) = 16 copies of x[0]
		const int b = img[0];
		const int4 q = make_int4(b, b, b, b);
#pragma unroll
		for (int i = 0; i < 4; ++i)
			*reinterpret_cast<int4*>(&edge[wave][4 * i]) = q;
	}
	if (last) { // B(n) = the last eight samples and eight copies of c = the last one; B(n+1) = 16 copies of c
		int w[8];
		znet::lds_load<8>(&img[(blk + 2) * RSTR], w);
		const int c = w[7];
		znet::oe_merge<2, 0>(w);
		znet::oe_merge<2, 2>(w);
		znet::oe_merge<2, 4>(w);
		znet::oe_merge<2, 6>(w);
		znet::oe_merge<4, 0>(w);
		znet::oe_merge<4, 4>(w);
		int p[16], s[16];
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			p[8 + i] = w[i]; // 1st quarter sorted
			p[12 + i] = c;   // 3rd quarter
		}
		znet::oe_merge<8, 0>(w);
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			p[i] = w[i];             // lower half sorted
			s[i] = min(w[i], c);     // the block sorted: the eight samples with eight copies of c spliced in
			s[8 + i] = max(w[i], c);
		}
		znet::lds_store<16>(&edge[wave][16], s);
		znet::lds_store<16>(&edge[wave][32], p);
		const int4 q = make_int4(c, c, c, c);
#pragma unroll
		for (int i = 0; i < 4; ++i)
			*reinterpret_cast<int4*>(&edge[wave][48 + 4 * i]) = q;
	}
}

// After the barrier: the neighbours' pieces by DPP shifts (the wave's edge record as `old`), then the
// selection tree.  out[g] = median of columns 16*blk + g - 23 .. 16*blk + g + 23.
__device__ __forceinline__ void m47_select(const int* img, const int (*edge)[64], int blk, int wave, const Pieces& pc,
                                           int (&out)[16])
{
	znet::Shared47 sh;
	{
		const int* ed = edge[wave];
		int eb[16], B[16];
		znet::lds_load<16>(ed + 16, eb);
#pragma unroll
		for (int i = 0; i < 16; ++i)
			B[i] = dpp_from_next(eb[i], pc.s16[i]);
		znet::mid16_of_two_sorted16(pc.s16, B, sh.cand);
		int e0[16], e1[16], el[16], hi[16], lo[16];
		znet::lds_load<16>(ed + 32, e0);
		znet::lds_load<16>(ed + 48, e1);
		znet::lds_load<16>(ed, el);
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int x1 = dpp_from_next(e0[i], pc.up[i]); // pieces of B(t+1)
			hi[i] = dpp_from_next(e1[i], x1);              // pieces of B(t+2)
			lo[i] = dpp_from_prev(el[i], pc.dn[i]);        // pieces of B(t-1)
		}
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			sh.lo_oct[i] = lo[i];
			sh.hi_oct[i] = hi[i];
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			sh.lo_q[0][i] = lo[8 + i];
			sh.lo_q[1][i] = lo[12 + i];
			sh.hi_q[0][i] = hi[8 + i];
			sh.hi_q[1][i] = hi[12 + i];
		}
	}
	znet::lds_load<16>(&img[blk * RSTR], sh.lo_raw);       // B(t-1) as it stands
	znet::lds_load<16>(&img[(blk + 3) * RSTR], sh.hi_raw); // B(t+2)
	znet::medians47_shared(sh, out);
}

} // namespace zm47
