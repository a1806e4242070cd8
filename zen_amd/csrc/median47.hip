// median47.hip -- the 47-tap frequency-direction median of the headline config (hop 1024: l_perc = 46,
// mask 47, libzen/hps.h:229 + libzen/mfilt.h:89) on whole 4096-bin rows: median47_dpp_kernel.
// Same results as median_net_freq_kernel<47> (and as MedianFilterCPU, libzen/mfilt.h:270-342).
//
// Why a second kernel.  On gfx950 v_min/v_max/v_min3/v_max3/v_med3 issue at half rate (~1.7 ns per wave
// instruction per SIMD against 1.0 ns for v_mov/v_add/v_fma, profiles/r01_ubench_valu_rates.txt), and the
// generic kernel is bound by exactly those (profiles/r01_c_median47_pmc.json).  With 47 taps and 16 outputs
// per thread every chunk the selection tree of median_net.h consumes is an aligned dyadic piece of the row
// (mid + 1 = 24 is a multiple of 8):
//
//   B(u) = x[16u-8 .. 16u+7]                       one 16-sample block per thread u
//   thread t needs  B(t), B(t+1) sorted             (the 32 samples common to its 16 windows)
//                   upper half / 2nd, 4th quarter of B(t-1), sorted;  B(t-1) raw (pairs, singles)
//                   lower half / 1st, 3rd quarter of B(t+2), sorted;  B(t+2) raw
//
// and the sorted halves and quarters of a block are the intermediate stages of its own Batcher merge sort
// (znet::pyramid16).  So every thread sorts ONE block (63 comparators), and everything else it needs
// sorted arrives from lanes t-1, t+1, t+2 through wave-wide DPP shifts (v_mov_b32_dpp wave_shl/shr:1, full
// rate, no LDS): 64 moves replace 58 comparators (116 half-rate instructions) of re-sorting.  The three
// lanes at a wave's ends take their neighbours' pieces from a 256-byte LDS edge record per wave, passed as
// the `old` operand of the same DPP moves (a lane whose DPP source lies outside the wave keeps `old`), so
// there is no lane-dependent code at all.  Blocks that hang over the row ends are (partly) border replicas
// (ippBorderRepl): B(-1) and B(257) are constant, B(256) is eight samples and eight copies of the last one,
// whose sorted pieces cost 19 comparators + 16 min/max in thread 255 instead of a 63-comparator sort.
//
// Per thread of 16 outputs: 126 (own sort) + 96 (middle 16 of two sorted blocks) + 304 (selection tree
// below the shared pieces) = 526 min/max/med3 against 642 + 126/4 in median_net_freq_kernel<47>, and a
// prologue without integer division, clamps or branches (whole rows only; other geometries use the generic
// kernel).  HBM traffic is unchanged: 4 B read + 4 B written per element.
#include "common.h"
#include "bounds.h"
#include "filters.h"
#include "median47_core.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

constexpr int NCH = 259;  // image chunks: blocks u = -1 .. 257  <->  chunk c = u + 1
using zm47::RSTR;
constexpr int COLS = 4096;

using znet::from_key;
using znet::to_key;

struct M47Args {
	const float* src;
	float* dst;
	long long src_stream_stride, dst_stream_stride;
	int row_base, ring; // source row of output row r: (row_base + r) % ring, with row_base < ring, r < ring
	int herm;           // FilterArgs::hermitian: bins 0..2048 stored, bins 0..2048 and 4073..4095 wanted
};

// Streaming accesses: every input is read once and every output written once, so both bypass the caches'
// retention ("nt"): the plain copy of this shape runs 154 us, the nontemporal one 139 us
// (profiles/r02_ubench_copy.txt).
__device__ __forceinline__ float4 load_nt(const float* p)
{
	ZH_CHK(p, 4);
	return make_float4(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2),
	                   __builtin_nontemporal_load(p + 3));
}
__device__ __forceinline__ void store_nt(float* p, float4 v)
{
	ZH_CHK(p, 4);
	__builtin_nontemporal_store(v.x, p);
	__builtin_nontemporal_store(v.y, p + 1);
	__builtin_nontemporal_store(v.z, p + 2);
	__builtin_nontemporal_store(v.w, p + 3);
}

__device__ __forceinline__ int* unit_ptr(int* img, int u) // 16-byte unit u of the row image (word 4u)
{
	return img + (u >> 2) * RSTR + (u & 3) * 4;
}

// VARIANT 0: results are transposed through the LDS image and stored 1 KB-contiguous per wave instruction.
// VARIANT 1: every thread stores its 16 consecutive results straight to global memory (four 16-byte stores,
//            64 bytes apart between lanes): two barriers and eight LDS instructions fewer (measured slower).
// VARIANT 2, 3 (diagnostics for tools/bench_median.py, results are NOT medians): 2 = the data movement
//            alone (HBM -> image -> transposed store, no sorting), 3 = everything but the global stores.
template <bool NONNEG, int VARIANT, bool HERM = false>
__global__ __launch_bounds__(256) void median47_dpp_kernel(M47Args p)
{
	constexpr bool DIRECT = VARIANT == 1;
	constexpr bool NT = VARIANT != 4; // VARIANT 4 (diagnostic): the default kernel with plain loads / stores
	__shared__ __attribute__((aligned(16))) int img[NCH * RSTR];
	// per wave: [0,16) pieces of the block left of lane 0; [16,32) sorted block right of lane 63,
	// [32,48) its pieces; [48,64) pieces of the block after that
	__shared__ __attribute__((aligned(16))) int edge[4][64];
	__shared__ __attribute__((aligned(16))) int negf[4]; // !NONNEG: wave w loaded a sample with its sign bit set

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int row = blockIdx.x;
	int srow_idx = p.row_base + row; // < 2 * ring: one conditional subtraction instead of a division
	if (srow_idx >= p.ring)
		srow_idx -= p.ring;
	const float* __restrict__ srow = p.src + (long long)blockIdx.y * p.src_stream_stride + (long long)srow_idx * COLS;
	float* __restrict__ drow = p.dst + (long long)blockIdx.y * p.dst_stream_stride + (long long)row * COLS;

	// ---- HBM -> LDS image.  Column c is image word c + 24; thread tid holds columns 4*tid + 1024*i.
	{
		float4 x[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			if (HERM && i >= 2) { // columns 2048 + ...: the mirror image of the stored half, back to front
				const int mc = COLS - (4 * tid + 1024 * i); // columns mc, mc-1, mc-2, mc-3
				ZH_CHK(srow + mc - 4, 5);
				const float4 v = *reinterpret_cast<const float4*>(srow + mc - 4);
				x[i] = make_float4(srow[mc], v.w, v.z, v.y);
			}
			else {
				ZH_CHK(srow + 4 * tid + 1024 * i, 4);
				x[i] = NT ? load_nt(srow + 4 * tid + 1024 * i) : *reinterpret_cast<const float4*>(srow + 4 * tid + 1024 * i);
			}
		}
		// Keys.  NONNEG (the engine: magnitudes): the raw bits order like the values.  Otherwise the row is staged as raw
		// bits all the same while every wave ORs the sign bits of what it loaded; only a row that holds a negative sample
		// (any set sign bit: -0 and negative NaNs included) is re-keyed in place below.  A caller's magnitude matrix thus
		// runs at the engine's speed without any promise, signed data stays exact (two VALU operations per staged value
		// and per result was the whole difference between the two builds: 0.63 against 0.74 of the HBM roof).
		int* wr = unit_ptr(img, tid + 6);
		if constexpr (!NONNEG) {
			int o = 0;
#pragma unroll
			for (int i = 0; i < 4; ++i)
				o |= __float_as_int(x[i].x) | __float_as_int(x[i].y) | __float_as_int(x[i].z) | __float_as_int(x[i].w);
			const bool wneg = __builtin_amdgcn_ballot_w64(o < 0) != 0ull;
			if (lane == 0)
				negf[wave] = wneg ? 1 : 0;
		}
#pragma unroll
		for (int i = 0; i < 4; ++i)
			*reinterpret_cast<int4*>(wr + i * 64 * RSTR) = make_int4(to_key<true>(x[i].x), to_key<true>(x[i].y),
			                                                         to_key<true>(x[i].z), to_key<true>(x[i].w));
		// replicate border (ippBorderRepl): words 0..23 = x[0], words 4120..4143 = x[4095]
		if (wave == 0) {
			const int b = __builtin_amdgcn_readfirstlane(to_key<true>(x[0].x));
			if (lane < 6)
				*reinterpret_cast<int4*>(unit_ptr(img, lane)) = make_int4(b, b, b, b);
		}
		if (wave == 3) {
			const int b = __builtin_amdgcn_readlane(to_key<true>(x[3].w), 63);
			if (lane < 6)
				*reinterpret_cast<int4*>(unit_ptr(img, 1030 + lane)) = make_int4(b, b, b, b);
		}
	}
	__syncthreads();
	bool signed_row = false; // workgroup-uniform
	// every thread re-keys the words it staged (and the border copies): the monotone key of median_net.h f2key
	auto rekey_image = [&]() {
		auto rekey = [](int* q) {
			int4 v = *reinterpret_cast<int4*>(q);
			v.x ^= (v.x >> 31) & 0x7fffffff;
			v.y ^= (v.y >> 31) & 0x7fffffff;
			v.z ^= (v.z >> 31) & 0x7fffffff;
			v.w ^= (v.w >> 31) & 0x7fffffff;
			*reinterpret_cast<int4*>(q) = v;
		};
		int* wr = unit_ptr(img, tid + 6);
#pragma unroll
		for (int i = 0; i < 4; ++i)
			rekey(wr + i * 64 * RSTR);
		if (wave == 0 && lane < 6)
			rekey(unit_ptr(img, lane));
		if (wave == 3 && lane < 6)
			rekey(unit_ptr(img, 1030 + lane));
		__syncthreads();
	};
	if constexpr (!NONNEG && (HERM || VARIANT == 2)) { // (the whole-row build reads the flags together with its own block, below)
		const int4 f = *reinterpret_cast<const int4*>(negf);
		signed_row = __builtin_amdgcn_readfirstlane(f.x | f.y | f.z | f.w) != 0;
		if (signed_row)
			rekey_image();
	}

	int out[16];
	int out_blk = tid;     // the block whose 16 outputs this thread holds
	bool out_valid = true;
	if constexpr (VARIANT == 2) {
		znet::lds_load<16>(&img[(tid + 1) * RSTR + 8], out);
		znet::lds_load<8>(&img[(tid + 2) * RSTR], *reinterpret_cast<int(*)[8]>(&out[8]));
	}
	else {
		// ---- own block sorted, wave edges published; then the neighbours' pieces and the selection tree.
		// Hermitian rows: P[4096-k] == P[k] for 23 < k < 2048, so waves 0 and 1 take blocks 0..127, wave 2 the
		// leftovers (lanes 0..31 blocks 128..159: bin 2048, and the pieces wave 1's last lanes need; lanes 32..63
		// blocks 224..255: the last 23 bins) and wave 3 sits the stage out -- as in rt_fused.hip.
		zm47::Pieces pc;
		if constexpr (HERM) {
			const int blk = wave < 2 ? tid : (lane < 32 ? 128 + lane : 192 + lane);
			if (wave < 3)
				zm47::m47_sort_and_publish(img, edge, blk, lane, wave, tid == 0, blk == 255, pc);
			__syncthreads();
			if (wave < 3)
				zm47::m47_select(img, edge, blk, wave, pc, out);
			out_blk = blk;
			out_valid = wave < 3;
		}
		else {
			// (the same steps as median47_core.h, written out: as one scope the compiler keeps this kernel at 72
			// registers and seven workgroups per CU, through the shared functions it needs 92)
		// ---- own block B(tid) = chunk tid + 1: sorted, with its sorted halves and quarters
		int s16[16], oct[16], quad[16];
		{
			int raw[16];
			znet::lds_load<16>(&img[(tid + 1) * RSTR], raw);
			if constexpr (!NONNEG) { // the sign flags of the four waves come back with the block: no LDS round trip of their own
				const int4 f = *reinterpret_cast<const int4*>(negf);
				signed_row = __builtin_amdgcn_readfirstlane(f.x | f.y | f.z | f.w) != 0;
				if (signed_row) { // the block in registers was staged (as raw bits) by other threads: same map, then the image
#pragma unroll
					for (int i = 0; i < 16; ++i)
						raw[i] ^= (raw[i] >> 31) & 0x7fffffff;
					__syncthreads(); // (every thread has read its block before anyone rewrites the image)
					rekey_image();
				}
			}
			znet::pyramid16(raw, s16, oct, quad);
		}
		// pieces wanted by the lanes to the left (they sit two and one blocks below this one) ...
		int up[16], dn[16];
	#pragma unroll
		for (int i = 0; i < 8; ++i) {
			up[i] = oct[i];     // lower half
			dn[i] = oct[8 + i]; // upper half
		}
	#pragma unroll
		for (int i = 0; i < 4; ++i) {
			up[8 + i] = quad[i];      // 1st quarter
			up[12 + i] = quad[8 + i]; // 3rd quarter
			dn[8 + i] = quad[4 + i];  // ... and by the lane to the right: 2nd and 4th quarter
			dn[12 + i] = quad[12 + i];
		}
		// ---- wave edges through LDS
		if (wave > 0 && lane < 2) {
			int* e = &edge[wave - 1][32 + 16 * lane];
			znet::lds_store<16>(e, up);
			if (lane == 0)
				znet::lds_store<16>(&edge[wave - 1][16], s16);
		}
		if (wave < 3 && lane == 63)
			znet::lds_store<16>(&edge[wave + 1][0], dn);
		if (tid == 0) { // B(-1) = 16 copies of x[0]
			const int b = img[0];
			const int4 q = make_int4(b, b, b, b);
	#pragma unroll
			for (int i = 0; i < 4; ++i)
				*reinterpret_cast<int4*>(&edge[0][4 * i]) = q;
		}
		if (tid == 255) { // B(256) = x[4088..4095] and eight copies of c = x[4095]; B(257) = 16 copies of c
			int w[8];
			znet::lds_load<8>(&img[257 * RSTR], w);
			const int c = w[7];
			znet::oe_merge<2, 0>(w);
			znet::oe_merge<2, 2>(w);
			znet::oe_merge<2, 4>(w);
			znet::oe_merge<2, 6>(w);
			znet::oe_merge<4, 0>(w);
			znet::oe_merge<4, 4>(w);
			int pc[16], s[16];
	#pragma unroll
			for (int i = 0; i < 4; ++i) {
				pc[8 + i] = w[i]; // 1st quarter sorted
				pc[12 + i] = c;   // 3rd quarter
			}
			znet::oe_merge<8, 0>(w);
	#pragma unroll
			for (int i = 0; i < 8; ++i) {
				pc[i] = w[i];            // lower half sorted
				s[i] = min(w[i], c);     // the block sorted: the eight samples with eight copies of c spliced in
				s[8 + i] = max(w[i], c);
			}
			znet::lds_store<16>(&edge[3][16], s);
			znet::lds_store<16>(&edge[3][32], pc);
			const int4 q = make_int4(c, c, c, c);
	#pragma unroll
			for (int i = 0; i < 4; ++i)
				*reinterpret_cast<int4*>(&edge[3][48 + 4 * i]) = q;
		}
		__syncthreads();

		// ---- neighbours' pieces: DPP shifts, the wave's edge record as `old`
		znet::Shared47 sh;
		{
			const int* ed = edge[wave];
			int eb[16], B[16];
			znet::lds_load<16>(ed + 16, eb);
	#pragma unroll
			for (int i = 0; i < 16; ++i)
				B[i] = zm47::dpp_from_next(eb[i], s16[i]);
			znet::mid16_of_two_sorted16(s16, B, sh.cand);
			int e0[16], e1[16], el[16], hi[16], lo[16];
			znet::lds_load<16>(ed + 32, e0);
			znet::lds_load<16>(ed + 48, e1);
			znet::lds_load<16>(ed, el);
	#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int x1 = zm47::dpp_from_next(e0[i], up[i]); // pieces of B(t+1)
				hi[i] = zm47::dpp_from_next(e1[i], x1);           // pieces of B(t+2)
				lo[i] = zm47::dpp_from_prev(el[i], dn[i]);        // pieces of B(t-1)
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
		znet::lds_load<16>(&img[tid * RSTR], sh.lo_raw);       // B(t-1) as it stands
		znet::lds_load<16>(&img[(tid + 3) * RSTR], sh.hi_raw); // B(t+2)

		znet::medians47_shared(sh, out);

		}
	}

	// a signed row's results go back through the key map: one workgroup-uniform branch per group of values (written as a
	// block so that the compiler keeps it a branch; as a select per value it costs every row three operations per result)
	auto unkey4 = [signed_row](int4& k) {
		if (signed_row) {
			k.x ^= (k.x >> 31) & 0x7fffffff;
			k.y ^= (k.y >> 31) & 0x7fffffff;
			k.z ^= (k.z >> 31) & 0x7fffffff;
			k.w ^= (k.w >> 31) & 0x7fffffff;
		}
	};
	if constexpr (DIRECT) {
#pragma unroll
		for (int v = 0; v < 4; ++v)
		{
			int4 k = make_int4(out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]);
			unkey4(k);
			ZH_CHK(drow + 16 * tid + 4 * v, 4);
			*reinterpret_cast<float4*>(drow + 16 * tid + 4 * v) =
			    make_float4(__int_as_float(k.x), __int_as_float(k.y), __int_as_float(k.z), __int_as_float(k.w));
		}
	}
	else {
		__syncthreads(); // every flank is in registers: the image can take the results
		if (!HERM || out_valid)
			znet::lds_store<16>(&img[(HERM ? out_blk : tid) * RSTR], out);
		__syncthreads();
		const int* rd = unit_ptr(img, tid);
		if (VARIANT == 3 && p.ring > 0) // always true: the host never passes ring <= 0
			return;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int c = 4 * tid + 1024 * i;
			if (HERM && c > 2048 && c < 4064) // Hermitian rows: nobody reads these columns
				continue;
			int4 k = *reinterpret_cast<const int4*>(rd + i * 64 * RSTR);
			unkey4(k);
			const float4 r = make_float4(__int_as_float(k.x), __int_as_float(k.y), __int_as_float(k.z), __int_as_float(k.w));
			ZH_CHK(drow + 4 * tid + 1024 * i, 4);
			if (NT)
				store_nt(drow + 4 * tid + 1024 * i, r);
			else
				*reinterpret_cast<float4*>(drow + 4 * tid + 1024 * i) = r;
		}
	}
}

} // namespace

// 47 taps, frequency direction, whole 4096-bin rows, 16-byte aligned.  *handled = false: generic kernel.
int launch_median47_dpp(const FilterArgs& a, hipStream_t stream, bool* handled)
{
	*handled = false;
	if (a.len != 47 || a.direction != ZEN_HIP_FREQUENCY || a.cols != COLS || g_opt_no_median47_dpp)
		return ZEN_HIP_OK;
	const bool vec_ok = ((reinterpret_cast<uintptr_t>(a.src) & 15) == 0) && ((reinterpret_cast<uintptr_t>(a.dst) & 15) == 0)
	                    && (a.src_stream_stride % 4 == 0) && (a.dst_stream_stride % 4 == 0);
	if (!vec_ok || a.ring_rows <= 0 || a.ring_rows > 0x3fffffff || a.n_out_rows > a.ring_rows)
		return ZEN_HIP_OK;
	// the row clamp of the time direction never acts on a frequency filter's own rows; if a caller asks for
	// rows outside [clamp_lo, clamp_hi] anyway, the generic kernel (which clamps) takes the call
	if (a.first_row < a.clamp_lo || a.first_row + a.n_out_rows - 1 > a.clamp_hi)
		return ZEN_HIP_OK;
	*handled = true;
	M47Args p;
	p.src = a.src;
	p.dst = a.dst;
	p.src_stream_stride = a.src_stream_stride;
	p.dst_stream_stride = a.dst_stream_stride;
	p.herm = a.hermitian;
	p.ring = (int)a.ring_rows;
	p.row_base = (int)(((a.first_row % a.ring_rows) + a.ring_rows) % a.ring_rows);
	dim3 grid((unsigned)a.n_out_rows, (unsigned)a.n_streams);
	const int variant = a.hermitian ? 0 : (int)g_opt_median47_variant; // the diagnostic builds know whole rows only
#define ZH_M47(NN, V) hipLaunchKernelGGL((median47_dpp_kernel<NN, V>), grid, dim3(256), 0, stream, p)
	if (a.hermitian) {
		if (a.nonneg)
			hipLaunchKernelGGL((median47_dpp_kernel<true, 0, true>), grid, dim3(256), 0, stream, p);
		else
			hipLaunchKernelGGL((median47_dpp_kernel<false, 0, true>), grid, dim3(256), 0, stream, p);
	}
	else if (a.nonneg) {
		switch (variant) {
		case 1: ZH_M47(true, 1); break;
#ifdef ZEN_HIP_DIAG
		case 2: ZH_M47(true, 2); break;
		case 3: ZH_M47(true, 3); break;
		case 4: ZH_M47(true, 4); break;
#endif
		default: ZH_M47(true, 0); break;
		}
	}
	else {
		switch (variant) {
		case 1: ZH_M47(false, 1); break;
#ifdef ZEN_HIP_DIAG
		case 2: ZH_M47(false, 2); break;
		case 3: ZH_M47(false, 3); break;
		case 4: ZH_M47(false, 4); break;
#endif
		default: ZH_M47(false, 0); break;
		}
	}
#undef ZH_M47
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace zen_hip_impl
