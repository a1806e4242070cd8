// stft.hip -- analysis kernel, overlap-add and the plain transform of the HPSS engine for gfx950 (fft_dev.h).
//
//   stft_kernel   : hps.cu:452-472 + :492 (new row only): frame assembly from the hop stream, sqrt-Hann
//                   window, zero-padding (pruned: the upper half is never loaded), forward FFT, spectrum
//                   and magnitude rows written into the ring.  Replaces thrust::copy x2, transform, fill,
//                   cufftExecC2C, the STFT rotation (2 overlapping copies) and the whole-matrix abs.
//   (the synthesis kernels are in istft.hip)
//   finalize_kernel: overlap-add of neighbouring frames + copy-out (hps.cu:435-449, :526-528, :341-363).
//   fft_kernel    : the plain FFTC2CWrapperGPU transform (fftw.h:35-43).
#include "common.h"
#include "bounds.h"
#include "fft_dev.h"
#include "fft_launch.h"
#include "rfft_dev.h"
#include "masks.h"
#include "stft.h"

#include <cfloat>
#include <type_traits>

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

using zfft::Plan;

// ------------------------------------------------------------------------------------------------
template <int LOG2N>
struct StftIn {
	const float* prev; // hop samples before the frame's second half
	const float* cur;  // hop samples
	const float* window;
	int hop;
	int nv_prev, nv_cur; // samples of `prev` / `cur` that exist (StftArgs::in_valid); the rest of the hop reads as zero
	__device__ __forceinline__ float2 operator()(int idx, int) const
	{
		// (one unconditional load from a selected address -- a sample that does not exist reads the window table instead
		// and is replaced by zero: `ok ? p[k] : 0` made every load a branch with a full wait inside)
		const bool first = idx < hop;
		const int k = first ? idx : idx - hop;
		const bool ok = k < (first ? nv_prev : nv_cur);
		const float* p = ok ? (first ? prev : cur) + k : window;
		ZH_CHK(p, 1);
		ZH_CHK(window + idx, 1);
		const float v = *p;
		const float x = ok ? v : 0.0f;
		return make_float2(x * window[idx], 0.0f); // window_functor hps.h:24-33
	}
};

// samples of hop i of a row that exist when the row holds in_valid samples
__device__ __forceinline__ int valid_in_hop(long long in_valid, int i, int hop)
{
	const long long v = in_valid - (long long)i * hop;
	return v <= 0 ? 0 : (v >= hop ? hop : (int)v);
}

struct StftOut {
	float2* S;  // n/2 + 1 bins are kept: the frame is real, so the spectrum is exactly Hermitian
	            // (radix-2 DAG + a twiddle table with tw[n/2-j] == -conj(tw[j]) bit for bit) and the
	            // synthesis kernel rebuilds S[n-k] = conj(S[k]).  Halves the spectrum traffic.
	float* mag; // bins 0..n/2, and their mirror images n/2+1..n-1 where somebody reads whole rows (`full`):
	            // |S[n-k]| == |S[k]| bit for bit, so the double-precision hypot is taken once per pair
	int n;
	bool full;
	__device__ __forceinline__ void operator()(int idx, float2 X, bool lower, int) const
	{
		if (lower || idx == (n >> 1)) {
			ZH_CHK(S + idx, 1);
			ZH_CHK(mag + idx, 1);
			S[idx] = X;
			const float m = zfft::cabs_exact(X.x, X.y); // complex_abs_functor hps.h:82-89
			mag[idx] = m;
			if (full && idx != 0 && idx != (n >> 1)) {
				ZH_CHK(mag + (n - idx), 1);
				mag[n - idx] = m;
			}
		}
	}
};

template <int LOG2N>
__global__ __launch_bounds__(Plan<LOG2N>::THREADS) void stft_kernel(StftArgs a)
{
	using PL = Plan<LOG2N>;
	extern __shared__ float2 lds[];
	const int tid = threadIdx.x, s = blockIdx.y, hop = a.hop;
	if (blockIdx.x == gridDim.x - 1) { // housekeeping block
		const float* last = a.in + (long long)s * a.in_stride + (long long)(a.n_frames - 1) * hop;
		const int nv = valid_in_hop(a.in_valid, a.n_frames - 1, hop);
		for (int i = tid; i < hop; i += PL::THREADS) {
			ZH_CHK(a.tail_next + ((long long)s * hop + i), 1);
			if (i < nv)
				ZH_CHK(last + i, 1);
			a.tail_next[(long long)s * hop + i] = i < nv ? last[i] : 0.0f;
		}
		if (a.prev_frames > 0) {
			for (int o = 0; o < 3; ++o) {
				if (!a.carry[o])
					continue;
				const float* y = a.Y[o] + (long long)s * a.y_stream_stride
				                 + (long long)(a.prev_frames - 1) * (2 * hop) + hop;
				for (int i = tid; i < hop; i += PL::THREADS) {
					ZH_CHK(a.carry[o] + ((long long)s * hop + i), 1);
					ZH_CHK(y + i, 1);
					a.carry[o][(long long)s * hop + i] = y[i];
				}
			}
		}
		return;
	}
	const int slot = tid / PL::TF, tf = tid - slot * PL::TF;
	const int f_ = blockIdx.x * PL::FRAMES_PER_BLOCK + slot;
	const bool active = f_ < a.n_frames;
	const int f = active ? f_ : a.n_frames - 1; // (an inactive slot transforms the last frame again and stores nothing)
	const float* in_s = a.in + (long long)s * a.in_stride;
	StftIn<LOG2N> in;
	in.prev = (f == 0) ? a.tail_prev + (long long)s * hop : in_s + (long long)(f - 1) * hop;
	in.cur = in_s + (long long)f * hop;
	in.window = a.window;
	in.hop = hop;
	in.nv_prev = f == 0 ? hop : valid_in_hop(a.in_valid, f - 1, hop);
	in.nv_cur = valid_in_hop(a.in_valid, f, hop);
	const long long row = ((a.row0 + f) % a.ring_rows) + (long long)s * a.ring_rows;
	StftOut out;
	out.S = a.S + row * a.s_stride;
	out.mag = a.mag + row * PL::N;
	out.n = PL::N;
	out.full = f >= a.mag_full_from;
	if constexpr (LOG2N == 10) { // (every pass's twiddles requested at its top: 0.65 -> 0.61 ms per 331 456 frames at nfft 1024; the
		const zfft::TwGlobalPre twp{a.tw}; // other sizes lose with it -- nfft 16384: 0.94 -> 1.02 ms -- and keep the scheduler's placement)
		zfft::fft_frame<LOG2N, false, true, false>(tf, lds + slot * PL::LDS_FLOAT2, twp, in, out, active);
	}
	else {
		zfft::fft_frame<LOG2N, false, true, false>(tf, lds + slot * PL::LDS_FLOAT2, a.tw, in, out, active);
	}
}

// ------------------------------------------------------------------------------------------------
// The same analysis through the real-input transform of rfft_dev.h: the frame is real (window_functor, hps.h:24-33; the
// imaginary parts are zeroed, hps.cu:456-465), so every sub-transform of the radix-2 DAG is exactly Hermitian and only
// its lower half is computed -- half the butterflies, half the LDS image, N/32 threads per frame, the oracle's bits.
template <int LOG2N>
struct StftInR {
	static constexpr bool HAS_FIRST_PASS = true;
	static constexpr int HOP = (1 << LOG2N) / 4; // hps.h:224-225: nwin = 2 hop, nfft = 4 hop (launch_stft checks)
	const float* prev;
	const float* cur;
	const float* window;
	int nv_prev, nv_cur;
	bool contiguous; // wave-uniform: every frame of this wavefront has all its samples, prev hop and hop next to each other
	__device__ __forceinline__ float operator()(int idx) const
	{
		const bool first = idx < HOP;
		const int k = first ? idx : idx - HOP;
		const bool ok = k < (first ? nv_prev : nv_cur);
		const float* p = ok ? (first ? prev : cur) + k : window; // (one unconditional load from a selected address: StftIn)
		ZH_CHK(p, 1);
		ZH_CHK(window + idx, 1);
		const float v = *p;
		const float x = ok ? v : 0.0f;
		return x * window[idx];
	}
	// All of a thread's loads in one of two forms, chosen by ONE wave-uniform branch.  The usual frame -- not a chunk's first
	// (whose previous hop is the saved tail), every sample there -- is 2 * hop consecutive samples: plain loads from one base.
	// The general form selects an address and a value per sample (five VALU instructions each).
	template <int NI, int R, int J, int TF, bool ZU>
	__device__ __forceinline__ void first_pass(int tf, float (&x)[NI][R]) const
	{
		static_assert(ZU, "the analysis frame is zero-padded");
		if (contiguous) {
			const float* base = cur - HOP;
			float v[NI][R / 2], w[NI][R / 2];
#pragma unroll
			for (int i = 0; i < NI; ++i)
#pragma unroll
				for (int m = 0; m < R / 2; ++m) {
					const int idx = tf + i * TF + m * J;
					ZH_CHK(base + idx, 1);
					ZH_CHK(window + idx, 1);
					v[i][m] = base[idx];
					w[i][m] = window[idx];
				}
#pragma unroll
			for (int i = 0; i < NI; ++i)
#pragma unroll
				for (int m = 0; m < R; ++m)
					x[i][m] = m < R / 2 ? v[i][m < R / 2 ? m : 0] * w[i][m < R / 2 ? m : 0] : 0.0f; // window_functor hps.h:24-33
		}
		else {
#pragma unroll
			for (int i = 0; i < NI; ++i)
#pragma unroll
				for (int m = 0; m < R; ++m)
					x[i][m] = m < R / 2 ? (*this)(tf + i * TF + m * J) : 0.0f;
		}
	}
};
// FULL rows (StftArgs::mag_full_from: the last W - 1 frames of a chunk) also get the mirrored upper half of the magnitudes;
// `any_full` is wave-uniform, so the usual frame pays a scalar branch per bin for it and nothing else.
struct StftOutR {
	float2* S;
	float* mag;
	int n;
	bool full, any_full;
	// bin = base + off, off a compile-time constant: S + base and mag + base are formed once per value of base (two per item),
	// the stores take their offsets as immediates
	__device__ __forceinline__ void operator()(int base, int off, float2 X, int slot) const
	{
		float2* Sb = S + base;
		float* mb = mag + base;
		ZH_CHK(Sb + off, 1);
		ZH_CHK(mb + off, 1);
		Sb[off] = X;
		// complex_abs_functor hps.h:82-89; bin nfft/2 (slot 16) is real: the root of an exact square
		const float m = slot == 16 ? __builtin_fabsf(X.x) : zfft::cabs_exact(X.x, X.y);
		mb[off] = m;
		if (any_full) {
			const int bin = base + off;
			if (full && bin != 0 && bin != (n >> 1)) {
				float* mm = mag + (n - base);
				ZH_CHK(mm - off, 1);
				mm[-off] = m;
			}
		}
	}
};

// The same results by way of the frame's LDS image, so that they leave in 16-byte stores of consecutive bins.  The last pass
// hands thread k bins k + c 2^s and d 2^s - k: stored from there, a wave instruction writes 8 bytes per lane for the spectrum
// and 4 for the magnitudes -- and that store pattern ALONE, in a kernel with no arithmetic and no loads, takes as long as the
// whole analysis kernel (tools/ubench_rowwrite.hip, nfft 1024: 0.56 ms per 331 456 frames; the same bytes as 16-byte stores
// 0.41).  The image is free once the last pass has its inputs in registers: the spectrum goes there in natural order and
// is copied out four floats per lane; then the magnitudes (kept in registers meanwhile) take the same road.
template <int LOG2N>
struct StftOutLds {
	float2* nat; // the frame's image, bin b at nat[b]
	float m[17];
	int bin[17];
	__device__ __forceinline__ void operator()(int base, int off, float2 X, int slot)
	{
		nat[base + off] = X;
		m[slot] = slot == 16 ? __builtin_fabsf(X.x) : zfft::cabs_exact(X.x, X.y); // complex_abs_functor hps.h:82-89 (StftOutR)
		bin[slot] = base + off;
	}
};
template <int LOG2N>
__device__ __forceinline__ void stft_store_rows(int tf, StftOutLds<LOG2N>& o, float2* __restrict__ S, float* __restrict__ mag, bool full,
                                                bool any_full)
{
	using RP = zfft::RPlan<LOG2N>;
	constexpr int N = RP::N, TF = RP::TF;
	static_assert((N / 2 + 1) * 8 <= RP::LDS_FLOAT2 * 8, "the spectrum's half fits the image");
	zfft::frame_sync<TF>();
	{
		const float4* n4 = reinterpret_cast<const float4*>(o.nat);
		float4* g4 = reinterpret_cast<float4*>(__builtin_assume_aligned(S, 16));
		float4 v[N / 4 / TF];
#pragma unroll
		for (int i = 0; i < N / 4 / TF; ++i)
			v[i] = n4[tf + i * TF];
#pragma unroll
		for (int i = 0; i < N / 4 / TF; ++i) {
			ZH_CHK(g4 + tf + i * TF, 1);
			g4[tf + i * TF] = v[i];
		}
		if (tf == 0) {
			ZH_CHK(S + N / 2, 1);
			S[N / 2] = o.nat[N / 2];
		}
	}
	zfft::frame_sync<TF>(); // the spectrum has been read: the magnitudes take its place
	float* natm = reinterpret_cast<float*>(o.nat);
#pragma unroll
	for (int sl = 0; sl < 16; ++sl)
		natm[o.bin[sl]] = o.m[sl];
	if (o.bin[16] >= 0)
		natm[o.bin[16]] = o.m[16];
	zfft::frame_sync<TF>();
	{
		const float4* n4 = reinterpret_cast<const float4*>(natm);
		float4* g4 = reinterpret_cast<float4*>(__builtin_assume_aligned(mag, 16));
		float4 v[N / 8 / TF];
#pragma unroll
		for (int i = 0; i < N / 8 / TF; ++i)
			v[i] = n4[tf + i * TF];
#pragma unroll
		for (int i = 0; i < N / 8 / TF; ++i) {
			ZH_CHK(g4 + tf + i * TF, 1);
			g4[tf + i * TF] = v[i];
		}
		if (tf == 0) {
			ZH_CHK(mag + N / 2, 1);
			mag[N / 2] = natm[N / 2];
		}
	}
	if (any_full) { // (wave-uniform; the last W - 1 frames of a chunk) the mirrored upper half: |S[n-k]| == |S[k]| bit for bit
		if (full) {
			for (int b = 1 + tf; b < N / 2; b += TF) {
				ZH_CHK(mag + (N - b), 1);
				mag[N - b] = natm[b];
			}
		}
	}
}
#ifndef ZEN_RFFT_WIDE_MASK
#define ZEN_RFFT_WIDE_MASK 0x0FC0 // nfft 64 .. 2048: a frame inside one wavefront
#endif

// (four waves per SIMD: two 512-thread workgroups per CU at nfft 16384, whose 74 KB images now both fit the LDS)
template <int LOG2N>
#ifndef ZEN_RFFT_WAVES
#define ZEN_RFFT_WAVES 4
#endif
__global__ __launch_bounds__(zfft::RPlan<LOG2N>::THREADS) __attribute__((amdgpu_waves_per_eu(ZEN_RFFT_WAVES, ZEN_RFFT_WAVES))) void stft_real_kernel(StftArgs a)
{
	using RP = zfft::RPlan<LOG2N>;
	extern __shared__ float2 lds[];
	const int tid = threadIdx.x, s = blockIdx.y, hop = a.hop;
	if (blockIdx.x == gridDim.x - 1) { // housekeeping block
		const float* last = a.in + (long long)s * a.in_stride + (long long)(a.n_frames - 1) * hop;
		const int nv = valid_in_hop(a.in_valid, a.n_frames - 1, hop);
		for (int i = tid; i < hop; i += RP::THREADS) {
			ZH_CHK(a.tail_next + ((long long)s * hop + i), 1);
			if (i < nv)
				ZH_CHK(last + i, 1);
			a.tail_next[(long long)s * hop + i] = i < nv ? last[i] : 0.0f;
		}
		if (a.prev_frames > 0) {
			for (int o = 0; o < 3; ++o) {
				if (!a.carry[o])
					continue;
				const float* y = a.Y[o] + (long long)s * a.y_stream_stride + (long long)(a.prev_frames - 1) * (2 * hop) + hop;
				for (int i = tid; i < hop; i += RP::THREADS) {
					ZH_CHK(a.carry[o] + ((long long)s * hop + i), 1);
					ZH_CHK(y + i, 1);
					a.carry[o][(long long)s * hop + i] = y[i];
				}
			}
		}
		return;
	}
	const int slot = tid / RP::TF, tf = tid - slot * RP::TF;
	__builtin_assume(tf >= 0 && tf < RP::TF);
	const int f_ = blockIdx.x * RP::FRAMES_PER_BLOCK + slot;
	// A slot past the last frame transforms the last frame again AND stores it again: the same values to the same places as the
	// slot that owns that frame (a benign duplicate), instead of a test around every store of every thread.
	const int f = f_ < a.n_frames ? f_ : a.n_frames - 1;
	const float* in_s = a.in + (long long)s * a.in_stride;
	StftInR<LOG2N> in;
	in.prev = (f == 0) ? a.tail_prev + (long long)s * hop : in_s + (long long)(f - 1) * hop;
	in.cur = in_s + (long long)f * hop;
	in.window = a.window;
	in.nv_prev = f == 0 ? hop : valid_in_hop(a.in_valid, f - 1, hop);
	in.nv_cur = valid_in_hop(a.in_valid, f, hop);
	in.contiguous = __builtin_amdgcn_ballot_w64(!(f > 0 && in.nv_prev == hop && in.nv_cur == hop)) == 0ull;
	// (launch_stft_t hands over row0 already reduced modulo ring_rows, and n_frames <= ring_rows: one conditional subtraction
	// instead of a 64-bit remainder per thread -- fifty VALU instructions)
	long long rr = a.row0 + f;
	rr = rr >= a.ring_rows ? rr - a.ring_rows : rr;
	const long long row = rr + (long long)s * a.ring_rows;
	const bool full = f >= a.mag_full_from;
	const bool any_full = __builtin_amdgcn_ballot_w64(full) != 0ull;
	float2* S_row = a.S + row * a.s_stride;
	float* mag_row = a.mag + row * RP::N;
#ifndef ZEN_RFFT_TWPRE_MASK
#define ZEN_RFFT_TWPRE_MASK 0xFFFF
#endif
	constexpr bool TWPRE = ((ZEN_RFFT_TWPRE_MASK) >> LOG2N) & 1; // (the passes' twiddles requested at the top of each pass: -4 % at
	using TWT = std::conditional_t<TWPRE, zfft::TwGlobalPre, zfft::TwGlobal>; // nfft 1024 and 16384, A/B per size)
	const TWT tw{a.tw};
	if constexpr ((((ZEN_RFFT_WIDE_MASK) >> LOG2N) & 1) != 0) {
		if (a.s_stride % 2 == 0) { // (16-byte aligned rows: the engine's are; workgroup-uniform)
			StftOutLds<LOG2N> out;
			out.nat = lds + slot * RP::LDS_FLOAT2;
			out.bin[16] = -1;
			out.m[16] = 0.0f;
			zfft::rfft_frame<LOG2N, true>(tf, lds + slot * RP::LDS_FLOAT2, tw, in, out, true);
			stft_store_rows<LOG2N>(tf, out, S_row, mag_row, full, any_full);
			return;
		}
	}
	StftOutR out;
	out.S = S_row;
	out.mag = mag_row;
	out.n = RP::N;
	out.full = full;
	out.any_full = any_full;
	zfft::rfft_frame<LOG2N, true>(tf, lds + slot * RP::LDS_FLOAT2, tw, in, out, true);
}

template <int LOG2N>
constexpr size_t rlds_bytes()
{
	return sizeof(float2) * (size_t)zfft::RPlan<LOG2N>::LDS_FLOAT2 * zfft::RPlan<LOG2N>::FRAMES_PER_BLOCK;
}

// streaming 16-byte accesses (aligned): the Y rows are read once, the outputs written once
__device__ __forceinline__ float4 load4_nt(const float* p)
{
	const float* q = reinterpret_cast<const float*>(__builtin_assume_aligned(p, 16));
	ZH_CHK(q, 4);
	return make_float4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2),
	                   __builtin_nontemporal_load(q + 3));
}
__device__ __forceinline__ void store4_nt(float* p, float4 v)
{
	float* q = reinterpret_cast<float*>(__builtin_assume_aligned(p, 16));
	ZH_CHK(q, 4);
	__builtin_nontemporal_store(v.x, q);
	__builtin_nontemporal_store(v.y, q + 1);
	__builtin_nontemporal_store(v.z, q + 2);
	__builtin_nontemporal_store(v.w, q + 3);
}

// ------------------------------------------------------------------------------------------------
// VEC: four consecutive samples of a hop per thread, 16-byte accesses (hop is a multiple of 4 and the engine's rows
// are aligned; the caller's `out` rows must be too: launch_finalize checks)
template <bool VEC>
__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs a)
{
	const int s = blockIdx.y, hop = a.hop;
	const float* Y = a.Y + (long long)s * a.y_stream_stride;
	const float* carry = a.carry + (long long)s * hop;
	float* out = a.out + (long long)s * a.out_stride;
	constexpr int V = VEC ? 4 : 1;
	const long long n = (long long)a.n_frames * hop / V;
	const int per_hop = hop / V;
	for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
	     e += (long long)gridDim.x * blockDim.x) {
		const long long i = e / per_hop;
		const int k = (int)(e - i * per_hop) * V;
		const float* prev = (i == 0) ? carry + k : Y + (i - 1) * 2 * hop + hop + k;
		const float* cur = Y + i * 2 * hop + k;
		if constexpr (VEC) {
			const float4 p = load4_nt(prev), c = load4_nt(cur);
			store4_nt(out + i * hop + k, make_float4(p.x + c.x, p.y + c.y, p.z + c.z, p.w + c.w));
		}
		else {
			ZH_CHK(prev, 1);
			ZH_CHK(cur, 1);
			ZH_CHK(out + (i * hop + k), 1);
			out[i * hop + k] = prev[0] + cur[0];
		}
	}
}

// The overlap-add with the destination arithmetic of the offline driver folded in (FinalizeArgs, second part): one
// pass over the Y rows instead of overlap-add, sum, shift and truncate as four (hpri.hip).
template <bool VEC>
__global__ __launch_bounds__(256) void finalize_spec_kernel(FinalizeArgs a)
{
	const int s = blockIdx.y, hop = a.hop;
	const float* Y = a.Y + (long long)s * a.y_stream_stride;
	const float* carry = a.carry + (long long)s * hop;
	const float* Y2 = a.Y2 ? a.Y2 + (long long)s * a.y_stream_stride : nullptr;
	const float* carry2 = a.Y2 ? a.carry2 + (long long)s * hop : nullptr;
	float* out = a.out + (long long)s * a.out_stride;
	constexpr int V = VEC ? 4 : 1;
	const long long n = (long long)a.n_frames * hop / V;
	const int per_hop = hop / V;
	for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
		const long long i = e / per_hop;
		const int k = (int)(e - i * per_hop) * V;
		float v[V];
		{
			const float* prev = (i == 0) ? carry + k : Y + (i - 1) * 2 * hop + hop + k;
			const float* cur = Y + i * 2 * hop + k;
			if constexpr (VEC) {
				const float4 p = load4_nt(prev), c = load4_nt(cur);
				v[0] = p.x + c.x, v[1] = p.y + c.y, v[2] = p.z + c.z, v[3] = p.w + c.w;
			}
			else {
				ZH_CHK(prev, 1);
				ZH_CHK(cur, 1);
				v[0] = prev[0] + cur[0];
			}
		}
		if (Y2) { // sum_vectors_functor hps.h:142-150 on the two finished hops
			const float* prev = (i == 0) ? carry2 + k : Y2 + (i - 1) * 2 * hop + hop + k;
			const float* cur = Y2 + i * 2 * hop + k;
			if constexpr (VEC) {
				const float4 p = load4_nt(prev), c = load4_nt(cur);
				v[0] = v[0] + (p.x + c.x), v[1] = v[1] + (p.y + c.y), v[2] = v[2] + (p.z + c.z), v[3] = v[3] + (p.w + c.w);
			}
			else {
				ZH_CHK(prev, 1);
				ZH_CHK(cur, 1);
				v[0] = v[0] + (prev[0] + cur[0]);
			}
		}
		else if (a.add_zero) { // the partner is not computed: the reference adds its all-zero accumulator
#pragma unroll
			for (int q = 0; q < V; ++q)
				v[q] = v[q] + 0.0f;
		}
		const long long p0 = a.pos0 + i * hop + k;
		const long long j = p0 - a.shift;
		if (VEC && j >= 0 && j + 4 <= a.len) {
			store4_nt(out + j, make_float4(v[0], v[1], v[2], v[3]));
		}
		else {
#pragma unroll
			for (int q = 0; q < V; ++q)
				if (j + q >= 0 && j + q < a.len) {
					ZH_CHK(out + (j + q), 1);
					out[j + q] = v[q];
				}
		}
		if (p0 + V > a.dup_from) {
#pragma unroll
			for (int q = 0; q < V; ++q) {
				const long long j2 = p0 + q - a.dup_shift;
				if (p0 + q >= a.dup_from && j2 >= 0 && j2 < a.dup_len) {
					ZH_CHK(out + j2, 1);
					out[j2] = v[q];
				}
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------
struct PlainIn {
	const float2* d;
	__device__ __forceinline__ float2 operator()(int idx, int) const
	{
		ZH_CHK(d + idx, 1);
		return d[idx];
	}
};
struct PlainOut {
	float2* d;
	__device__ __forceinline__ void operator()(int idx, float2 X, bool, int) const
	{
		ZH_CHK(d + idx, 1);
		d[idx] = X;
	}
};

template <int LOG2N, bool INV>
__global__ __launch_bounds__(Plan<LOG2N>::THREADS) void fft_kernel(float2* data, const float2* tw, int batch)
{
	using PL = Plan<LOG2N>;
	extern __shared__ float2 lds[];
	const int tid = threadIdx.x;
	const int slot = tid / PL::TF, tf = tid - slot * PL::TF;
	const long long f_ = (long long)blockIdx.x * PL::FRAMES_PER_BLOCK + slot;
	const bool active = f_ < batch;
	const long long f = active ? f_ : batch - 1;
	PlainIn in{data + f * PL::N};
	PlainOut out{data + f * PL::N};
	zfft::fft_frame<LOG2N, INV, false, false>(tf, lds + slot * PL::LDS_FLOAT2, tw, in, out, active);
}

template <int LOG2N>
int launch_stft_t(const StftArgs& a, hipStream_t stream)
{
	using PL = Plan<LOG2N>;
	if (!g_opt_no_rfft && a.hop * 4 == PL::N && a.n_frames <= a.ring_rows && a.row0 >= 0) { // (StftInR::HOP; the engine never asks for anything else, hps.h:224-225)
		using RP = zfft::RPlan<LOG2N>;
		auto kern = stft_real_kernel<LOG2N>;
		ZH_TRY(set_lds(kern, rlds_bytes<LOG2N>()));
		dim3 grid((unsigned)ceil_div((size_t)a.n_frames, (size_t)RP::FRAMES_PER_BLOCK) + 1, (unsigned)a.n_streams);
		StftArgs ar = a;
		ar.row0 = a.row0 % a.ring_rows; // (see the kernel)
		hipLaunchKernelGGL(kern, grid, dim3(RP::THREADS), rlds_bytes<LOG2N>(), stream, ar);
		ZH_HIP(hipGetLastError());
		return ZEN_HIP_OK;
	}
	auto kern = stft_kernel<LOG2N>;
	ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
	dim3 grid((unsigned)ceil_div((size_t)a.n_frames, (size_t)PL::FRAMES_PER_BLOCK) + 1, (unsigned)a.n_streams);
	hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int LOG2N>
int launch_fft_t(float2* data, const float2* tw, size_t batch, int inverse, hipStream_t stream)
{
	using PL = Plan<LOG2N>;
	dim3 grid((unsigned)ceil_div(batch, (size_t)PL::FRAMES_PER_BLOCK));
	if (inverse) {
		auto kern = fft_kernel<LOG2N, true>;
		ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
		hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, data, tw, (int)batch);
	}
	else {
		auto kern = fft_kernel<LOG2N, false>;
		ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
		hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, data, tw, (int)batch);
	}
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace

int launch_stft(int log2n, const StftArgs& a, hipStream_t stream)
{
	if (a.n_frames <= 0)
		return ZEN_HIP_OK;
#define CALL(L) launch_stft_t<L>(a, stream)
	ZH_DISPATCH_LOG2N(log2n, CALL)
#undef CALL
}

int launch_fft(int log2n, float2* data, const float2* tw, size_t batch, int inverse, hipStream_t stream)
{
	if (batch == 0)
		return ZEN_HIP_OK;
#define CALL(L) launch_fft_t<L>(data, tw, batch, inverse, stream)
	ZH_DISPATCH_LOG2N(log2n, CALL)
#undef CALL
}

int launch_finalize(const FinalizeArgs& a, hipStream_t stream)
{
	if (a.n_frames <= 0)
		return ZEN_HIP_OK;
	const bool vec = a.hop % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && (a.n_streams <= 1 || a.out_stride % 4 == 0);
	const long long n = (long long)a.n_frames * a.hop / (vec ? 4 : 1);
	long long blocks = (n + 255) / 256;
	if (blocks > 4096)
		blocks = 4096;
	if (vec)
		hipLaunchKernelGGL(finalize_kernel<true>, dim3((unsigned)blocks, (unsigned)a.n_streams), dim3(256), 0, stream, a);
	else
		hipLaunchKernelGGL(finalize_kernel<false>, dim3((unsigned)blocks, (unsigned)a.n_streams), dim3(256), 0, stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

int launch_finalize_spec(const FinalizeArgs& a, hipStream_t stream)
{
	if (a.n_frames <= 0)
		return ZEN_HIP_OK;
	const bool vec = a.hop % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && (a.n_streams <= 1 || a.out_stride % 4 == 0)
	                 && a.shift % 4 == 0 && a.pos0 % 4 == 0;
	const long long n = (long long)a.n_frames * a.hop / (vec ? 4 : 1);
	long long blocks = (n + 255) / 256;
	if (blocks > 4096)
		blocks = 4096;
	if (vec)
		hipLaunchKernelGGL(finalize_spec_kernel<true>, dim3((unsigned)blocks, (unsigned)a.n_streams), dim3(256), 0, stream, a);
	else
		hipLaunchKernelGGL(finalize_spec_kernel<false>, dim3((unsigned)blocks, (unsigned)a.n_streams), dim3(256), 0, stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace zen_hip_impl
