// fft_dev.h -- LDS-resident power-of-two complex FFT for gfx950, bit-identical to the oracle's
// recursive radix-2 decimation-in-time transform (oracle/zen_oracle.c fft_rec; the reference's
// FFTC2CWrapper, libzen/fftw.h:20-129, is an unnormalised C2C DFT).
//
// Arithmetic DAG.  Let Y_s[j][k] be the size-2^s DFT of the decimated sequence x[j + n*N/2^s]:
//     Y_0[j][0]          = x[j]
//     Y_s[j][k]          = Y_{s-1}[j][k] + w * Y_{s-1}[j + N/2^s][k]
//     Y_s[j][k + 2^(s-1)] = Y_{s-1}[j][k] - w * Y_{s-1}[j + N/2^s][k],   w = tw[k * N/2^s]
// which is exactly what the oracle's recursion evaluates (even/odd split, one twiddle table of N/2
// entries).  Every thread owns 16 complex values and runs r = 2..4 consecutive stages on them in
// registers (a "radix-2^r pass" that is still the radix-2 DAG: no 3-multiply radix-4 shortcuts, so every
// rounding matches).  Between passes the values are exchanged through LDS in the autosort layout
// addr(level s) = k * (N/2^s) + j, so that pass 0 reads the input in natural order (coalesced, straight
// from HBM) and the last pass writes the spectrum in natural order (coalesced, straight to HBM).
//
// Twiddles come from the host table (exact octant symmetry): tw[i + N/4] == -i*tw[i] bit for bit, so
// half of each stage's twiddles are formed by a swap/negate instead of a load.
//
// Two shortcuts skip arithmetic whose value is known: the first stage of a zero-padded frame (ZU: A +- w*0 = A)
// and the products with the twiddles 1 and -i of a first pass (butterfly<..., TRIV>).  Every nonzero value is
// still the oracle's; an exact zero may carry the other sign.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace zfft {

template <int V>
struct Log2 {
	static constexpr int value = 1 + Log2<V / 2>::value;
};
template <>
struct Log2<1> {
	static constexpr int value = 0;
};

// Complex values per thread: 16 everywhere, except that the 16384-point transform may run on 512 threads of 32 values
// (-DZEN_FFT16K_V=32): three passes (radix 32, 32, 16) and two exchanges through LDS instead of four and three, 8
// wavefronts per frame instead of 16 -- the same radix-2 DAG, the same values.  A/B: build.py, DESIGN.md section 8.
#ifndef ZEN_FFT16K_V
#define ZEN_FFT16K_V 16
#endif

template <int LOG2N>
struct Plan {
	static_assert(LOG2N >= 5 && LOG2N <= 14, "nfft 32..16384");
	static_assert(ZEN_FFT16K_V == 16 || ZEN_FFT16K_V == 32, "16 or 32 values per thread");
	static constexpr int N = 1 << LOG2N;
	static constexpr int V = LOG2N == 14 ? ZEN_FFT16K_V : 16; // complex values per thread
	static constexpr int LOG2V = V == 32 ? 5 : 4;
	static constexpr int P = (LOG2N + LOG2V - 1) / LOG2V; // passes
	static constexpr int BASE = LOG2N / P, REM = LOG2N % P;
	static constexpr int r(int p) { return BASE + (p < REM ? 1 : 0); } // stages in pass p
	static constexpr int s(int p) // stages completed before pass p
	{
		int a = 0;
		for (int i = 0; i < p; ++i)
			a += r(i);
		return a;
	}
	static constexpr int TF = N / V;             // threads per frame: V complex values each
	// One 8-byte slot of padding per 2^PAD_SHIFT keeps the stride-R accesses of the late passes off a single bank: per
	// 16 up to nfft 4096 (34.8 KB per 256 threads: four workgroups per CU); per 8 at nfft 8192 / 16384, where a late pass
	// reads runs of 8 values (with one slot per 16 two such runs share banks) and a CU holds two / one frame anyway (73.7
	// / 147 KB): synthesis -3.5 % (soft masks -10 %), analysis -2.6 % per offline batch step.  At nfft 1024 / 2048 the
	// same padding gains 3.5 % in the analysis and loses 20 % in the soft-mask synthesis (its build then spills); at
	// nfft 4096 it costs the fused kernel 11 % (its lean layout lives inside the image); no padding at all: +25..60 %.
#ifndef ZEN_FFT16K_PAD
#define ZEN_FFT16K_PAD 3
#endif
	static constexpr int PAD_SHIFT = LOG2N == 14 ? ZEN_FFT16K_PAD : (LOG2N >= 13 ? 3 : 4);
#ifdef ZEN_FFT_SWIZZLE // A/B (re-measured in round 5): an unpadded image, element i at i ^ ((i >> 3) & 31)
	static constexpr bool SWZ = LOG2N >= 13;
#else
	static constexpr bool SWZ = false;
#endif
	// In-place exchanges (PassRunner; -DZEN_FFT_INPLACE, the 16384-point plan of 4 + 4 + 3 + 3 stages): digit d_0 of the index
	// (bits 13..10) is what consecutive lanes differ in during the last two passes, so the unpadded image is XOR-swizzled
	// with those bits: slot bits [4:3] ^= i[11:10], [2] ^= i[13], [1] ^= i[12], [0] ^= i[6] -- every wave instruction of
	// every pass then touches each of the 32 eight-byte bank pairs exactly twice (worked out per pass in DESIGN.md section 8)
#if defined(ZEN_FFT_INPLACE)
	static constexpr bool INPLACE = LOG2N == 14 && V == 16;
#else
	static constexpr bool INPLACE = false;
#endif
#if defined(ZEN_FFT_INPLACE)
	static constexpr int LDS_FLOAT2 = (SWZ || INPLACE) ? N : N + (N >> PAD_SHIFT);
	static __device__ __forceinline__ int pad(int i)
	{
		if (INPLACE)
			return i ^ ((((i >> 10) & 3) << 3) | (((i >> 13) & 1) << 2) | (((i >> 12) & 1) << 1) | ((i >> 6) & 1));
		return SWZ ? (i ^ ((i >> 3) & 31)) : i + (i >> PAD_SHIFT);
	}
#else
	static constexpr int LDS_FLOAT2 = SWZ ? N : N + (N >> PAD_SHIFT); // padded frame image in LDS
	static __device__ __forceinline__ int pad(int i) { return SWZ ? (i ^ ((i >> 3) & 31)) : i + (i >> PAD_SHIFT); }
#endif
	// pad(base + off) for a compile-time offset: where off is a multiple of the padding period its share of the padded address
	// is a constant of the access (an immediate of the LDS instruction) and the shift / add runs once per base, not once per
	// access.  Per translation unit (ZEN_FFT_FOLD_ADDR): the kernels that sit at a register limit keep the code they have.
	static __device__ __forceinline__ int pad_off(int base, int off)
	{
#ifdef ZEN_FFT_FOLD_ADDR
		if (!SWZ && (off & ((1 << PAD_SHIFT) - 1)) == 0)
			return pad(base) + off + (off >> PAD_SHIFT);
#endif
		return pad(base + off);
	}
	static constexpr int FRAMES_PER_BLOCK = (TF >= 256) ? 1 : 256 / TF;
	static constexpr int THREADS = TF * FRAMES_PER_BLOCK;
};

// (Plan::pad: the padded position of element i of a frame image; an unpadded XOR-swizzled image, 5 instead of 4
// workgroups per CU, measured no faster)

// PK: two-lane float arithmetic, which the backend maps to the packed instructions (v_pk_mul_f32 / v_pk_add_f32: two
// results per lane and cycle); every lane's product and sum is rounded separately, as in the scalar form.  The
// throughput kernels use it (TwGlobal: fused launch 0.544 -> 0.532 ms, nfft-16384 transforms -1..3 %); the single-hop
// kernels (TwRegs: all registers in use, one wavefront per SIMD) do not -- the aligned register pairs the packed
// operands need made them spill 30 to 40 registers and cost 2 us per hop.
typedef float v2f __attribute__((ext_vector_type(2)));

template <bool PK>
__device__ __forceinline__ float2 cmul(float2 w, float2 b)
{
	// oracle: tr = wr*br - wi*bi ; ti = wr*bi + wi*br   (each product and sum rounded separately)
	if constexpr (PK) {
		const v2f t1 = (v2f){w.x, w.x} * (v2f){b.x, b.y};
		const v2f t2 = (v2f){w.y, w.y} * (v2f){b.y, b.x};
		// r = (t1.x - t2.x, t1.y + t2.y): one packed add with the first lane's second operand negated (a - b == a + (-b)
		// exactly).  Written as t1 + (v2f){-t2.x, t2.y} the backend negates BOTH lanes with an extra packed add and moves
		// the second one back (two more instructions per complex product: a fifth of a transform's VALU work); the
		// half-negation is a source modifier of the instruction, spelled out here.
		v2f r;
		asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(t1), "v"(t2));
		return make_float2(r.x, r.y);
	}
	else {
		return make_float2(w.x * b.x - w.y * b.y, w.x * b.y + w.y * b.x);
	}
}
template <bool PK>
__device__ __forceinline__ float2 cadd(float2 a, float2 b)
{
	if constexpr (PK) {
		const v2f r = (v2f){a.x, a.y} + (v2f){b.x, b.y};
		return make_float2(r.x, r.y);
	}
	else {
		return make_float2(a.x + b.x, a.y + b.y);
	}
}
template <bool PK>
__device__ __forceinline__ float2 csub(float2 a, float2 b)
{
	if constexpr (PK) {
		const v2f r = (v2f){a.x, a.y} - (v2f){b.x, b.y};
		return make_float2(r.x, r.y);
	}
	else {
		return make_float2(a.x - b.x, a.y - b.y);
	}
}

// Where a pass takes its twiddles from.  TwGlobal: the table in global memory, loaded where they are used (one
// dependent L2 / HBM round trip per pass).  TwRegs: every twiddle a thread will need in any pass of a forward
// or inverse transform of this size, loaded once up front (the single-hop kernel, where six such round trips are
// a seventh of the call); slot = position of the load inside a pass: stage q, frequency group c < max(1, 2^(q-2)).
struct TwGlobal {
	static constexpr bool PLAIN = true; // the table of the transform itself: entry 0 is 1 (see butterfly)
	static constexpr bool PACKED = true; // throughput kernels: packed butterflies (cmul)
	const float2* __restrict__ p;
	__device__ __forceinline__ float2 get(int, int, int, int idx) const { return p[idx]; }
};

// The table in global memory, every twiddle of a pass requested at the TOP of the pass (PassRunner), before the pass reads
// its inputs from LDS: left to the scheduler the loads of a stage sit in front of that stage, each waited for on the spot --
// up to five dependent trips to the L2 per pass in the kernels whose transform sits in a loop (istft_run_wide_kernel).
struct TwGlobalPre {
	static constexpr bool PLAIN = true;
	static constexpr bool PACKED = true;
	static constexpr bool PRELOAD = true;
	const float2* __restrict__ p;
	__device__ __forceinline__ float2 get(int, int, int, int idx) const { return p[idx]; }
};
// one pass's twiddles in registers (TwRegs for a single pass; the slots butterfly() never asks for are never loaded)
template <int LOG2N, int PASS, bool PACKED_>
struct TwPassRegs {
	static constexpr bool PLAIN = true;
	static constexpr bool PACKED = PACKED_;
	using PL = Plan<LOG2N>;
	static constexpr int rr = PL::r(PASS), R = 1 << rr, NB = PL::V / R, sL = PL::s(PASS), log2J = LOG2N - sL - rr;
	float2 w[NB][8];
	__device__ __forceinline__ float2 get(int, int i, int slot, int) const { return w[i][slot]; }
	__device__ __forceinline__ void fill(int tf, const float2* __restrict__ p)
	{
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			const int k = (tf + i * PL::TF) >> log2J;
#pragma unroll
			for (int q = 1; q <= rr; ++q) {
				const int nload = q == 1 ? 1 : (1 << (q - 2));
#pragma unroll
				for (int c = 0; c < nload; ++c)
					w[i][(q == 1 ? 0 : (1 << (q - 2))) + c] = p[(k << (LOG2N - sL - q)) + (c << (LOG2N - q))];
			}
		}
	}
};
template <class T, class = void>
struct tw_preloads : std::false_type {};
template <class T>
struct tw_preloads<T, std::enable_if_t<T::PRELOAD>> : std::true_type {};

// the table staged in LDS by the kernel (synthesis at nfft <= 2048: a table read costs the LDS a cycle, not the vector
// memory path 8 bytes per lane -- per frame the three passes ask it for 12 KB of twiddles, more than the spectrum)
struct TwLds {
	static constexpr bool PLAIN = true;
	static constexpr bool PACKED = true;
	const float2* p;
	__device__ __forceinline__ float2 get(int, int, int, int idx) const { return p[idx]; }
};

// PLAIN = false: the registers hold something else than the transform's own table (rt_wide.hip's second step)
template <int LOG2N, bool PLAIN_ = true>
struct TwRegs {
	static constexpr bool PLAIN = PLAIN_;
	static constexpr bool PACKED = false; // single-hop kernels: scalar butterflies (cmul)
	using PL = Plan<LOG2N>;
	static constexpr int NBMAX = PL::V >> PL::BASE; // groups per thread in the pass with the fewest stages
	float2 w[PL::P][NBMAX][8];
	__device__ __forceinline__ float2 get(int pass, int i, int slot, int) const { return w[pass][i][slot]; }
	template <int PASS = 0>
	__device__ __forceinline__ void fill(int tf, const float2* __restrict__ p)
	{
		constexpr int rr = PL::r(PASS), R = 1 << rr, NB = PL::V / R, sL = PL::s(PASS), log2J = LOG2N - sL - rr;
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			const int k = (tf + i * PL::TF) >> log2J;
#pragma unroll
			for (int q = 1; q <= rr; ++q) {
				const int nload = q == 1 ? 1 : (1 << (q - 2));
#pragma unroll
				for (int c = 0; c < nload; ++c)
					w[PASS][i][(q == 1 ? 0 : (1 << (q - 2))) + c] = p[(k << (LOG2N - sL - q)) + (c << (LOG2N - q))];
			}
		}
		if constexpr (PASS + 1 < PL::P)
			fill<PASS + 1>(tf, p);
	}
	// the same for a transform that is one step of a larger one (rt_wide.hip): f(stage, idx) maps the table index
	// of this 2^LOG2N-point transform's stage (1-based) to the entry of the large transform's table
	template <int PASS = 0, class F>
	__device__ __forceinline__ void fill_with(int tf, const F& f)
	{
		constexpr int rr = PL::r(PASS), R = 1 << rr, NB = PL::V / R, sL = PL::s(PASS), log2J = LOG2N - sL - rr;
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			const int k = (tf + i * PL::TF) >> log2J;
#pragma unroll
			for (int q = 1; q <= rr; ++q) {
				const int nload = q == 1 ? 1 : (1 << (q - 2));
#pragma unroll
				for (int c = 0; c < nload; ++c)
					w[PASS][i][(q == 1 ? 0 : (1 << (q - 2))) + c] = f(sL + q, (k << (LOG2N - sL - q)) + (c << (LOG2N - q)));
			}
		}
		if constexpr (PASS + 1 < PL::P)
			fill_with<PASS + 1>(tf, f);
	}
};

// r = log2(R) radix-2 DIT stages on R values held by one thread.
//   in : a[m] = Y_s[j + m*J][k], m < R            (J = N / (2^s * R))
//   out: a[c] = Y_{s+r}[j][k + c*2^s], c < R
// ZU: a[R/2..R) are known zeros (zero-padded analysis frame): the first stage is then a copy.
//
// Trivial twiddles.  In the first pass of a transform (k == 0) frequency group c = 0 of every stage has the
// twiddle tw[0] = 1 and, from the second stage on, group c = nc/2 has tw[N/4] = -i (the table has that symmetry
// exactly): w * B is then B or (B.y, -B.x) -- what the general complex product evaluates to as well (x*1 = x,
// x*0 = +-0, y +- 0 = y), except for the SIGN of an exact zero result (and 0 * inf).  The same kind of shortcut
// as ZU; it removes 22 of the 32 complex products of a 16-value first pass.  TRIV: first pass, plain table.
template <int R, bool INV, bool ZU, bool TRIV, class TW>
__device__ __forceinline__ void butterfly(float2 (&a)[R], int k, int log2L, int log2N, const TW& tw, int pass, int grp)
{
	constexpr int r = Log2<R>::value;
	float2 b[R];
#pragma unroll
	for (int q = 1; q <= r; ++q) {
		const int half = R >> q;     // sub-sequences left after this stage
		const int nc = 1 << (q - 1); // frequency groups entering this stage
		float2 w[(R / 2) > 0 ? (R / 2) : 1];
#pragma unroll
		for (int c = 0; c < nc; ++c) {
			if (q >= 2 && c >= (nc >> 1)) { // index + N/4  ==  multiply by -i (forward) / +i (inverse)
				float2 w0 = w[c - (nc >> 1)];
				w[c] = INV ? make_float2(-w0.y, w0.x) : make_float2(w0.y, -w0.x);
			}
			else if (TRIV && c == 0) {
				w[c] = make_float2(1.0f, 0.0f); // not loaded, not multiplied with (below)
			}
			else {
				int idx = (k << (log2N - log2L - q)) + (c << (log2N - q));
				float2 t = tw.get(pass, grp, (q == 1 ? 0 : (1 << (q - 2))) + c, idx);
				w[c] = INV ? make_float2(t.x, -t.y) : t;
			}
		}
#pragma unroll
		for (int c = 0; c < nc; ++c) {
#pragma unroll
			for (int m = 0; m < half; ++m) {
				float2 A = a[c * 2 * half + m];
				if (ZU && q == 1) {
					b[c * half + m] = A; // A + w*0, A - w*0
					b[(c + nc) * half + m] = A;
				}
				else if (TRIV && c == 0) { // w = 1
					const float2 B = a[c * 2 * half + m + half];
					b[c * half + m] = cadd<TW::PACKED>(A, B);
					b[(c + nc) * half + m] = csub<TW::PACKED>(A, B);
				}
				else if (TRIV && q >= 2 && c == (nc >> 1)) { // w = -i (forward), +i (inverse): t = (B.y, -B.x) / (-B.y, B.x)
					const float2 B = a[c * 2 * half + m + half];
					if constexpr (TW::PACKED) { // A + / - (B.y, -B.x) as one packed add each: lanes of B swapped, one negated
						const v2f Av = (v2f){A.x, A.y}, Bv = (v2f){B.x, B.y};
						v2f p, n; // p = (A.x + B.y, A.y - B.x), n = (A.x - B.y, A.y + B.x)
						asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(p) : "v"(Av), "v"(Bv));
						asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(n) : "v"(Av), "v"(Bv));
						b[c * half + m] = INV ? make_float2(n.x, n.y) : make_float2(p.x, p.y);
						b[(c + nc) * half + m] = INV ? make_float2(p.x, p.y) : make_float2(n.x, n.y);
					}
					else if (INV) {
						b[c * half + m] = make_float2(A.x - B.y, A.y + B.x);
						b[(c + nc) * half + m] = make_float2(A.x + B.y, A.y - B.x);
					}
					else {
						b[c * half + m] = make_float2(A.x + B.y, A.y - B.x);
						b[(c + nc) * half + m] = make_float2(A.x - B.y, A.y + B.x);
					}
				}
				else {
					float2 t = cmul<TW::PACKED>(w[c], a[c * 2 * half + m + half]);
					b[c * half + m] = cadd<TW::PACKED>(A, t);
					b[(c + nc) * half + m] = csub<TW::PACKED>(A, t);
				}
			}
		}
#pragma unroll
		for (int i = 0; i < R; ++i)
			a[i] = b[i];
	}
}

// Synchronisation between the LDS phases of a pass.  A frame is transformed by TF = N/16 threads; when those are
// one wavefront or less (nfft <= 1024: several frames per workgroup, each in its own wavefront) nobody outside the
// wavefront touches the frame's image: a wavefront executes its LDS instructions in order, so only the compiler has
// to be kept from moving accesses across the phase boundary -- no s_barrier that would hold four independent
// frames in lock step.
template <int TF>
__device__ __forceinline__ void frame_sync()
{
	if constexpr (TF <= 64) {
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
	else {
#ifdef ZEN_FFT_LDS_BARRIER
		// the frame's threads exchange through LDS only: its counter drained, then the barrier.  __syncthreads() is a fence as
		// well, which on gfx950 also waits for every global load and store in flight (s_waitcnt vmcnt(0)): the twiddles a pass
		// has asked for, the hop a run kernel stores in the background
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
		__syncthreads();
#endif
	}
}

// An input functor may split its work: Raw load(idx, slot) (memory only), prepare() and float2 finish(raw, idx, slot).  The first
// pass then issues the loads of all its elements before it finishes the first one, whatever the scheduler would have
// made of in(): sixteen loads in flight instead of a trip to memory per element (see istft.hip IstftIn).
template <class T, class = void>
struct has_split_input : std::false_type {};
template <class T>
struct has_split_input<T, std::void_t<typename T::Raw>> : std::true_type {};

// SYNC_FIRST: in() of the first pass reads LDS that the pass's own stores may overwrite (the fused kernel
// keeps |S| and P inside the frame image): a barrier separates the two, as in every later pass.
template <int LOG2N, int PASS, bool INV, bool ZU, bool HALF_OUT, class In, class Out, bool SYNC_FIRST = false,
          class TW = TwGlobal>
struct PassRunner {
	using PL = Plan<LOG2N>;
	static __device__ __forceinline__ void run(int tf, float2* __restrict__ lds, const TW& tw, In& in, Out& out, bool active)
	{
		constexpr int N = PL::N, TF = PL::TF;
		constexpr int rr = PL::r(PASS), R = 1 << rr, NB = PL::V / R;
		constexpr int sL = PL::s(PASS);                    // log2 of sub-DFT size entering the pass
		constexpr int log2J = LOG2N - sL - rr, J = 1 << log2J; // sub-sequences left after the pass
		constexpr bool FIRST = PASS == 0, LAST = PASS == PL::P - 1;
		constexpr bool ZUP = ZU && FIRST;
		constexpr bool PRE = tw_preloads<TW>::value;
		// In-place exchanges (-DZEN_FFT_INPLACE, per translation unit).  The autosort layout writes a pass's results where the
		// NEXT pass wants them (addr = k * (N/2^s) + j), other positions than the pass read: a barrier between its reads and its
		// writes, and one behind the writes.  In place, output c of a sub-transform goes where input m = c came from; a thread
		// then overwrites nothing but what it read itself and the first of the two barriers is only needed in the last pass
		// (whose reads the next transform's first pass must not overtake): P barriers per transform instead of 2P - 2.  Same
		// butterflies, same operands, same twiddles -- the values do not know where they are kept.  The price is the address of
		// a later pass's inputs: the digits c_0 .. c_{p-1} the earlier passes produced sit in the image in the order they were
		// made (c_0 on top), the reverse of their order in k.
#ifdef ZEN_FFT_INPLACE
		constexpr bool INPL = PL::INPLACE;
		auto image_k = [](int k) -> int { // where the sub-transforms of output index k sit in the image
			if constexpr (!INPL || PASS < 2) {
				return k;
			}
			else {
				int rv = 0;
#pragma unroll
				for (int g = 0; g < PASS; ++g)
					rv |= ((k >> PL::s(g)) & ((1 << PL::r(g)) - 1)) << (sL - PL::s(g + 1));
				return rv;
			}
		};
#endif
		TwPassRegs<LOG2N, PASS, TW::PACKED> twp;
		if constexpr (PRE)
			twp.fill(tf, tw.p);

		float2 v[NB][R];
		if constexpr (FIRST && !ZUP && has_split_input<In>::value) {
			typename In::Raw raw[NB][R];
#pragma unroll
			for (int i = 0; i < NB; ++i)
#pragma unroll
				for (int m = 0; m < R; ++m)
					raw[i][m] = in.load(m * J + ((tf + i * TF) & (J - 1)), m * NB + i);
			in.prepare(); // (whatever else the functor has to wait for: behind the loads, not in front of them)
#pragma unroll
			for (int i = 0; i < NB; ++i)
#pragma unroll
				for (int m = 0; m < R; ++m)
					v[i][m] = in.finish(raw[i][m], m * J + ((tf + i * TF) & (J - 1)), m * NB + i);
		}
		else
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			const int b = tf + i * TF;
#ifdef ZEN_FFT_INPLACE
			const int k = image_k(b >> log2J), j = b & (J - 1);
#else
			const int k = b >> log2J, j = b & (J - 1);
#endif
#pragma unroll
			for (int m = 0; m < R; ++m) {
				if (ZUP && m >= R / 2) {
					v[i][m] = make_float2(0.f, 0.f);
				}
				else if (FIRST) {
					// (unconditional: an inactive frame reads valid memory too, see fft_frame -- `active ? in() : 0` put every
					// one of the 16 loads into a branch of its own with a full wait inside: 16 dependent trips to memory)
					v[i][m] = in(m * J + j, /*slot=*/m * NB + i);
				}
				else {
#ifdef ZEN_FFT_FOLD_ADDR
					v[i][m] = lds[PL::pad_off(k * R * J + j, m * J)];
#else
					v[i][m] = lds[PL::pad((k * R + m) * J + j)];
#endif
				}
			}
		}
#ifdef ZEN_FFT_INPLACE
		if ((FIRST && SYNC_FIRST) || (!FIRST && (!INPL || LAST)))
			frame_sync<TF>(); // (in place: only the next transform's first pass overwrites what others read)
#else
		if (!FIRST || SYNC_FIRST)
			frame_sync<TF>(); // every thread has its inputs in registers: LDS may be overwritten
#endif
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			const int b = tf + i * TF;
			const int k = b >> log2J;
			if constexpr (PRE)
				butterfly<R, INV, ZUP, (FIRST && TW::PLAIN)>(v[i], k, sL, LOG2N, twp, PASS, i);
			else
				butterfly<R, INV, ZUP, (FIRST && TW::PLAIN)>(v[i], k, sL, LOG2N, tw, PASS, i);
#pragma unroll
			for (int c = 0; c < R; ++c) {
				const int idx = b + c * (N / R);
				if (LAST) {
					if (!HALF_OUT || c < R / 2) { // idx < N/2  <=>  c < R/2
						if (active)
							out(idx, v[i][c], /*lower_half=*/c < R / 2, /*slot=*/c * NB + i);
					}
				}
#ifdef ZEN_FFT_INPLACE
				else if constexpr (INPL) {
					lds[PL::pad((image_k(k) * R + c) * J + (b & (J - 1)))] = v[i][c];
				}
#endif
				else {
#ifdef ZEN_FFT_FOLD_ADDR
					lds[PL::pad_off(b, c * (N / R))] = v[i][c];
#else
					lds[PL::pad(idx)] = v[i][c];
#endif
				}
			}
		}
		if constexpr (!LAST) {
			frame_sync<TF>();
			PassRunner<LOG2N, PASS + 1, INV, ZU, HALF_OUT, In, Out, SYNC_FIRST, TW>::run(tf, lds, tw, in, out, active);
		}
	}
};

// One N-point transform by the TF threads that own a frame.  `lds` is that frame's padded LDS image
// (Plan::LDS_FLOAT2 float2).  in(idx, slot) -> float2 supplies x[idx] (only idx < N/2 is asked for when ZU);
// out(idx, X, lower, slot) receives X[idx] (only idx < N/2 when HALF_OUT); `lower` is the compile-time
// fact idx < N/2.  `slot` (compile-time, 0..15) numbers a thread's 16 values: idx = tf + slot*TF both for
// the first pass's inputs and the last pass's outputs, so a spectrum can stay in registers between a
// forward and an inverse transform (rt_fused.hip).  All threads of the block must call this
// together (it contains block barriers); inactive frames pass active = false: out() is not called for them, in() IS
// (its results are dropped), so their `in` must point at readable memory -- the callers clamp the frame index.
template <int LOG2N, bool INV, bool ZU, bool HALF_OUT, class In, class Out, bool SYNC_FIRST = false>
__device__ __forceinline__ void fft_frame(int tf, float2* __restrict__ lds, const float2* __restrict__ tw,
                                          In& in, Out& out, bool active)
{
	const TwGlobal g{tw};
	PassRunner<LOG2N, 0, INV, ZU, HALF_OUT, In, Out, SYNC_FIRST, TwGlobal>::run(tf, lds, g, in, out, active);
}
// the same with every pass's twiddles requested at its top (TwGlobalPre)
template <int LOG2N, bool INV, bool ZU, bool HALF_OUT, class In, class Out, bool SYNC_FIRST = false>
__device__ __forceinline__ void fft_frame(int tf, float2* __restrict__ lds, const TwGlobalPre& tw, In& in, Out& out, bool active)
{
	PassRunner<LOG2N, 0, INV, ZU, HALF_OUT, In, Out, SYNC_FIRST, TwGlobalPre>::run(tf, lds, tw, in, out, active);
}
// the same with the table in LDS (TwLds)
template <int LOG2N, bool INV, bool ZU, bool HALF_OUT, class In, class Out, bool SYNC_FIRST = false>
__device__ __forceinline__ void fft_frame(int tf, float2* __restrict__ lds, const TwLds& tw, In& in, Out& out, bool active)
{
	PassRunner<LOG2N, 0, INV, ZU, HALF_OUT, In, Out, SYNC_FIRST, TwLds>::run(tf, lds, tw, in, out, active);
}
// the same with the twiddles already in registers (TwRegs::fill)
template <int LOG2N, bool INV, bool ZU, bool HALF_OUT, class In, class Out, bool SYNC_FIRST = false>
__device__ __forceinline__ void fft_frame(int tf, float2* __restrict__ lds, const TwRegs<LOG2N>& tw, In& in, Out& out,
                                          bool active)
{
	PassRunner<LOG2N, 0, INV, ZU, HALF_OUT, In, Out, SYNC_FIRST, TwRegs<LOG2N>>::run(tf, lds, tw, in, out, active);
}

// |z| exactly as the oracle's zo_cabs: (float)sqrt((double)re*re + (double)im*im).
// The square root is the sequence the compiler emits for a correctly rounded double sqrt (rsq, one Goldschmidt
// step on g ~ sqrt and h ~ 1/(2 sqrt), two residual corrections) WITHOUT its range scaling: the sum of two squared
// floats is 0 or lies in [2^-298, 2^257), far from the doubles (< 2^-767) the scaling is for; that is a compare, a
// select and two ldexp less per value, all at the half rate of double precision (tools/check_sqrt.hip compares it
// with sqrt() on 3e10 pairs of every kind the path can produce).
__device__ __forceinline__ double sqrt_of_sum_of_squares(double x)
{
	const double y = __builtin_amdgcn_rsq(x);
	double g = x * y;
	double h = 0.5 * y;
	const double r = __builtin_fma(-h, g, 0.5);
	g = __builtin_fma(g, r, g);
	h = __builtin_fma(h, r, h);
	double d = __builtin_fma(-g, g, x);
	g = __builtin_fma(d, h, g);
	d = __builtin_fma(-g, g, x);
	g = __builtin_fma(d, h, g);
	return (x == 0.0 || x == __builtin_inf()) ? x : g; // rsq(0) = inf, rsq(inf) = 0: the products above are NaN there
}

__device__ __forceinline__ float cabs_exact(float re, float im)
{
	const double r = (double)re, i = (double)im;
	// (double)re * re is exact (24 x 24 bits), so fma(r, r, i*i) rounds the same real number as r*r + i*i does: one
	// double-precision instruction less, the same bits
	return (float)sqrt_of_sum_of_squares(__builtin_fma(r, r, i * i));
}

} // namespace zfft
