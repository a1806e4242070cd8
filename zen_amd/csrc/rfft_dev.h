// rfft_dev.h -- forward transform of a REAL frame: the Hermitian half of fft_dev.h's radix-2 DIT DAG, bit for bit.
//
// (libzen/hps.cu:456-465: the analysis multiplies the frame by a real window, zeroes the imaginary parts and runs the
// forward C2C transform; oracle/zen_oracle.c fft_rec is the DAG.)
//
// With real x every sub-transform Y_s[j][.] of the DAG (fft_dev.h) is exactly Hermitian, Y_s[j][2^s - k] == conj(Y_s[j][k])
// in every bit: by induction over the stages, with h = 2^(s-1), A = Y_{s-1}[j], B = Y_{s-1}[j + N/2^s],
//     Y_s[k]     = A[k] + w_k B[k]                       w_k = tw[k N/2^s]
//     Y_s[h + k] = A[k] - w_k B[k]
// and the host table has tw[N/2 - i] == -conj(tw[i]) exactly, so w_{h-k} B[h-k] = -conj(w_k) conj(B[k]) = -conj(w_k B[k])
// (a product of conjugates rounds to the conjugate of the product, negation is exact), hence
//     Y_s[h - k] = conj(A[k]) - conj(w_k B[k]) = conj(Y_s[h + k]),      Y_s[2h - k] = conj(Y_s[k]).
// So only k = 0..h of every sub-transform is computed and kept: each butterfly (A[k], B[k]), k <= h/2, yields Y[k] = A + wB
// and Y[h - k] = conj(A - wB) -- the oracle's values -- half the butterflies, half the LDS image.  Y[0] and Y[h] are real
// (imaginary parts exact zeros in the oracle; the sign of such a zero is not reproduced, as with fft_dev.h's shortcuts)
// and share one slot, so a level is exactly N/2 complex slots:
//     level s >= 1, sub-sequence j < J_s = N/2^s:  slot kappa in [0, h):  kappa = 0: (Re Y[0], Re Y[h]);  else Y[kappa]
//     address (before padding) kappa * J_s + j
// A thread owns 16 slots (N/32 threads per frame, half of fft_dev.h's) and runs the same r = 2..4 stages per pass on them:
//   * pass 0: the thread's items are R = 2^r real samples x[j + m N/R]; real_dag0 evaluates the R-point sub-DAG at k = 0;
//   * later passes, item (k, j), 1 <= k < h: the R complex values Y_s[j + mJ][k] go through fft_dev.h's own butterfly();
//     outputs c < R/2 are bins k + c 2^s, outputs c >= R/2 are stored conjugated at bins (R - c) 2^s - k;
//   * later passes, item (0, j): the packed slots hold R real values each of k = 0 and k = h: real_dag0 and real_dagh.
// The last pass hands bins 0..N/2 to the output functor in natural order.
#pragma once
#include "fft_dev.h"

#pragma clang fp contract(off)

namespace zfft {

template <int LOG2N>
struct RPlan {
	using PL = Plan<LOG2N>;
	static_assert(PL::V == 16, "sixteen slots per thread");
	static constexpr int N = PL::N, P = PL::P;
	static constexpr int r(int p) { return PL::r(p); }
	static constexpr int s(int p) { return PL::s(p); }
	static constexpr int TF = N / 32;    // threads per frame
	static constexpr int SLOTS = N / 2;  // complex slots per level
#ifndef ZEN_RFFT_PAD16K
#define ZEN_RFFT_PAD16K 3
#endif
	static constexpr int PAD_SHIFT = LOG2N >= 14 ? ZEN_RFFT_PAD16K : 4; // (the image of a 2^L-point real frame is that of a 2^(L-1)-point complex one)
	static constexpr int LDS_FLOAT2 = SLOTS + (SLOTS >> PAD_SHIFT);
	static __device__ __forceinline__ int pad(int i) { return i + (i >> PAD_SHIFT); }
	// pad(base + off) for an offset that is a multiple of 2^PAD_SHIFT (or a base that is zero): the offset's share is a
	// compile-time constant of the access (base may be negative as long as the sum is not: the shift is arithmetic)
	static __device__ __forceinline__ int pad_off(int base, int off)
	{
		return (off & ((1 << PAD_SHIFT) - 1)) == 0 ? pad(base) + off + (off >> PAD_SHIFT) : pad(base + off);
	}
#ifndef ZEN_RFFT_BLOCK_THREADS
#define ZEN_RFFT_BLOCK_THREADS 256
#endif
	static constexpr int FRAMES_PER_BLOCK = (TF >= ZEN_RFFT_BLOCK_THREADS) ? 1 : ZEN_RFFT_BLOCK_THREADS / TF;
	static constexpr int THREADS = TF * FRAMES_PER_BLOCK;
};

// (-i w) * b: the product with a twiddle a quarter turn on (tw[i + N/4] == -i tw[i] bit for bit) without forming that twiddle.
// fft_dev.h's butterfly() builds w' = (w.y, -w.x) and multiplies: tr = w'.x b.x - w'.y b.y, ti = w'.x b.y + w'.y b.x.  Here the
// same four products (a negated factor negates the rounded product exactly) meet in one packed add with the signs as
// modifiers: (w.y b.x + w.x b.y, w.y b.y - w.x b.x) -- the same bits, one packed negation per derived twiddle less.
template <bool PK>
__device__ __forceinline__ float2 cmul_negi(float2 w, float2 b)
{
	if constexpr (PK) {
		const v2f t1 = (v2f){w.y, w.y} * (v2f){b.x, b.y};
		const v2f t2 = (v2f){w.x, w.x} * (v2f){b.y, b.x};
		v2f r;
		asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(t1), "v"(t2));
		return make_float2(r.x, r.y);
	}
	else {
		return make_float2(w.y * b.x + w.x * b.y, w.y * b.y - w.x * b.x);
	}
}
// conj(a - b) = (a.x - b.x, b.y - a.y) as one packed add (x - y == -(y - x) exactly): the conjugated outputs of a pass cost
// no sign flips of their own
template <bool PK>
__device__ __forceinline__ float2 csub_conj(float2 a, float2 b)
{
	if constexpr (PK) {
		const v2f av = (v2f){a.x, a.y}, bv = (v2f){b.x, b.y};
		v2f r;
		asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(av), "v"(bv));
		return make_float2(r.x, r.y);
	}
	else {
		return make_float2(a.x - b.x, b.y - a.y);
	}
}

// fft_dev.h's butterfly() for the general items of the real transform: forward, no known-value shortcuts, and the upper half
// of the results -- which the real transform keeps as conjugates, see RPass -- comes out conjugated:
//   in : a[m] = Y_s[j + m*J][k], m < R
//   out: a[c] = Y_{s+r}[j][k + c 2^s], c < R/2;   a[c] = conj(Y_{s+r}[j][k + c 2^s]), c >= R/2
template <int R, class TW>
__device__ __forceinline__ void rbutterfly(float2 (&a)[R], int k, int log2L, int log2N, const TW& tw, int pass, int grp)
{
	constexpr int r = Log2<R>::value;
	constexpr bool PK = TW::PACKED;
	float2 b[R];
#pragma unroll
	for (int q = 1; q <= r; ++q) {
		const int half = R >> q;     // sub-sequences left after this stage
		const int nc = 1 << (q - 1); // frequency groups entering this stage
		float2 w[(R / 4) > 0 ? (R / 4) : 1]; // the loaded twiddles: groups c < max(1, nc/2); group c + nc/2 is a quarter turn on
#pragma unroll
		for (int c = 0; c < (nc >= 2 ? nc / 2 : 1); ++c) {
			const int idx = (k << (log2N - log2L - q)) + (c << (log2N - q));
			w[c] = tw.get(pass, grp, (q == 1 ? 0 : (1 << (q - 2))) + c, idx);
		}
#pragma unroll
		for (int c = 0; c < nc; ++c) {
#pragma unroll
			for (int m = 0; m < half; ++m) {
				const float2 A = a[c * 2 * half + m], B = a[c * 2 * half + m + half];
				const float2 t = (q >= 2 && c >= (nc >> 1)) ? cmul_negi<PK>(w[c - (nc >> 1)], B) : cmul<PK>(w[c], B);
				b[c * half + m] = cadd<PK>(A, t);
				b[(c + nc) * half + m] = q == r ? csub_conj<PK>(A, t) : csub<PK>(A, t);
			}
		}
#pragma unroll
		for (int i = 0; i < R; ++i)
			a[i] = b[i];
	}
}

// R-point sub-DAG of r = log2 R stages on REAL inputs at k = 0 (any level s: the twiddles of frequency group c at the
// pass's stage q are tw[c N/2^q], whatever s is).
//   in : x[m] = Re Y_s[j + m J][0]
//   out: y[c] = Y_{s+r}[j][c 2^s], c = 0..R/2; y[0] and y[R/2] are real (their .y is not set)
// After stage q the groups c = 0..2^(q-1) of the 2^q are kept (c and 2^q - c are conjugates; c = 0 and c = 2^(q-1) real).
// ZU: x[R/2..R) are known zeros (zero-padded frame): the first stage is a copy.
template <int R, bool ZU, bool PK>
__device__ __forceinline__ void real_dag0(const float (&x)[R], float2 (&y)[R / 2 + 1], const float2* __restrict__ tw, int log2N)
{
	constexpr int r = Log2<R>::value;
	float2 a[R], b[R]; // group c, element m of a stage with cnt elements per group: [c * cnt + m]
#pragma unroll
	for (int m = 0; m < R / 2; ++m) { // stage 1: w = 1
		if (ZU) {
			a[m].x = x[m];
			a[R / 2 + m].x = x[m];
		}
		else {
			a[m].x = x[m] + x[m + R / 2];
			a[R / 2 + m].x = x[m] - x[m + R / 2];
		}
		a[m].y = 0.0f;
		a[R / 2 + m].y = 0.0f;
	}
#pragma unroll
	for (int q = 1; q < r; ++q) {      // stage q + 1
		const int cnt = R >> q;        // elements per input group
		const int half = cnt >> 1;     // per output group
		const int H = 1 << (q - 1);    // input groups 0..H
#pragma unroll
		for (int c = 0; c <= H; ++c) {
#pragma unroll
			for (int m = 0; m < half; ++m) {
				const float2 A = a[c * cnt + m], B = a[c * cnt + m + half];
				if (c == 0) { // w = 1, real
					b[m] = make_float2(A.x + B.x, 0.0f);
					b[(2 * H) * half + m] = make_float2(A.x - B.x, 0.0f);
				}
				else if (c == H) { // w = tw[N/4] = -i, A and B real: A + wB = (A, -B), and its partner is its own conjugate
					b[c * half + m] = make_float2(A.x, -B.x);
				}
				else {
					const float2 w = tw[c << (log2N - q - 1)];
					const float2 t = cmul<PK>(w, B);
					b[c * half + m] = cadd<PK>(A, t);
					b[(2 * H - c) * half + m] = csub_conj<PK>(A, t);
				}
			}
		}
#pragma unroll
		for (int i = 0; i < R; ++i)
			a[i] = b[i];
	}
#pragma unroll
	for (int c = 0; c <= R / 2; ++c)
		y[c] = a[c];
}

// The same at k = h = 2^(s-1) (the other real bin of a sub-transform): twiddles tw[(2c + 1) N/2^(q+1)].
//   in : x[m] = Re Y_s[j + m J][h]
//   out: y[c] = Y_{s+r}[j][h + c 2^s], c < R/2 (all complex; group c and 2^q - 1 - c are conjugates)
template <int R, bool PK>
__device__ __forceinline__ void real_dagh(const float (&x)[R], float2 (&y)[R / 2], const float2* __restrict__ tw, int log2N)
{
	constexpr int r = Log2<R>::value;
	float2 a[R], b[R];
#pragma unroll
	for (int m = 0; m < R / 2; ++m) // stage 1: w = tw[N/4] = -i on real values
		a[m] = make_float2(x[m], -x[m + R / 2]);
#pragma unroll
	for (int q = 1; q < r; ++q) {  // stage q + 1: input groups 0..2^(q-1) - 1
		const int cnt = R >> q, half = cnt >> 1, G = 1 << (q - 1);
#pragma unroll
		for (int c = 0; c < G; ++c) {
			const float2 w = tw[(2 * c + 1) << (log2N - q - 2)];
#pragma unroll
			for (int m = 0; m < half; ++m) {
				const float2 A = a[c * cnt + m], B = a[c * cnt + m + half];
				const float2 t = cmul<PK>(w, B);
				b[c * half + m] = cadd<PK>(A, t);
				b[(2 * G - 1 - c) * half + m] = csub_conj<PK>(A, t);
			}
		}
#pragma unroll
		for (int i = 0; i < R / 2; ++i)
			a[i] = b[i];
	}
#pragma unroll
	for (int c = 0; c < R / 2; ++c)
		y[c] = a[c];
}

// An input functor may take over the whole first-pass load of a thread: x[i][m] = sample tf + i*TF + m*J (m < R/2 when ZU, the
// rest zeros), e.g. to choose between two forms of ALL its loads with one wave-uniform branch.
template <class T, class = void>
struct has_first_pass : std::false_type {};
template <class T>
struct has_first_pass<T, std::void_t<decltype(T::HAS_FIRST_PASS)>> : std::true_type {};

// One pass of the real transform, split in two so that the caller places the synchronisation (and a host test can run the
// threads of a frame one after the other): load() brings the thread's inputs into registers, compute() runs the stages and
// stores to the LDS image of the next level or, in the last pass, hands bins 0..N/2 to out(bin, value).
//   In : float in(int idx) -> x[idx] (only idx < N/2 when ZU)
//   Out: void out(int base, int off, float2 X, int slot): bin base + off, off and slot constants after unrolling (a thread's
//        sixteen results use two values of base per item); slot < 16 numbers the thread's outputs, the thread that holds bin 0
//        hands over bin N/2 as slot 16
template <int LOG2N, int PASS, bool ZU, class TW>
struct RPass {
	using RP = RPlan<LOG2N>;
	static constexpr int N = RP::N, TF = RP::TF;
	static constexpr int rr = RP::r(PASS), R = 1 << rr;
	static constexpr int sL = RP::s(PASS);
	static constexpr int log2J = LOG2N - sL - rr, J = 1 << log2J;
	static constexpr bool FIRST = PASS == 0, LAST = PASS == RP::P - 1;
	static constexpr int NI = FIRST ? 32 / R : 16 / R; // items per thread
	static constexpr bool PRE = tw_preloads<TW>::value && !FIRST;
	// the twiddles of the pass's general items in registers (TwPassRegs with this plan's thread count), requested by load()
	// in front of the reads of the image: left to the scheduler they sit in front of their stage, each waited for on the spot
	struct TwP {
		static constexpr bool PLAIN = true;
		static constexpr bool PACKED = TW::PACKED;
		float2 w[PRE ? NI : 1][8];
		__device__ __forceinline__ float2 get(int, int i, int slot, int) const { return w[i][slot]; }
	};
	struct Regs {
		float x[FIRST ? NI : 1][FIRST ? R : 1];    // pass 0: real samples
		float2 v[FIRST ? 1 : NI][FIRST ? 1 : R];   // later passes: slots
		TwP tw;
	};

	template <class In>
	static __device__ __forceinline__ void load(int tf, const float2* __restrict__ lds, In& in, Regs& g, const float2* __restrict__ tw_p = nullptr)
	{
		if constexpr (FIRST && has_first_pass<In>::value) {
			in.template first_pass<NI, R, J, TF, ZU>(tf, g.x); // (a functor that wants to see all of a thread's loads at once)
		}
		else if constexpr (FIRST) {
#pragma unroll
			for (int i = 0; i < NI; ++i)
#pragma unroll
				for (int m = 0; m < R; ++m)
					g.x[i][m] = (ZU && m >= R / 2) ? 0.0f : in(tf + i * TF + m * J);
		}
		else {
			if constexpr (PRE) {
#pragma unroll
				for (int i = 0; i < NI; ++i) {
					const int k = (tf + i * TF) >> log2J;
#pragma unroll
					for (int q = 1; q <= rr; ++q) {
						const int nload = q == 1 ? 1 : (1 << (q - 2));
#pragma unroll
						for (int c = 0; c < nload; ++c)
							g.tw.w[i][(q == 1 ? 0 : (1 << (q - 2))) + c] = tw_p[(k << (LOG2N - sL - q)) + (c << (LOG2N - q))];
					}
				}
			}
#pragma unroll
			for (int i = 0; i < NI; ++i) {
				const int b = tf + i * TF, k = b >> log2J, j = b & (J - 1);
#pragma unroll
				for (int m = 0; m < R; ++m)
					g.v[i][m] = lds[RP::pad_off(k * R * J + j, m * J)]; // (one address per item, the rest immediates)
			}
		}
	}

	// slot: the thread's sixteen outputs numbered i * R + c (compile-time after unrolling); in the last pass the thread that
	// holds the two real bins hands bin N/2 over as a seventeenth (slot 16)
	template <class Out>
	static __device__ __forceinline__ void emit(float2* __restrict__ lds, Out& out, bool active, int bin, int j, float2 X, int slot)
	{
		static_assert(!LAST, "the last pass calls the output functor itself");
		(void)out, (void)active, (void)slot;
		lds[RP::pad_off(j, bin * J)] = X;
	}

	template <class Out>
	static __device__ __forceinline__ void compute(int tf, float2* __restrict__ lds, const TW& tw, Out& out, bool active, Regs& g)
	{
		if constexpr (FIRST) {
#pragma unroll
			for (int i = 0; i < NI; ++i) {
				const int j = tf + i * TF;
				float2 y[R / 2 + 1];
				real_dag0<R, ZU, TW::PACKED>(g.x[i], y, tw.p, LOG2N);
				static_assert(!LAST, "a transform has at least two passes");
				emit(lds, out, active, 0, j, make_float2(y[0].x, y[R / 2].x), i * (R / 2)); // the two real bins: one slot
#pragma unroll
				for (int c = 1; c < R / 2; ++c)
					emit(lds, out, active, c, j, y[c], i * (R / 2) + c); // bin c 2^0
			}
		}
		else {
			static_assert(J <= TF, "only a thread's first item can be the one with k == 0");
#pragma unroll
			for (int i = 0; i < NI; ++i) {
				const int b = tf + i * TF, k = b >> log2J, j = b & (J - 1);
				// The item's R results, then ONE sequence of stores for all lanes.  (Written as two branches that each stored their
				// own results, every wavefront with a k == 0 lane -- all of them at nfft <= 2048 -- ran the output functor's code
				// twice: the magnitudes' double-precision square roots, the address arithmetic, the stores.)
				// Where they go: result c < R/2 is bin k1 + c 2^s, result c >= R/2 is bin (R - c) 2^s - k2, with k1 = k2 = k for a
				// general item.  The k == 0 item fits the same two formulas with k1 = 0, k2 = h = 2^(s-1): its sub-DAG at bin 0
				// yields bins c 2^s (c < R/2; the two real ones share c = 0), the one at bin h yields h + c' 2^s = (c' + 1) 2^s - h,
				// the place of result R - 1 - c'.  So the addresses of all lanes are two bases plus compile-time offsets.
				float2 o[R];
				float nyq = 0.0f;
				int k1 = k, k2 = k;
				const bool sp = i == 0 && tf < J; // k == 0: the packed real bins 0 and h of J sub-sequences
				float x0[R], xh[R]; // (the packed real values, before the general stages run over the registers)
#pragma unroll
				for (int m = 0; m < R; ++m) {
					x0[m] = g.v[i][m].x;
					xh[m] = g.v[i][m].y;
				}
				auto general = [&]() {
#ifdef ZEN_RFFT_OLD_BFLY
					if constexpr (PRE)
						butterfly<R, false, false, false>(g.v[i], k, sL, LOG2N, g.tw, PASS, i);
					else
						butterfly<R, false, false, false>(g.v[i], k, sL, LOG2N, tw, PASS, i);
#pragma unroll
					for (int c = 0; c < R; ++c)
						o[c] = c < R / 2 ? g.v[i][c] : make_float2(g.v[i][c].x, -g.v[i][c].y);
#else
					if constexpr (PRE)
						rbutterfly<R>(g.v[i], k, sL, LOG2N, g.tw, PASS, i);
					else
						rbutterfly<R>(g.v[i], k, sL, LOG2N, tw, PASS, i);
#pragma unroll
					for (int c = 0; c < R; ++c)
						o[c] = g.v[i][c]; // (the upper half comes out conjugated)
#endif
				};
				auto special = [&]() {
					float2 y0[R / 2 + 1], yh[R / 2];
					real_dag0<R, false, TW::PACKED>(x0, y0, tw.p, LOG2N);
					real_dagh<R, TW::PACKED>(xh, yh, tw.p, LOG2N);
					o[0] = make_float2(y0[0].x, LAST ? 0.0f : y0[R / 2].x); // one slot of the image; two bins of the spectrum (nyq)
					nyq = y0[R / 2].x;
#pragma unroll
					for (int c = 1; c < R / 2; ++c)
						o[c] = y0[c];
#pragma unroll
					for (int c = 0; c < R / 2; ++c)
						o[R - 1 - c] = yh[c];
					k1 = 0;
					k2 = 1 << (sL - 1);
				};
				if (i != 0) {
					general();
				}
				else if (J >= 64) { // whole wavefronts of k == 0 items: one or the other
					if (sp)
						special();
					else
						general();
				}
				else { // a few lanes of a wavefront: the general stages for everybody (on those lanes: of values nobody uses), then
					general(); // the real sub-DAGs over them
					if (sp)
						special();
				}
				const int a1 = k1 * J + j, a2 = j - k2 * J; // (the image: bin * J + j)
#pragma unroll
				for (int c = 0; c < R; ++c) {
					if constexpr (LAST) { // bin = base + a compile-time offset: the functor forms two addresses per item, not sixteen
						if (active)
							out(c < R / 2 ? k1 : -k2, c < R / 2 ? (c << sL) : ((R - c) << sL), o[c], i * R + c);
					}
					else
						lds[RP::pad_off(c < R / 2 ? a1 : a2, (c < R / 2 ? c : R - c) * (N / R))] = o[c];
				}
				if constexpr (LAST) {
					if (sp && active) // (one lane per frame)
						out(0, N / 2, make_float2(nyq, 0.0f), 16);
				}
			}
		}
	}
};

template <int LOG2N, int PASS, bool ZU, class In, class Out, class TW>
struct RPassRunner {
	using RP = RPlan<LOG2N>;
	static __device__ __forceinline__ void run(int tf, float2* __restrict__ lds, const TW& tw, In& in, Out& out, bool active)
	{
		using PS = RPass<LOG2N, PASS, ZU, TW>;
		typename PS::Regs g;
		PS::load(tf, lds, in, g, tw.p);
		if constexpr (PASS != 0)
			frame_sync<RP::TF>(); // every thread has its inputs in registers: the image may be overwritten
		PS::compute(tf, lds, tw, out, active, g);
		if constexpr (PASS + 1 < RP::P) {
			frame_sync<RP::TF>();
			RPassRunner<LOG2N, PASS + 1, ZU, In, Out, TW>::run(tf, lds, tw, in, out, active);
		}
	}
};

// One N-point transform of a real frame by the N/32 threads that own it; `lds`: the frame's image, RPlan::LDS_FLOAT2 float2.
// All threads of the block call this together; inactive frames pass active = false (out() is not called, in() is).
template <int LOG2N, bool ZU, class In, class Out, class TW = TwGlobal>
__device__ __forceinline__ void rfft_frame(int tf, float2* __restrict__ lds, const TW& tw, In& in, Out& out, bool active)
{
	RPassRunner<LOG2N, 0, ZU, In, Out, TW>::run(tf, lds, tw, in, out, active);
}

} // namespace zfft
