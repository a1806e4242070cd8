// rt_wide.hip -- one launch per hop for the causal median path at the long transforms (hop 2048 / 4096: nfft
// 8192 / 16384, frequency masks of 85 .. 187 taps): HPR::process_next_hop (libzen/hps.cu:429-579) for ONE hop
// per stream.
//
// A single frame of this size is too much for one compute unit: a 16384-point transform keeps the four SIMDs of a
// CU busy for 15 us, the inverse for 20, and one thread's share of a 187-tap median is a 10 us dependent chain --
// the four launches of the general engine add up to 57 us of kernel time per hop (65 us per call; this kernel:
// 39 us, 48 per call).  Here the frame is spread over G = nfft/4096 workgroups that meet at grid barriers, and each
// transform is cut in two steps so that no step needs data from another workgroup:
//
//   the radix-2 decimation-in-time DAG of fft_dev.h, N = M*J with M = 128:
//     step A (stages 1..7):   J independent M-point transforms of the decimated sequences x[j + n*J], j < J.  They
//                             use every J-th entry of the twiddle table: tw[k * N/2^s] = tw[(k * M/2^s) * J].
//     step B (stages 8..log2 N): for each kappa < M the values Y_7[j][kappa], j < J, go through a J-point transform
//                             whose stage-t twiddle of frequency q is tw[(kappa + M*q) * J/2^t]
//                             = tw[(q * J/2^t) * M + kappa * J/2^t]: the index of a plain J-point transform,
//                             scaled and shifted.  It delivers X[kappa + M*q].
//   Both steps run through PassRunner of fft_dev.h with a twiddle source that does this index arithmetic, so
//   every butterfly is the one the single-workgroup kernels (and the oracle's recursion) evaluate: same values.
//
//   phase 1  carries + input tail (a share per workgroup); analysis step A -> exchange buffer   | grid barrier
//   phase 2  analysis step B -> spectrum row (bins 0..N/2), whole magnitude row (hps.cu:492)     | grid barrier
//   phase 3  frequency median (median_big.h): one 16-bin block per thread, bins 0..N/2+15 and the
//            last blocks of the row (the rest is the mirror image: |S| is exactly Hermitian)     | grid barrier
//   per output: mask * spectrum (hps.h:100-140, :58-66), synthesis step A                        | grid barrier
//               synthesis step B, *COLA, overlap-add with the carry -> Y row + finished hop      | arrival count
//               (a full barrier only if another output follows); whoever arrives last publishes the hop's
//               sequence number
//
// XCD-aware launch: workgroups are dealt to the eight XCDs round-robin, each with its own L2.  The grid has 8*G
// workgroups and only every eighth works (the others retire at once), so the G that cooperate share one L2: the
// barrier word and the exchanged data stay there.  Correctness does not depend on the placement: the workgroups
// check where they run (grid_sync below) and release at agent scope unless they all share an XCD.
#include "common.h"
#include "fft_dev.h"
#include "masks.h"
#include "median_big.h"
#include "row_load.h"
#include "rt_fused.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

using zfft::Plan;
using znet::from_key;

constexpr int RSTR = 20; // LDS words per 16-sample block (median_big.hip)
constexpr int WGT = 256; // threads per workgroup
constexpr int XCDS = 8;

template <int LOG2N>
struct WideGeo {
	static constexpr int N = 1 << LOG2N;
	static constexpr int LOG2M = 7, M = 1 << LOG2M;            // step A: M-point transforms
	static constexpr int LOG2J = LOG2N - LOG2M, J = 1 << LOG2J; // step B: J-point transforms
	static constexpr int G = N / 16 / WGT;                     // cooperating workgroups (16 values per thread)
	using PA = Plan<LOG2M>;
	using PB = Plan<LOG2J>;
	static constexpr int FA = WGT / PA::TF, FB = WGT / PB::TF; // frames per workgroup in either step
	static_assert(PA::TF <= 64 && PB::TF <= 64, "frames are synchronised inside their wavefront");
	static constexpr int LDS_FFT = sizeof(float2) * (FA * PA::LDS_FLOAT2 > FB * PB::LDS_FLOAT2 ? FA * PA::LDS_FLOAT2 : FB * PB::LDS_FLOAT2);
};

// Twiddles.  Step A: the M-point transform reads every J-th entry of the table.  Step B: stage t of the J-point
// transform of column kappa reads entry (idx * M) + kappa * J/2^t, idx the index a plain J-point transform would
// use.  Both sets are loaded into registers once, at the start of the call (fft_dev.h TwRegs): the analysis and the
// synthesis use the same ones (conjugated), and no pass waits for a table lookup.
template <int LOG2J>
struct TwStrideMap {
	const float2* __restrict__ p;
	__device__ __forceinline__ float2 operator()(int, int idx) const { return p[idx << LOG2J]; }
};
template <int LOG2M, int LOG2J>
struct TwTwistMap {
	const float2* __restrict__ p;
	int kappa;
	__device__ __forceinline__ float2 operator()(int t, int idx) const { return p[(idx << LOG2M) + (kappa << (LOG2J - t))]; }
};

// ---- grid barrier of the G working workgroups of one stream.  The counter only grows: arrival number `target`
// completes the barrier.  A bounded spin: a workgroup that never arrives (it cannot, with G <= 4 on an idle or
// busy device -- they are always co-resident eventually) must not hang the device; the flag word bar[1] records it.
//
// Visibility.  Agent scope on a multi-XCD device means writing the L2 back (buffer_wbl2: every dirty line of the
// call so far) -- 1 to 2 us per barrier, more than the barrier itself.  Workgroups on ONE XCD share that L2: their
// stores are there once vmcnt says so (the vector L1 is write-through), and a reader only has to drop its L1
// (buffer_inv).  So the workgroups vote: at the start of the call each adds 1 to the nibble of the XCD it runs on
// (XCC_ID register) in the call's vote word; before the first barrier each waits until the nibbles add up to G
// (long since true) and all see the same word: if one nibble holds all G votes, the call's barriers release at
// workgroup scope only (`light`), else at agent scope.  Two vote words alternate from call to call; workgroup 0
// clears the next call's word, which nobody touches during this call.
__device__ __forceinline__ unsigned xcc_id()
{
	unsigned v;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
	return v & 15u;
}
// what a workgroup adds to the vote word: one vote in the nibble of its XCD -- or, should the register ever name an
// XCD beyond the eight nibbles, a vote in two of them, which can only turn the light path off
__device__ __forceinline__ unsigned xcc_vote()
{
	const unsigned x = xcc_id();
	return x < 8 ? 1u << (4 * x) : 0x10000001u;
}

template <int G>
__device__ __forceinline__ bool placement_vote_result(const unsigned* vote)
{
	__shared__ unsigned vote_seen;
	if (threadIdx.x == 0) {
		unsigned v = 0;
		for (int spins = 0; spins < (1 << 16); ++spins) {
			v = __hip_atomic_load(vote, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			unsigned sum = 0;
			for (int x = 0; x < 8; ++x)
				sum += (v >> (4 * x)) & 15u;
			if (sum >= (unsigned)G)
				break;
			__builtin_amdgcn_s_sleep(1);
		}
		vote_seen = v;
	}
	__syncthreads();
	const unsigned v = vote_seen;
	bool one = false;
	for (int x = 0; x < 8; ++x)
		one = one || v == ((unsigned)G << (4 * x));
	return one;
}

// What every wave does before it arrives at a barrier.  Light: the stores only have to be IN the shared L2, which
// they are once the memory counter has drained -- a workgroup-scope release fence compiles to nothing on gfx950 (no
// s_waitcnt at all: the arrival atomic could overtake stores still in flight), so the wait is written out.
// Agent scope: the fence emits buffer_wbl2 + s_waitcnt vmcnt(0).
__device__ __forceinline__ void release_stores(bool light)
{
	if (light)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	else
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

__device__ __forceinline__ void grid_sync(unsigned* bar, unsigned target, bool light, unsigned* fail)
{
	release_stores(light);
	__syncthreads();
	if (threadIdx.x == 0) {
		// the last to arrive (the one everybody waits for) knows it from the value the add returns: no polling trip
		const unsigned before = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		int spins = 0;
		while ((int)(before + 1 - target) < 0
		       && (int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
			__builtin_amdgcn_s_sleep(1);
			if (++spins > (1 << 22)) {
				__hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (fail) { // host-mapped: every later copy_* of the engine reports the hop as failed (hpr.hip)
					__hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					__threadfence_system();
				}
				break;
			}
		}
	}
	__syncthreads();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// ---- functors of the four transform steps
template <int LOG2J>
struct FwdAIn { // x[j + n*J] of the windowed, zero-padded frame (only n < M/2 is asked for: ZU)
	const float* prev;
	const float* cur;
	const float* window;
	int hop, j;
	__device__ __forceinline__ float2 operator()(int n, int) const
	{
		const int idx = j + (n << LOG2J);
		ZH_CHK(idx < hop ? prev + idx : cur + (idx - hop), 1);
		ZH_CHK(window + idx, 1);
		const float x = idx < hop ? prev[idx] : cur[idx - hop];
		return make_float2(x * window[idx], 0.0f); // window_functor hps.h:24-33
	}
};
template <int LOG2J>
struct XchOut { // Y_7[j][k] -> exchange buffer, laid out [k][j]: step B reads a column contiguously
	float2* T;
	int j;
	__device__ __forceinline__ void operator()(int k, float2 X, bool, int) const
	{
		ZH_CHK(T + ((k << LOG2J) + j), 1);
		T[(k << LOG2J) + j] = X;
	}
};
template <int LOG2J>
struct XchIn {
	const float2* T;
	int kappa;
	__device__ __forceinline__ float2 operator()(int jj, int) const
	{
		ZH_CHK(T + ((kappa << LOG2J) + jj), 1);
		return T[(kappa << LOG2J) + jj];
	}
};
template <int LOG2M>
struct FwdBOut { // X[kappa + M*q]: spectrum ring row (bins 0..N/2) and the whole magnitude row (StftOut of stft.hip)
	float2* S;
	float* mag;
	int kappa, n;
	__device__ __forceinline__ void operator()(int q, float2 X, bool, int) const
	{
		const int k = kappa + (q << LOG2M);
		if (k <= (n >> 1)) {
			ZH_CHK(S + k, 1);
			ZH_CHK(mag + k, 1);
			S[k] = X;
			const float m = zfft::cabs_exact(X.x, X.y); // complex_abs_functor hps.h:82-89
			mag[k] = m;
			if (k != 0 && k != (n >> 1)) {
				ZH_CHK(mag + (n - k), 1);
				mag[n - k] = m; // |S[n-k]| == |S[k]| bit for bit
			}
		}
	}
};
template <int LOG2J>
struct InvAIn { // (S * mask)[j + n*J] (IstftIn of istft.hip): the upper half of the spectrum is the conjugate mirror image
	const float2* S;
	const float* H;
	const float* P;
	MaskCfg cfg;
	int which, n, p_mid, j;
	HardThr thr; // hard masks by exact comparison instead of the division (masks.h)
	__device__ __forceinline__ float2 operator()(int nn, int) const
	{
		const int idx = j + (nn << LOG2J);
		const bool mirror = idx > (n >> 1);
		const int lo = mirror ? n - idx : idx;
		ZH_CHK(S + lo, 1);
		float2 z = S[lo];
		if (mirror)
			z.y = -z.y;
		const int pi = (mirror && idx >= n - p_mid) ? idx : lo;
		ZH_CHK(H + lo, 1);
		ZH_CHK(P + pi, 1);
		const float m = mask_value_thr(which, H[lo], P[pi], cfg, thr);
		return make_float2(z.x * m, z.y * m); // apply_mask_functor hps.h:58-66
	}
};
template <int LOG2M>
struct InvBOut { // x[kappa + M*q], q < J/2 (HALF_OUT): the nwin real outputs that are used
	float* Y;
	float cola;
	float* ready;
	const float* carry;
	int hop, kappa;
	bool through; // the finished hop goes to host-mapped memory with system-scope (write-through) stores: rt_fused.hip publish_ready<LIGHT>
	__device__ __forceinline__ void operator()(int q, float2 x, bool, int) const
	{
		const int idx = kappa + (q << LOG2M);
		const float y = x.x * cola; // overlap_add_functor hps.h:68-80
		ZH_CHK(Y + idx, 1);
		Y[idx] = y;
		if (idx < hop) {
			ZH_CHK(ready + idx, 1);
			ZH_CHK(carry + idx, 1);
			const float v = carry[idx] + y; // hps.cu:526-528 + :341-363
			if (through)
				__hip_atomic_store(ready + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			else
				ready[idx] = v;
		}
	}
};

// What changes from hop to hop of a stream (HopOfArgs / HopVar of rt_fused.h), and what a resident launch carries from one hop
// to the next: the arrival count of the stream's barrier word and the outcome of the placement vote (the workgroups do not
// move while they stay).  (The twiddles of both steps are loaded again every hop: kept across the loop -- with every address the
// compiler then hoists out of it -- the kernel needs 540 bytes of scratch per lane on top of 256 + 256 registers.)
struct WideHop {
	unsigned seq;
	long long row0;
	const float* tail_prev;
	float* tail_next;
	int prev_frames;
};
template <int LOG2N>
struct WideKeep {
	unsigned arrivals;
	bool light, voted;
};

// one hop by the G cooperating workgroups (the per-launch kernel runs it once; the resident one once per hop it is handed)
template <int LOG2N, int W>
__device__ __forceinline__ void rt_wide_body(const RtFusedArgs& a, const WideHop& hv, WideKeep<LOG2N>& keep, const int g, const int s,
                                             const int t, float2* lds)
{
	using GEO = WideGeo<LOG2N>;
	using PA = typename GEO::PA;
	using PB = typename GEO::PB;
	using GM = zbig::Geo<W>;
	constexpr int N = GEO::N, G = GEO::G, LOG2M = GEO::LOG2M, LOG2J = GEO::LOG2J;
	const int hop = a.hop;
	const int u = g * WGT + t; // thread of the frame
	unsigned* bar = a.bar + 4 * s;
	unsigned& arrivals = keep.arrivals;
	// diagnostic (tools/rt_latency.cpp --stamps): 100 MHz stamps of workgroup 0 around every phase and barrier
	unsigned long long stamps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	auto stamp = [&](int k) { // (k is a literal everywhere: the array stays in registers)
		if (a.stamps && k < 12)
			stamps[k] = __builtin_amdgcn_s_memrealtime();
	};
	unsigned* vote = bar + 2 + (a.bar_parity & 1);
	if (t == 0 && !keep.voted) // the placement vote (grid_sync): counted long before the first barrier asks for it
		__hip_atomic_fetch_add(vote, xcc_vote(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	bool& light = keep.light;
	auto sync = [&](int k) { // stamps k (arrival) and k + 1 (release)
		stamp(k);
		arrivals += G;
		if (k == 1 && !keep.voted) { // (a resident launch: once, before its first barrier)
			light = placement_vote_result<G>(vote) && a.diag != 4; // ("rt_fused_diag" 4: agent-scope barriers, for timing)
			if (g == 0 && t == 0)
				__hip_atomic_store(bar + 2 + ((a.bar_parity & 1) ^ 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			keep.voted = true;
		}
		grid_sync(bar, arrivals, light, a.wide_fail);
		stamp(k + 1);
	};
	stamp(0);

	const long long row = (hv.row0 % a.ring_rows) + (long long)s * a.ring_rows;
	float2* Srow = a.S + row * a.s_stride;
	float* mrow = a.mag + row * N;
	float* prow = a.P + (long long)s * a.p_stream_stride;
	float2* T = a.xch + (long long)s * N;

	zfft::TwRegs<LOG2M> twA;
	zfft::TwRegs<LOG2J, false> twB; // (not the J-point transform's own table: no trivial twiddles)
	// Lanes of step A: neighbouring lanes hold neighbouring sequences j (same position inside the sequence), so the
	// strided elements x[j + n*J] they ask for are neighbours in memory; a frame still lives inside one wavefront.
	// Step B: neighbouring lanes are neighbouring threads of one column, whose exchange-buffer entries are contiguous.
	constexpr int FWA = 64 / PA::TF; // step-A frames per wavefront
	const int tfA = (t & 63) / FWA, fA = (t >> 6) * FWA + (t & 63) % FWA, jA = g * GEO::FA + fA;
	twA.fill_with(tfA, TwStrideMap<LOG2J>{a.tw});
	twB.fill_with(u % PB::TF, TwTwistMap<LOG2M, LOG2J>{a.tw, u / PB::TF});
	// ---- phase 1: carries and input tail, a share for every workgroup (loads first, then stores); analysis step A
	{
		constexpr int CP = N / 4 / (G * WGT); // samples of a hop per thread
		const float* cur = a.in + (long long)s * a.in_stride;
		float v[CP];
#pragma unroll
		for (int i = 0; i < CP; ++i)
			v[i] = cur[u + i * (G * WGT)];
#pragma unroll
		for (int i = 0; i < CP; ++i)
			hv.tail_next[(long long)s * hop + u + i * (G * WGT)] = v[i];
		if (hv.prev_frames > 0) {
			for (int o = 0; o < 3; ++o) {
				if (!a.carry[o])
					continue;
				const float* y = a.Y[o] + (long long)s * a.y_stream_stride + (long long)(hv.prev_frames - 1) * (2 * hop) + hop;
#pragma unroll
				for (int i = 0; i < CP; ++i)
					v[i] = y[u + i * (G * WGT)];
#pragma unroll
				for (int i = 0; i < CP; ++i)
					a.carry[o][(long long)s * hop + u + i * (G * WGT)] = v[i];
			}
		}
	}
	{
		const int j = jA, tf = tfA;
		FwdAIn<LOG2J> in{hv.tail_prev + (long long)s * hop, a.in + (long long)s * a.in_stride, a.window, hop, j};
		XchOut<LOG2J> out{T, j};
		zfft::PassRunner<LOG2M, 0, false, true, false, FwdAIn<LOG2J>, XchOut<LOG2J>, false, zfft::TwRegs<LOG2M>>::run(
		    tf, lds + fA * PA::LDS_FLOAT2, twA, in, out, true);
	}
	sync(1);
	// ---- phase 2: analysis step B
	{
		const int kappa = u / PB::TF, tf = u % PB::TF;
		XchIn<LOG2J> in{T, kappa};
		FwdBOut<LOG2M> out{Srow, mrow, kappa, N};
		zfft::PassRunner<LOG2J, 0, false, false, false, XchIn<LOG2J>, FwdBOut<LOG2M>, false, zfft::TwRegs<LOG2J, false>>::run(
		    tf, lds + (t / PB::TF) * PB::LDS_FLOAT2, twB, in, out, true);
	}
	sync(3);
	// ---- phase 3: frequency median of the magnitude row (hps.cu:496), one 16-bin block per thread.
	// Piece A: blocks g*256 .. of 0..MAIN (bins 0..N/2+15); piece B: the last NT blocks of the row, whose replicate
	// border is not the mirror image of the first ones'  -- they follow piece A's last thread.
	{
		constexpr int NBLK = N / 16, MAIN = NBLK / 2, NT = (GM::m + 15) / 16;
		constexpr int HALO = GM::a + GM::b + 3; // raw blocks around a piece: a+2 left, b+1 right
		constexpr int NRAW_A = WGT + HALO, NSORT_A = WGT + GM::NB - 1;
		int* raw = reinterpret_cast<int*>(lds);
		int* srt = raw + (NRAW_A + NT + HALO) * RSTR;
		// The MAIN + 1 blocks of piece A in equal shares (round 5; they used to go out 256 at a time: at nfft 16384 two workgroups
		// staged and sorted 256 blocks each, the third one 13, the fourth none -- the selection itself is one block per thread
		// either way, the staging and sorting loops in front of it are half as long now); the tail with the last share.
		constexpr int PER = (MAIN + 1 + G - 1) / G;
		static_assert(PER + NT <= WGT, "a workgroup's blocks: one per thread");
		const int bA = g * PER;
		const int nA = MAIN + 1 - bA < 0 ? 0 : (MAIN + 1 - bA > PER ? PER : MAIN + 1 - bA);
		const bool ownsB = g == G - 1;
		const int nB = ownsB ? NT : 0;
		const int rawA = nA ? nA + HALO : 0, rawB = nB ? nB + HALO : 0;     // raw chunks of either piece
		const int srtA = nA ? nA + GM::NB - 1 : 0, srtB = nB ? nB + GM::NB - 1 : 0;
		for (int vi = t; vi < (rawA + rawB) * 4; vi += WGT) { // chunk c of a piece = block (first - (a+2)) + c
			const bool pb = vi >= rawA * 4;
			const int v = pb ? vi - rawA * 4 : vi;
			const int vc = 16 * ((pb ? NBLK - NT : bA) - (GM::a + 2)) + 4 * v;
			const int chunk = pb ? NRAW_A + (v >> 2) : (v >> 2);
			*reinterpret_cast<int4*>(&raw[chunk * RSTR + 4 * (v & 3)]) = row_vec_keys<true>(mrow, vc, N, 0);
		}
		__syncthreads();
		for (int e = t; e < srtA + srtB; e += WGT) { // sorted entry e of a piece = raw chunk e + 2 of that piece
			const bool pb = e >= srtA;
			const int le = pb ? e - srtA : e;
			int v[16];
			znet::lds_load<16>(&raw[((pb ? NRAW_A : 0) + le + 2) * RSTR], v);
			znet::sort_net<16>(v);
			znet::lds_store<16>(&srt[((pb ? NSORT_A : 0) + le) * RSTR], v);
		}
		__syncthreads();
		const bool inA = t < nA, inB = !inA && t - nA < nB;
		if (inA || inB) {
			const int loc = inA ? t : t - nA;
			struct Loader {
				const int* srt_t;
				const int* raw_t;
				__device__ __forceinline__ void sorted(int i, int* v) const { znet::lds_load<16>(srt_t + i * RSTR, v); }
				__device__ __forceinline__ void rawl(int j, int* v) const { znet::lds_load<16>(raw_t + j * RSTR, v); }
				__device__ __forceinline__ void rawr(int j, int* v) const { znet::lds_load<16>(raw_t + (GM::a + GM::b + 2 + j) * RSTR, v); }
			} ld{&srt[((inA ? 0 : NSORT_A) + loc) * RSTR], &raw[((inA ? 0 : NRAW_A) + loc) * RSTR]};
			int out[16];
			zbig::medians_big<W>(ld, out);
			float* d = prow + 16 * ((inA ? bA : NBLK - NT) + loc);
#pragma unroll
			for (int v = 0; v < 4; ++v)
				*reinterpret_cast<float4*>(d + 4 * v) = make_float4(from_key<true>(out[4 * v]), from_key<true>(out[4 * v + 1]),
				                                                    from_key<true>(out[4 * v + 2]), from_key<true>(out[4 * v + 3]));
		}
	}
	sync(5);
	// ---- synthesis per enabled output (hps.cu:498-579; H = |S| of the same row: causal)
	for (int oi = 0; oi < a.n_out; ++oi) {
		const int which = a.out_id[oi];
		{
			const int j = jA, tf = tfA;
			InvAIn<LOG2J> in{Srow, mrow, prow, MaskCfg{a.beta, a.beta_h, a.soft, a.power, 0, a.out_h, a.out_p}, which, N, GM::m, j,
			                  HardThr{a.thr, a.thr_h, a.thr_inclusive, a.thr_h_inclusive}};
			XchOut<LOG2J> out{T, j};
			zfft::PassRunner<LOG2M, 0, true, false, false, InvAIn<LOG2J>, XchOut<LOG2J>, false, zfft::TwRegs<LOG2M>>::run(
			    tf, lds + fA * PA::LDS_FLOAT2, twA, in, out, true);
		}
		if (oi == 0)
			sync(7);
		else
			sync(12);
		float* ready = a.ready[which] + (long long)s * hop;
		{
			const int kappa = u / PB::TF, tf = u % PB::TF;
			XchIn<LOG2J> in{T, kappa};
			InvBOut<LOG2M> out{a.Y[which] + (long long)s * a.y_stream_stride, a.cola, ready, a.carry[which] + (long long)s * hop, hop, kappa,
			                   a.publish_seq == 1};
			zfft::PassRunner<LOG2J, 0, true, false, true, XchIn<LOG2J>, InvBOut<LOG2M>, false, zfft::TwRegs<LOG2J, false>>::run(
			    tf, lds + (t / PB::TF) * PB::LDS_FLOAT2, twB, in, out, true);
		}
		// The finished hop is host-mapped: visible there before the sequence word.  publish_seq == 2 (default): system-scope fence
		// + release store.  publish_seq == 1 (opt-in light form): the samples went out as write-through stores and are on their
		// way once the memory counter has counted them off (release_stores waits for it in front of every arrival) -- no
		// write-back of the XCD's whole L2, as in the one-workgroup kernels (rt_fused.hip publish_ready<LIGHT>).  The light form
		// only where the cooperating workgroups share an XCD (`light`: the placement vote): the flag is stored by whoever
		// arrives last, and from another XCD its order behind the other workgroups' samples would rest on nothing.
		const int pub = (a.publish_seq == 1 && !light) ? 2 : a.publish_seq;
		if (pub == 2)
			__threadfence_system();
		else if (pub)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (oi + 1 < a.n_out) { // the exchange buffer is needed again: a full barrier
			if (oi == 0)
				sync(9);
			else
				sync(12);
			if (pub == 2 && g == 0 && t == 0)
				__hip_atomic_store(reinterpret_cast<unsigned*>(ready + hop), hv.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			else if (pub && g == 0 && t == 0)
				__hip_atomic_store(reinterpret_cast<unsigned*>(ready + hop), hv.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		else { // the call's last barrier: nobody waits, whoever arrives last publishes the hop
			stamp(oi == 0 ? 9 : 12);
			arrivals += G;
			release_stores(light);
			__syncthreads();
			if (t == 0) {
				const unsigned before = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
				if (before + 1 == arrivals && pub == 2)
					__hip_atomic_store(reinterpret_cast<unsigned*>(ready + hop), hv.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
				else if (before + 1 == arrivals && pub) // (every workgroup drained its write-through stores before it arrived)
					__hip_atomic_store(reinterpret_cast<unsigned*>(ready + hop), hv.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
			stamp(oi == 0 ? 10 : 12);
		}
	}
	if (a.stamps && g == 0 && s == 0 && t == 0) {
#pragma unroll
		for (int k = 0; k < 12; ++k)
			a.stamps[k] = stamps[k];
	}
	if (a.stamps && s == 0 && t == 0) // where the cooperating workgroups ran, and whether the barriers were light
		a.stamps[12 + g] = xcc_id() | (light ? 16u : 0u);
}

template <int LOG2N, int W>
__global__ __launch_bounds__(WGT) __attribute__((amdgpu_waves_per_eu(1, 1))) void rt_wide_kernel(RtFusedArgs a)
{
	if (blockIdx.x % XCDS != 0)
		return; // see "XCD-aware launch" above
	extern __shared__ float2 lds[];
	WideKeep<LOG2N> keep;
	keep.arrivals = a.bar_base;
	keep.light = false;
	keep.voted = false;
	const WideHop hv{a.seq, a.row0, a.tail_prev, a.tail_next, a.prev_frames};
	rt_wide_body<LOG2N, W>(a, hv, keep, (int)(blockIdx.x / XCDS), (int)blockIdx.y, (int)threadIdx.x, lds);
}

// The same as a RESIDENT kernel for the per-hop API (zen_hip_hpr_set_resident; rt_fused.hip rt_fused_resident_kernel is the
// one-workgroup form): the G cooperating workgroups stay on their CUs between the hops of the stream.  Workgroup 0 alone
// watches the host's mailbox (resident_next_hop: a hop, the stop word, or idle for too long) and hands its decision to the
// others through a word of device memory -- they must agree on every hop and on when to leave, or one of them would wait at
// a grid barrier for a workgroup that has gone.  The word: (number of the decision << 33) | (leave << 32) | sequence number;
// a follower waits for decision k + 1, then takes the system-scope acquire fence the leader took (the host wrote the hop's
// samples before the mailbox word).  Carried from hop to hop: the barrier's arrival count and the placement vote's outcome
// (nobody moves).  One stream (the launcher checks).
template <int LOG2N, int W>
__global__ __launch_bounds__(WGT) __attribute__((amdgpu_waves_per_eu(1, 1))) void rt_wide_resident_kernel(RtFusedArgs a0, const ResidentCtl* ctl,
                                                                                                      ResidentOut* ro, unsigned long long* go,
                                                                                                      unsigned seq_start,
                                                                                                      unsigned long long idle_ticks, unsigned max_hops)
{
	if (blockIdx.x % XCDS != 0)
		return;
	extern __shared__ float2 lds[];
	__shared__ unsigned s_cmd[2];
	const int g = blockIdx.x / XCDS, t = threadIdx.x;
	WideKeep<LOG2N> keep;
	keep.arrivals = a0.bar_base;
	keep.light = false;
	keep.voted = false;
	unsigned last = seq_start, k = 0;
	for (;;) {
		unsigned sq = last;
		bool have;
		if (g == 0) {
			have = resident_next_hop(ctl, last, idle_ticks, k >= max_hops, s_cmd, &sq);
			if (t == 0) // (release: nothing of this workgroup's last hop is still in flight -- its last barrier drained the counter)
				__hip_atomic_store(go, ((unsigned long long)(k + 1) << 33) | ((unsigned long long)(have ? 0 : 1) << 32) | sq, __ATOMIC_RELEASE,
				                   __HIP_MEMORY_SCOPE_AGENT);
		}
		else {
			if (t == 0) {
				unsigned long long w;
				for (;;) { // (the leader decides within idle_ticks: a bounded wait)
					w = __hip_atomic_load(go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
					if ((unsigned)(w >> 33) == k + 1)
						break;
					__builtin_amdgcn_s_sleep(1);
				}
				s_cmd[0] = (unsigned)(w >> 32) & 1u;
				s_cmd[1] = (unsigned)w;
			}
			__syncthreads();
			have = s_cmd[0] == 0u;
			sq = s_cmd[1];
			if (have)
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
		}
		if (!have)
			break;
		WideHop hv;
		hv.seq = sq;
		hv.row0 = a0.row0 + k;
		hv.tail_prev = (k & 1u) ? a0.tail_next : a0.tail_prev; // the two input-tail buffers flip with every call
		hv.tail_next = (k & 1u) ? const_cast<float*>(a0.tail_prev) : a0.tail_next;
		hv.prev_frames = k > 0u ? 1 : a0.prev_frames;
		int t_o = t, g_o = g; // (opaque per hop: otherwise every address of the body -- all functions of the thread's place alone --
		asm volatile("" : "+v"(t_o)); // is hoisted out of the loop and kept in registers the kernel does not have)
		asm volatile("" : "+s"(g_o));
		rt_wide_body<LOG2N, W>(a0, hv, keep, g_o, 0, t_o, lds);
		__syncthreads(); // (s_cmd is rewritten next)
		last = sq;
		++k;
	}
	if (g == 0)
		resident_leave(ro, last, k);
}

template <int LOG2N, int W>
int launch_wide_t(const RtFusedArgs& a, hipStream_t stream)
{
	using GEO = WideGeo<LOG2N>;
	using GM = zbig::Geo<W>;
	constexpr int NT = (GM::m + 15) / 16, HALO = GM::a + GM::b + 3;
	constexpr size_t lds_med = sizeof(int) * RSTR * (size_t)((WGT + HALO + NT + HALO) + (WGT + GM::NB - 1 + NT + GM::NB - 1));
	constexpr size_t lds = lds_med > (size_t)GEO::LDS_FFT ? lds_med : (size_t)GEO::LDS_FFT;
	auto kern = rt_wide_kernel<LOG2N, W>;
	if (lds > 64 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(kern, dim3(XCDS * GEO::G, (unsigned)a.n_streams), dim3(WGT), lds, stream, a);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

template <int LOG2N, int W>
int launch_wide_res_t(const RtFusedArgs& a, const ResidentCtl* ctl, ResidentOut* ro, unsigned long long* go, unsigned seq_start,
                      unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream)
{
	using GEO = WideGeo<LOG2N>;
	using GM = zbig::Geo<W>;
	constexpr int NT = (GM::m + 15) / 16, HALO = GM::a + GM::b + 3;
	constexpr size_t lds_med = sizeof(int) * RSTR * (size_t)((WGT + HALO + NT + HALO) + (WGT + GM::NB - 1 + NT + GM::NB - 1));
	constexpr size_t lds = lds_med > (size_t)GEO::LDS_FFT ? lds_med : (size_t)GEO::LDS_FFT;
	auto kern = rt_wide_resident_kernel<LOG2N, W>;
	if (lds > 64 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(kern, dim3(XCDS * GEO::G, 1), dim3(WGT), lds, stream, a, ctl, ro, go, seq_start, idle_ticks, max_hops);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace

int launch_rt_wide_resident(int log2n, int freq_len, const RtFusedArgs& a, const ResidentCtl* ctl, ResidentOut* ro, unsigned long long* go,
                            unsigned seq_start, unsigned long long idle_ticks, unsigned max_hops, hipStream_t stream)
{
	if (a.n_out != 1 || a.n_frames != 1 || a.n_streams != 1 || !a.publish_seq || a.stamps)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "resident wide kernel: one stream, one output, single hops, host-mapped hop buffer");
#define ZH_WR(L, W_) return launch_wide_res_t<L, W_>(a, ctl, ro, go, seq_start, idle_ticks, max_hops, stream)
	switch (log2n * 1000 + freq_len) {
	case 13085: ZH_WR(13, 85);
	case 13093: ZH_WR(13, 93);
	case 13129: ZH_WR(13, 129);
	case 13171: ZH_WR(13, 171);
	case 13187: ZH_WR(13, 187);
	case 14085: ZH_WR(14, 85);
	case 14093: ZH_WR(14, 93);
	case 14129: ZH_WR(14, 129);
	case 14171: ZH_WR(14, 171);
	case 14187: ZH_WR(14, 187);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "no resident wide kernel for nfft 2^%d with a %d-tap mask", log2n, freq_len);
	}
#undef ZH_WR
}

// hop 2048 and 4096 at 44.1 / 48 kHz (l_perc = 93 / 85 and 187 / 171 taps), and the other sample rates whose masks
// the block-merge median covers: hop 2048 at 22.05 / 24 / 32 kHz (187 / 171 / 129), hop 4096 at 88.2 / 96 / 64 kHz
// (93 / 85 / 129)
bool rt_wide_available(int log2n, int freq_len)
{
	switch (log2n * 1000 + freq_len) {
	case 13085: case 13093: case 13129: case 13171: case 13187:
	case 14085: case 14093: case 14129: case 14171: case 14187: return true;
	default: return false;
	}
}

// grid barriers of one single-output call (the engine advances its arrival count by this much per call)
unsigned rt_wide_arrivals(int log2n, int n_out)
{
	const unsigned G = log2n == 13 ? WideGeo<13>::G : WideGeo<14>::G;
	return G * (unsigned)(3 + 2 * n_out);
}

int launch_rt_wide(int log2n, int freq_len, const RtFusedArgs& a, hipStream_t stream)
{
	switch (log2n * 1000 + freq_len) {
	case 13085: return launch_wide_t<13, 85>(a, stream);
	case 13093: return launch_wide_t<13, 93>(a, stream);
	case 13129: return launch_wide_t<13, 129>(a, stream);
	case 13171: return launch_wide_t<13, 171>(a, stream);
	case 13187: return launch_wide_t<13, 187>(a, stream);
	case 14085: return launch_wide_t<14, 85>(a, stream);
	case 14093: return launch_wide_t<14, 93>(a, stream);
	case 14129: return launch_wide_t<14, 129>(a, stream);
	case 14171: return launch_wide_t<14, 171>(a, stream);
	case 14187: return launch_wide_t<14, 187>(a, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "no wide single-hop kernel for nfft 2^%d with a %d-tap mask", log2n, freq_len);
	}
}

} // namespace zen_hip_impl
