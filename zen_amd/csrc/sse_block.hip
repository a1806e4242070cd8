// sse_block.hip -- blocks of frames on the SSE path (HPR<GPU>::apply_sse_filter, libzen/hps.cu:582-652; BASELINE
// configs[4]) in TWO launches instead of four, with no H / P / |S|^2 rows in HBM:
//
//   stft_kernel (stft.hip)  : frames -> spectrum ring rows + magnitude ring rows, as on every path
//   sse_synth_kernel (here) : per consumed frame, by the TF = nfft/16 threads that will transform it:
//       1/|S|^2 of the frame's own row           -> an LDS row with the replicate border of box.h:217-288 (ippBorderRepl)
//       time box: the rows around it in the ring -> sum of 1/|S|^2 in ascending frame order (a tap past the consumed row
//                                                   of a causal engine is that row again, hps.h:265-268), / len,
//                                                   H = (l_harm + 1) / mean                          (hps.cu:596, :602-604)
//       frequency box over the LDS row           -> ascending bin order, / len, P = (l_perc + 1) / mean  (hps.cu:597-601)
//       Wiener masks (sse_mask_functor hps.h:132-140) of every enabled output, S * mask, inverse FFT, * COLA -> Y rows
//
// The general engine ran stft + box_time_kernel + box_freq_kernel + istft_kernel: 1/|S|^2 twice per sample, H and P rows
// (whole rows: the box sum is not mirror symmetric) written and read back, |S| read three times.  Here the magnitude
// rows are read once per tap of the time box (neighbouring frames share them through L2), nothing else moves.
// Same arithmetic, operation for operation, as box.hip + istft.hip MODE 0 (sum of the taps in ascending order, then
// / len, then (1/x) * factor): bit-identical outputs, interchangeable call by call ("no_sse_block" selects the old path).
//
// The box sums of a frame live in the frame's FFT image (free until the first inverse pass stores into it): the
// 1/|S|^2 row with its borders, behind it the H row (bins 0..nfft/2: H is mirror symmetric, P is not).  The masks of a
// thread's sixteen bins are in registers before the first transform starts, next to the spectrum.
#include "common.h"
#include "exact_div.h"
#include "fft_dev.h"
#include "fft_launch.h"
#include "masks.h"
#include "sse_block.h"

#include <cfloat>

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

using zfft::Plan;

constexpr int HALO = 128; // floats of replicate border on either side of the 1/|S|^2 row (box <= 255 taps)

struct SynthIn {
	const float2* z; // the thread's sixteen spectrum values (mirrored bins already conjugated)
	const float* m;  // ... and their masks for this output
	__device__ __forceinline__ float2 operator()(int, int slot) const
	{
		return make_float2(z[slot].x * m[slot], z[slot].y * m[slot]); // apply_mask_functor hps.h:58-66
	}
};

struct SynthOut {
	float* Y;
	float cola;
	__device__ __forceinline__ void operator()(int idx, float2 x, bool, int) const
	{
		ZH_CHK(Y + idx, 1);
		Y[idx] = x.x * cola; // the product of overlap_add_functor hps.h:68-80; the sum is in finalize_kernel
	}
};

template <int LOG2N>
// (four workgroups per CU up to nfft 2048 -- 128 registers hold the spectrum, the masks and a transform pass without
// scratch: 0.541 -> 0.529 ms per 51 680 frames at nfft 2048; at nfft 4096 that would spill 36 bytes per lane: three there)
__global__ __launch_bounds__(Plan<LOG2N>::THREADS, LOG2N <= 11 ? 4 : 3) void sse_synth_kernel(SseBlockArgs b)
{
	using PL = Plan<LOG2N>;
	constexpr int N = PL::N, TF = PL::TF, NLO = 9; // slots 0..7 of a thread are bins below nfft/2; slot 8 of thread 0 is bin nfft/2
	static_assert(N + 2 * HALO + N / 2 + 1 <= 2 * PL::LDS_FLOAT2, "the box rows must fit the frame image");
	extern __shared__ float2 lds[];
	const IstftArgs& a = b.ia;
	const int tid = threadIdx.x, s = blockIdx.z;
	const int slot = tid / TF, tf = tid - slot * TF;
	const int f_ = blockIdx.x * PL::FRAMES_PER_BLOCK + slot;
	const bool active = f_ < a.n_frames;
	const int f = active ? f_ : a.n_frames - 1; // (an inactive slot works on the last frame again and stores nothing)
	float2* img = lds + slot * PL::LDS_FLOAT2;
	float* pre = reinterpret_cast<float*>(img); // pre[HALO + k] = (1 / (|S[k]| * |S[k]|)) * 1   (hps.h:91-98, :45-56)
	float* Hrow = pre + N + 2 * HALO;           // bins 0 .. nfft/2
	const long long ar = a.crow0 + f;           // absolute ring row of the consumed frame
	const long long ring_base = (long long)s * a.ring_rows;
	const int base_slot = (int)(ar % a.ring_rows);
	const int mid_t = b.len_t >> 1, mid_f = b.len_f >> 1;

	// ---- the frame's spectrum: bins tf + i*TF, the upper half as the conjugate of its mirror image (stft.h)
	float2 z[16];
	{
		const float2* S = a.S + (ring_base + base_slot) * a.s_stride;
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int idx = tf + i * TF;
			ZH_CHK(S + (idx > N / 2 ? N - idx : idx), 1);
			z[i] = S[idx > N / 2 ? N - idx : idx];
		}
	}
	// the time box's rows, four per turn: their loads (tap_load), then their sums (tap_sum)
	auto tap_load = [&](int j0, float (&m)[4][NLO], int (&d)[4]) {
#pragma unroll
		for (int jj = 0; jj < 4; ++jj) {
			const int j = j0 + jj < b.len_t ? j0 + jj : b.len_t - 1;
			long long r = ar - mid_t + j;
			r = r < b.clamp_lo ? b.clamp_lo : (r > b.clamp_hi ? b.clamp_hi : r);
			if (b.causal_self && r > ar)
				r = ar;
			d[jj] = (int)(r - ar); // |d| < ring_rows: the ring slot without a 64-bit remainder per row
			int rs = base_slot + d[jj];
			rs = rs < 0 ? rs + (int)a.ring_rows : (rs >= (int)a.ring_rows ? rs - (int)a.ring_rows : rs);
			const float* mrow = b.mag + (ring_base + rs) * N;
			if (d[jj] != 0) { // (frame-uniform.  The consumed row itself -- every tap past it of a causal engine -- is at hand: `own`)
#pragma unroll
				for (int i = 0; i < NLO; ++i) {
					ZH_CHK(mrow + (i < 8 ? tf + i * TF : N / 2), 1);
					m[jj][i] = mrow[i < 8 ? tf + i * TF : N / 2];
				}
			}
			else {
#pragma unroll
				for (int i = 0; i < NLO; ++i)
					m[jj][i] = 0.0f;
			}
		}
	};
	float m0[4][NLO]; // the first four rows of the time box: in flight together with the frame's own rows (one round trip
	int d0[4];        // to memory less per frame: 0.522 -> 0.511 ms per 51 680 frames)
	tap_load(0, m0, d0);
	// ---- 1/|S|^2 of the frame's own row (lower half computed, both halves stored)
	float own[NLO];
	{
		const float* mrow = b.mag + (ring_base + base_slot) * N;
		float m[NLO];
#pragma unroll
		for (int i = 0; i < NLO; ++i) {
			ZH_CHK(mrow + (i < 8 ? tf + i * TF : N / 2), 1);
			m[i] = mrow[i < 8 ? tf + i * TF : N / 2]; // (slot 8: every thread reads bin nfft/2, thread 0 uses it)
		}
		{ // (1 / (|S| |S|)) * 1 (hps.h:91-98, :45-56) by the short exact reciprocal (exact_div.h)
			float sq[NLO], z[NLO];
#pragma unroll
			for (int i = 0; i < NLO; ++i)
				sq[i] = m[i] * m[i];
			zdiv::recip_batch<NLO>(sq, z);
#pragma unroll
			for (int i = 0; i < NLO; ++i)
				own[i] = z[i] * 1.0F;
		}
#pragma unroll
		for (int i = 0; i < NLO; ++i) {
			const int idx = i < 8 ? tf + i * TF : N / 2;
			if (i < 8 || tf == 0) {
				pre[HALO + idx] = own[i];
				if (idx != 0 && idx != N / 2)
					pre[HALO + N - idx] = own[i]; // |S[n-k]| == |S[k]| bit for bit
			}
		}
	}
	auto tap_sum = [&](int j0, const float (&m)[4][NLO], const int (&d)[4], float (&acc)[NLO]) {
#pragma unroll
		for (int jj = 0; jj < 4; ++jj) {
			if (j0 + jj < b.len_t) {
				if (d[jj] == 0) { // the consumed row itself: its 1/|S|^2 is at hand
#pragma unroll
					for (int i = 0; i < NLO; ++i)
						acc[i] = (j0 + jj) == 0 ? own[i] : acc[i] + own[i];
				}
				else {
					float sq[NLO], z[NLO];
#pragma unroll
					for (int i = 0; i < NLO; ++i)
						sq[i] = m[jj][i] * m[jj][i];
					zdiv::recip_batch<NLO>(sq, z);
#pragma unroll
					for (int i = 0; i < NLO; ++i) {
						const float v = z[i] * 1.0F;
						acc[i] = (j0 + jj) == 0 ? v : acc[i] + v;
					}
				}
			}
		}
	};
	// ---- time box: rows ar - mid_t .. ar + mid_t in ascending order (box_time_kernel: clamped rows, causal_self)
	{
		float acc[NLO];
		tap_sum(0, m0, d0, acc);
		for (int j0 = 4; j0 < b.len_t; j0 += 4) {
			float m[4][NLO];
			int d[4];
			tap_load(j0, m, d);
			tap_sum(j0, m, d, acc);
		}
		const float flen_t = (float)b.len_t;
		float rt[NLO], zt[NLO];
		zdiv::div_const_batch<NLO>(acc, flen_t, 1.0f / flen_t, rt); // the box mean: sum / length (box.h:266-286)
		zdiv::recip_batch<NLO>(rt, zt);
#pragma unroll
		for (int i = 0; i < NLO; ++i) {
			if (i < 8 || tf == 0)
				Hrow[i < 8 ? tf + i * TF : N / 2] = zt[i] * b.fac_h; // (1 / mean) * (l_harm + 1): hps.cu:602-604
		}
	}
	zfft::frame_sync<TF>();
	{ // replicate border of the 1/|S|^2 row
		const float v0 = pre[HALO], v1 = pre[HALO + N - 1];
		for (int g = tf; g < HALO; g += TF) {
			pre[g] = v0;
			pre[HALO + N + g] = v1;
		}
	}
	zfft::frame_sync<TF>();
	// ---- frequency box of the thread's sixteen bins: taps idx - mid_f .. idx + mid_f in ascending order (box_freq_kernel);
	// the sixteen sums are interleaved tap by tap: independent chains hide each other's latency
	float hm[16], pm[16];
	{
		float accf[16];
		const float* p0 = pre + HALO + tf - mid_f;
#pragma unroll
		for (int i = 0; i < 16; ++i)
			accf[i] = p0[i * TF];
		for (int j = 1; j < b.len_f; ++j) {
			// sixteen reads in flight, then the sixteen sums: left to itself the scheduler pairs every two reads with their wait
			// (two taps of one bin at a time: 190 trips to the LDS per frame and thread, one after the other)
			float t[16];
#pragma unroll
			for (int i = 0; i < 16; ++i)
				t[i] = p0[i * TF + j];
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int i = 0; i < 16; ++i)
				accf[i] = accf[i] + t[i];
			__builtin_amdgcn_sched_barrier(0);
		}
		const float flen_f = (float)b.len_f;
		float rf[16], zf[16];
		zdiv::div_const_batch<16>(accf, flen_f, 1.0f / flen_f, rf);
		zdiv::recip_batch<16>(rf, zf);
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int idx = tf + i * TF;
			pm[i] = zf[i] * b.fac_p; // (1 / mean) * (l_perc + 1): hps.cu:599-601
			hm[i] = Hrow[idx > N / 2 ? N - idx : idx];
			if (idx > N / 2)
				z[i].y = -z[i].y;
		}
	}
	// ---- the masks of every output, before the image is overwritten
	const MaskCfg cfg{a.beta, a.beta_h, a.soft, a.power, 1, a.out_h, a.out_p};
	float mk[2][16];
#pragma unroll
	for (int oi = 0; oi < 2; ++oi) {
		if (oi < a.n_out) {
			const int which = a.out_id[oi];
#pragma unroll
			for (int i = 0; i < 16; ++i)
				mk[oi][i] = mask_value(which, hm[i], pm[i], cfg);
		}
	}
	zfft::frame_sync<TF>(); // every thread has read the rows: the transforms may use the image
#pragma unroll
	for (int oi = 0; oi < 2; ++oi) {
		if (oi < a.n_out) {
			SynthIn in{z, mk[oi]};
			SynthOut out{a.Y[oi] + (long long)s * a.y_stream_stride + (long long)f * (N / 2), a.cola};
			int tf_o = tf; // (opaque per output: otherwise every LDS address of the transform is hoisted out of the loop, istft.hip)
			int tw_off = 0; // (an opaque offset, not an opaque pointer: the table keeps its address space, istft.hip)
			asm volatile("" : "+v"(tf_o));
			asm volatile("" : "+s"(tw_off));
			const float2* tw_o = a.tw + tw_off;
			zfft::fft_frame<LOG2N, true, false, true>(tf_o, img, tw_o, in, out, active);
			if (oi + 1 < a.n_out)
				zfft::frame_sync<TF>(); // the image is reused by the next output
		}
	}
}

template <int LOG2N>
int launch_sse_block_t(const SseBlockArgs& b, hipStream_t stream)
{
	using PL = Plan<LOG2N>;
	auto kern = sse_synth_kernel<LOG2N>;
	ZH_TRY(set_lds(kern, lds_bytes<LOG2N>()));
	dim3 grid((unsigned)ceil_div((size_t)b.ia.n_frames, (size_t)PL::FRAMES_PER_BLOCK), 1, (unsigned)b.ia.n_streams);
	hipLaunchKernelGGL(kern, grid, dim3(PL::THREADS), lds_bytes<LOG2N>(), stream, b);
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace

// transform sizes 512 .. 4096 (hops 128 .. 1024), boxes that fit the LDS border, at most two outputs (the SSE path has
// no residual, hps.cu:582-652), rows of the ring within an int
bool sse_block_available(int log2n, int len_t, int len_f, int n_out, long long ring_rows)
{
	// (len_t <= 255: the box means divide by the mask length through zdiv::div_const_batch, proven for 1..255: tools/check_div.hip)
	return log2n >= 9 && log2n <= 12 && len_f >= 1 && len_f <= 2 * HALO - 1 && len_t >= 1 && len_t <= 255 && len_t < ring_rows && n_out >= 1
	       && n_out <= 2 && ring_rows < 0x7fffffffLL && !g_opt_no_sse_block;
}

int launch_sse_block(int log2n, const SseBlockArgs& b, hipStream_t stream)
{
	if (b.ia.n_frames <= 0 || b.ia.n_out <= 0)
		return ZEN_HIP_OK;
	switch (log2n) {
	case 9: return launch_sse_block_t<9>(b, stream);
	case 10: return launch_sse_block_t<10>(b, stream);
	case 11: return launch_sse_block_t<11>(b, stream);
	case 12: return launch_sse_block_t<12>(b, stream);
	default: ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "no fused SSE synthesis for nfft 2^%d", log2n);
	}
}

} // namespace zen_hip_impl
