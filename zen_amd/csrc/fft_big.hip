// fft_big.hip -- FFTC2CWrapperGPU (libzen/fftw.h:20-49) for nfft = 32768, the top of the reference's FFT bench
// sweep (libzen/fftw.bench.cu:231-252): a frame of that size does not fit one workgroup's LDS (256 KB of float2),
// so the transform is cut in two steps exactly as rt_wide.hip cuts the 8192 / 16384-point ones, with the exchange
// through a global buffer instead of a grid barrier:
//
//   the radix-2 decimation-in-time DAG of fft_dev.h, N = M*J with M = 128, J = 256:
//     step A (stages 1..7):   J independent M-point transforms of the decimated sequences x[j + n*J]; stage s reads
//                             tw[(k * M/2^s) * J]: every J-th entry of the N-point table.           data -> T[k][j]
//     step B (stages 8..15):  for each kappa < M a J-point transform of Y_7[j][kappa], j < J, whose stage-t twiddle
//                             of frequency q is tw[(q * J/2^t) * M + kappa * J/2^t].         T -> data[kappa + M*q]
//   Every butterfly is the one the oracle's recursion evaluates (tests/test_two_step_fft.py models these indices in
//   float32 numpy and reproduces the oracle's transform bit for bit, sizes 2^13 .. 2^15).
//
// Not on the HPSS path (the engine's largest transform is 16384); a drop-in / bench-parity entry point.
#include "common.h"
#include "fft_dev.h"
#include "stft.h"

#pragma clang fp contract(off)

namespace zen_hip_impl {
namespace {

using zfft::Plan;

constexpr int WGT = 256;

template <int LOG2N>
struct BigGeo {
	static constexpr int N = 1 << LOG2N;
	static constexpr int LOG2M = 7, M = 1 << LOG2M;
	static constexpr int LOG2J = LOG2N - LOG2M, J = 1 << LOG2J;
	using PA = Plan<LOG2M>;
	using PB = Plan<LOG2J>;
	static constexpr int FA = WGT / PA::TF, FB = WGT / PB::TF; // frames per workgroup in either step
	static_assert(PA::TF <= 64 && PB::TF <= 64, "frames are synchronised inside their wavefront");
	static_assert(J % FA == 0 && M % FB == 0, "whole workgroups");
};

template <int LOG2J>
struct StrideTw { // step A: the M-point transform's table index idx names entry idx*J of the N-point table
	const float2* __restrict__ p;
	__device__ __forceinline__ float2 operator()(int, int idx) const { return p[idx << LOG2J]; }
};
template <int LOG2M, int LOG2J>
struct TwistTw { // step B: stage t, plain J-point index idx  ->  entry idx*M + kappa*J/2^t
	const float2* __restrict__ p;
	int kappa;
	__device__ __forceinline__ float2 operator()(int t, int idx) const { return p[(idx << LOG2M) + (kappa << (LOG2J - t))]; }
};

template <int LOG2J>
struct AIn {
	const float2* x;
	int j;
	__device__ __forceinline__ float2 operator()(int n, int) const
	{
		ZH_CHK(x + (j + (n << LOG2J)), 1);
		return x[j + (n << LOG2J)];
	}
};
template <int LOG2J>
struct AOut { // Y_7[j][k] -> T[k][j]: step B reads a column contiguously
	float2* T;
	int j;
	__device__ __forceinline__ void operator()(int k, float2 X, bool, int) const
	{
		ZH_CHK(T + ((k << LOG2J) + j), 1);
		T[(k << LOG2J) + j] = X;
	}
};
template <int LOG2J>
struct BIn {
	const float2* T;
	int kappa;
	__device__ __forceinline__ float2 operator()(int jj, int) const
	{
		ZH_CHK(T + ((kappa << LOG2J) + jj), 1);
		return T[(kappa << LOG2J) + jj];
	}
};
template <int LOG2M>
struct BOut {
	float2* x;
	int kappa;
	__device__ __forceinline__ void operator()(int q, float2 X, bool, int) const
	{
		ZH_CHK(x + (kappa + (q << LOG2M)), 1);
		x[kappa + (q << LOG2M)] = X;
	}
};

template <int LOG2N, bool INV>
__global__ __launch_bounds__(WGT) void fft_big_a_kernel(const float2* __restrict__ data, float2* __restrict__ xch,
                                                        const float2* __restrict__ tw)
{
	using G = BigGeo<LOG2N>;
	using PA = typename G::PA;
	extern __shared__ float2 lds[];
	const int t = threadIdx.x;
	// neighbouring lanes hold neighbouring sequences j (same position inside the sequence): the strided elements
	// x[j + n*J] they ask for are neighbours in memory; a frame still lives inside one wavefront
	constexpr int FW = 64 / PA::TF; // frames per wavefront
	const int tf = (t & 63) / FW, f = (t >> 6) * FW + (t & 63) % FW;
	const int j = blockIdx.x * G::FA + f;
	const long long base = (long long)blockIdx.y * G::N;
	zfft::TwRegs<G::LOG2M> twr;
	twr.fill_with(tf, StrideTw<G::LOG2J>{tw});
	AIn<G::LOG2J> in{data + base, j};
	AOut<G::LOG2J> out{xch + base, j};
	zfft::PassRunner<G::LOG2M, 0, INV, false, false, AIn<G::LOG2J>, AOut<G::LOG2J>, false, zfft::TwRegs<G::LOG2M>>::run(
	    tf, lds + f * PA::LDS_FLOAT2, twr, in, out, true);
}

template <int LOG2N, bool INV>
__global__ __launch_bounds__(WGT) void fft_big_b_kernel(float2* __restrict__ data, const float2* __restrict__ xch,
                                                        const float2* __restrict__ tw)
{
	using G = BigGeo<LOG2N>;
	using PB = typename G::PB;
	extern __shared__ float2 lds[];
	const int t = threadIdx.x;
	const int f = t / PB::TF, tf = t % PB::TF;
	const int kappa = blockIdx.x * G::FB + f;
	const long long base = (long long)blockIdx.y * G::N;
	zfft::TwRegs<G::LOG2J, false> twr; // not the J-point transform's own table: no trivial-twiddle shortcuts
	twr.fill_with(tf, TwistTw<G::LOG2M, G::LOG2J>{tw, kappa});
	BIn<G::LOG2J> in{xch + base, kappa};
	BOut<G::LOG2M> out{data + base, kappa};
	zfft::PassRunner<G::LOG2J, 0, INV, false, false, BIn<G::LOG2J>, BOut<G::LOG2M>, false, zfft::TwRegs<G::LOG2J, false>>::run(
	    tf, lds + f * PB::LDS_FLOAT2, twr, in, out, true);
}

template <int LOG2N, bool INV>
int launch_big_t(float2* data, float2* xch, const float2* tw, size_t batch, hipStream_t stream)
{
	using G = BigGeo<LOG2N>;
	const size_t lds_a = sizeof(float2) * G::FA * G::PA::LDS_FLOAT2, lds_b = sizeof(float2) * G::FB * G::PB::LDS_FLOAT2;
	static_assert(sizeof(float2) * G::FA * G::PA::LDS_FLOAT2 <= 64 * 1024 && sizeof(float2) * G::FB * G::PB::LDS_FLOAT2 <= 64 * 1024,
	              "default dynamic LDS limit");
	for (size_t b0 = 0; b0 < batch; b0 += 65535) { // gridDim.y
		const unsigned nb = (unsigned)(batch - b0 < 65535 ? batch - b0 : 65535);
		float2* d = data + b0 * G::N;
		float2* x = xch + b0 * G::N;
		hipLaunchKernelGGL((fft_big_a_kernel<LOG2N, INV>), dim3(G::J / G::FA, nb), dim3(WGT), lds_a, stream, d, x, tw);
		hipLaunchKernelGGL((fft_big_b_kernel<LOG2N, INV>), dim3(G::M / G::FB, nb), dim3(WGT), lds_b, stream, d, x, tw);
	}
	ZH_HIP(hipGetLastError());
	return ZEN_HIP_OK;
}

} // namespace

// `xch`: batch * nfft float2 of scratch (the values the two steps exchange)
int launch_fft_big(int log2n, float2* data, float2* xch, const float2* tw, size_t batch, int inverse, hipStream_t stream)
{
	if (batch == 0)
		return ZEN_HIP_OK;
	if (log2n != 15)
		ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "two-step transform: nfft = 2^%d (only 32768)", log2n);
	return inverse ? launch_big_t<15, true>(data, xch, tw, batch, stream) : launch_big_t<15, false>(data, xch, tw, batch, stream);
}

} // namespace zen_hip_impl
