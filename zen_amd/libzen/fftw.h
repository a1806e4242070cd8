// fftw.h -- FFTC2CWrapperGPU over the C-ABI.  Same surface as the reference's libzen/fftw.h:20-49:
// ctor (nfft), public `nfft`, public in-place buffer `fft_vec`, forward(), backward(); unnormalised in
// both directions (cuFFT semantics).  The transform itself is the hand-written LDS FFT of
// zen_amd/csrc/fft_dev.h, reached through zen_hip_fft_*.
#ifndef ZG_FFT_INTERNAL_H
#define ZG_FFT_INTERNAL_H

#include <complex>
#include <cstddef>

#include <devvec.h>

namespace zen {
namespace internal {
	namespace fftw {
		class FFTC2CWrapperGPU {
		public:
			std::size_t nfft;

			zen::internal::device_vector<std::complex<float>> fft_vec;

			FFTC2CWrapperGPU(std::size_t nfft)
			    : nfft(nfft)
			    , fft_vec(nfft)
			    , plan(nullptr)
			{
				throw_or_die(zen_hip_fft_create(nfft, &plan), "FFTC2CWrapperGPU");
			}
			FFTC2CWrapperGPU(const FFTC2CWrapperGPU&) = delete;
			~FFTC2CWrapperGPU() { zen_hip_fft_destroy(plan); }

			void forward() { throw_or_die(zen_hip_fft_exec(plan, ptr(), 0, nullptr), "fft forward"); }

			void backward() { throw_or_die(zen_hip_fft_exec(plan, ptr(), 1, nullptr), "fft backward"); }

		private:
			float* ptr() { return reinterpret_cast<float*>(fft_vec.raw()); }
			zen_hip_fft_t plan;
		};
	} // namespace fftw
} // namespace internal
} // namespace zen

#endif /* ZG_FFT_INTERNAL_H */
