// core.h -- the policy layer: TypeTraits<Backend> names the container / FFT / filter / window types that
// one algorithm template is written against (same role and member names as the reference's
// libzen/core.h:17-41).  Only the GPU bundle ships; the CPU bundle is the oracle (test infrastructure).
#ifndef ZG_CORE_H
#define ZG_CORE_H

#include <complex>

#include <box.h>
#include <devvec.h>
#include <fftw.h>
#include <libzen/zen.h>
#include <mfilt.h>
#include <win.h>

namespace zen {
namespace internal {
	namespace core {
		template <zen::Backend>
		struct TypeTraits;

		template <>
		struct TypeTraits<zen::Backend::GPU> {
			using InputPointer = thrust::device_ptr<float>;
			using RealVector = zen::internal::device_vector<float>;
			using ComplexVector = zen::internal::device_vector<std::complex<float>>;
			using Window = zen::internal::win::WindowGPU;
			using FFTC2CWrapper = zen::internal::fftw::FFTC2CWrapperGPU;
			using MedianFilter = zen::internal::hps::mfilt::MedianFilterGPU;
			using BoxFilter = zen::internal::hps::box::BoxFilterGPU;
		};
	} // namespace core
} // namespace internal
} // namespace zen

#endif /* ZG_CORE_H */
