// core.h -- the policy layer.  Same role and typedef names as the reference's libzen/core.h:17-41:
// TypeTraits<Backend> bundles the vector / FFT / median / box / window types one algorithm template is
// written against.  Only the GPU bundle ships in the product; the CPU bundle (the oracle-backed
// restatement used as the parity reference) lives under oracle/ and is test infrastructure.
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
		template <zen::Backend T>
		struct TypeTraits {
		};

		template <>
		struct TypeTraits<zen::Backend::GPU> {
			typedef thrust::device_ptr<float> InputPointer;
			typedef zen::internal::device_vector<float> RealVector;
			typedef zen::internal::device_vector<std::complex<float>> ComplexVector;
			typedef zen::internal::fftw::FFTC2CWrapperGPU FFTC2CWrapper;
			typedef zen::internal::hps::mfilt::MedianFilterGPU MedianFilter;
			typedef zen::internal::hps::box::BoxFilterGPU BoxFilter;
			typedef zen::internal::win::WindowGPU Window;
		};
	} // namespace core
} // namespace internal
} // namespace zen

#endif /* ZG_CORE_H */
