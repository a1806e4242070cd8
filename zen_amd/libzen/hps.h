// hps.h (internal) -- HPR<Backend::GPU>, the algorithm object of the reference's libzen/hps.h:152-322,
// as a thin owner of the streaming engine behind the C-ABI (zen_hip_hpr_*).  Public data members keep
// the reference's names (fs, hop, nwin, nfft, beta, l_harm, l_perc, lag, stft_width, COLA_factor, the
// output flags) so code and tests written against HPR<B> read the same; the device state itself
// (sliding STFT, magnitudes, masks, overlap-add accumulators) is owned by the engine.
#ifndef ZG_HPS_INTERNAL_H
#define ZG_HPS_INTERNAL_H

#include <cstddef>
#include <vector>

#include <core.h>
#include <libzen/hps.h>

namespace zen {
namespace internal {
	namespace hps {
		static constexpr float Eps = std::numeric_limits<float>::epsilon();

		template <zen::Backend B>
		class HPR;

		template <>
		class HPR<zen::Backend::GPU> {
			typedef zen::internal::core::TypeTraits<zen::Backend::GPU>::InputPointer InputPointer;

		public:
			float fs;
			std::size_t hop;
			std::size_t nwin;
			std::size_t nfft;
			float beta;
			int l_harm;
			int l_perc;
			int lag;
			std::size_t stft_width;
			float COLA_factor;

			bool output_percussive;
			bool output_harmonic;
			bool output_residual;
			bool use_sse;
			bool soft_mask;

			HPR(float fs, std::size_t hop, float beta, unsigned int output_flags,
			    mfilt::MedianFilterDirection causality, bool copy_bord, std::size_t max_hops_per_call = 64);
			HPR(const HPR&) = delete;
			~HPR();

			void use_sse_filter();
			void use_soft_mask();
			void reset_buffers();

			void process_next_hop(InputPointer in_hop);
			void process_hops(InputPointer in, std::size_t n_hops, float* harm, float* perc, float* resid);

			// first `hop` samples of the accumulators, as copy_* hands them out (hps.cu:341-363)
			void copy_out(unsigned int which, float* out_dev);
			// host snapshots of the same, for tests written like libzen/hps.test.cu
			std::vector<float> percussive_out();
			std::vector<float> harmonic_out();
			std::vector<float> residual_out();

			zen_hip_hpr_t engine;

		private:
			std::vector<float> snapshot(unsigned int which);
		};
	} // namespace hps
} // namespace internal
} // namespace zen

#endif /* ZG_HPS_INTERNAL_H */
