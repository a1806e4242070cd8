// libzen/hps.h -- public HPR classes, signature-compatible with the reference's
// libzen/libzen/hps.h:25-118 (same constructors, methods, defaults and pimpl layout idea).
#ifndef ZG_HPS_PUB_H
#define ZG_HPS_PUB_H

#include <array>
#include <complex>
#include <cstddef>
#include <vector>

#include <libzen/device_ptr.h>
#include <libzen/io.h>
#include <libzen/zen.h>

// forward declare private implementations
namespace zen {
namespace internal {
	namespace hps {
		template <zen::Backend B>
		class HPR;
	}; // namespace hps
};     // namespace internal
};     // namespace zen

namespace zen {
namespace hps {
	const unsigned int OUTPUT_HARMONIC = 1;
	const unsigned int OUTPUT_PERCUSSIVE = 1 << 1;
	const unsigned int OUTPUT_RESIDUAL = 1 << 2;

	template <zen::Backend B>
	class HPRIOffline {
	public:
		HPRIOffline(float fs,
		            std::size_t hop_h,
		            std::size_t hop_p,
		            float beta_h,
		            float beta_p);

		// nocopybord is accepted for source compatibility.  The MI355X engine implements the
		// reference CPU (IPP) filter semantics -- centred mask, replicate border -- for which the
		// flag has no effect (reference: libzen/mfilt.h:289 "not used for CPU").
		HPRIOffline(float fs,
		            std::size_t hop_h,
		            std::size_t hop_p,
		            float beta_h,
		            float beta_p,
		            bool nocopybord);

		HPRIOffline(float fs, std::size_t hop_h, std::size_t hop_p);

		HPRIOffline(float fs);
		~HPRIOffline();

		// pass the entire song in the in vec; returns a triplet of harmonic,percussive,residual
		// results of audio.size()
		std::array<std::vector<float>, 3> process(std::vector<float> audio);

		void use_sse_filter();
		void use_soft_mask();

	private:
		// two cascading HPR objects: driedger's offline iterative algorithm "HPR-I"
		zen::internal::hps::HPR<B>* p_impl_h;
		zen::internal::hps::HPR<B>* p_impl_p;

		std::size_t hop_h, hop_p;
		void* engine; // zen_hip_hpri_t (GPU) -- replaces the reference's two IOGPU staging members
	};

	template <zen::Backend B>
	class HPRRealtime {
	public:
		HPRRealtime(float fs,
		            std::size_t hop,
		            float beta,
		            unsigned int output_flags);

		HPRRealtime(float fs,
		            std::size_t hop,
		            float beta,
		            unsigned int output_flags,
		            bool nocopybord);
		HPRRealtime(float fs, std::size_t hop, unsigned int output_flags);
		HPRRealtime(float fs, unsigned int output_flags);
		~HPRRealtime();

		// pass in a real-time stream of the input, one hop at a time
		void process_next_hop(thrust::device_ptr<float> in);

		void copy_harmonic(thrust::device_ptr<float> out);
		void copy_percussive(thrust::device_ptr<float> out);
		void copy_residual(thrust::device_ptr<float> out);

		void process_next_hop(float* in);

		void copy_harmonic(float* out);
		void copy_percussive(float* out);
		void copy_residual(float* out);

		void warmup();
		void warmup(zen::io::IOGPU& io);

		void use_sse_filter();
		void use_soft_mask();

		// MI355X extension (not in the reference): n_hops consecutive hops in one call; any output
		// pointer may be null.  Bit-identical to n_hops process_next_hop + copy_* calls.
		void process_hops(thrust::device_ptr<float> in, std::size_t n_hops, thrust::device_ptr<float> harm,
		                  thrust::device_ptr<float> perc, thrust::device_ptr<float> resid);

	private:
		zen::internal::hps::HPR<B>* p_impl;
	};
}; // namespace hps
}; // namespace zen

#endif /* ZG_HPS_PUB_H */
