// libzen/hps.h -- public separation classes of the MI355X drop-in.
//
// Interface-compatible with sevagh/Zen's libzen/libzen/hps.h:25-118: the same two class templates on
// zen::Backend, the same constructor overloads and defaults (hop 256, beta 2.0 for realtime;
// 4096 / 256 / 2.0 / 2.0 for offline), the same method names, the OUTPUT_* flags.  Only Backend::GPU is
// instantiated here (zen_amd/libzen/hps.cpp); the CPU restatement lives under oracle/ as the checker.
#ifndef ZG_HPS_PUB_H
#define ZG_HPS_PUB_H

#include <array>
#include <cstddef>
#include <vector>

#include <libzen/device_ptr.h>
#include <libzen/io.h>
#include <libzen/zen.h>

namespace zen {
namespace internal {
	namespace hps {
		template <zen::Backend B>
		class HPR; // the streaming engine object behind both public classes (hps.h, hps.cpp)
	}
} // namespace internal

namespace hps {
	// which separated signals an engine produces (bit flags, may be or-ed)
	const unsigned int OUTPUT_HARMONIC = 1;
	const unsigned int OUTPUT_PERCUSSIVE = 1 << 1;
	const unsigned int OUTPUT_RESIDUAL = 1 << 2;

	// Causal, hop-by-hop separation for streaming use.  One call consumes `hop` new samples; each
	// copy_* hands out the `hop` oldest samples of that output's overlap-add accumulator and returns
	// once they are visible to the host (so a mapped IOGPU::host_out can be read right away).
	template <zen::Backend B>
	class HPRRealtime {
	public:
		HPRRealtime(float fs, unsigned int output_flags);                  // hop 256, beta 2.0
		HPRRealtime(float fs, std::size_t hop, unsigned int output_flags); // beta 2.0
		HPRRealtime(float fs, std::size_t hop, float beta, unsigned int output_flags);
		// nocopybord: accepted for source compatibility; no effect under the replicate-border
		// (reference CPU) filter semantics this backend implements
		HPRRealtime(float fs, std::size_t hop, float beta, unsigned int output_flags, bool nocopybord);
		~HPRRealtime();
		HPRRealtime(const HPRRealtime&) = delete;
		HPRRealtime& operator=(const HPRRealtime&) = delete;

		void use_sse_filter(); // box-filter (stochastic spectrum estimation) variant instead of medians
		void use_soft_mask();  // Wiener-style masks with exponent (int)beta instead of binary masks

		void process_next_hop(thrust::device_ptr<float> in);
		void copy_percussive(thrust::device_ptr<float> out);
		void copy_harmonic(thrust::device_ptr<float> out);
		void copy_residual(thrust::device_ptr<float> out);

		// raw-pointer forms (the reference's CPU-backend signatures): device-accessible memory here
		void process_next_hop(float* in);
		void copy_percussive(float* out);
		void copy_harmonic(float* out);
		void copy_residual(float* out);

		void warmup(zen::io::IOGPU& io); // 1000 throw-away hops, then a state reset
		void warmup();

		// MI355X extension (off by default): serve process_next_hop / copy_* from one workgroup that stays on the device
		// between hops instead of a launch per hop; it leaves after idle_ms without a hop (0: off).  Same samples.
		void use_resident_kernel(int idle_ms);

		// MI355X extension: n_hops consecutive hops in one call (bit-identical to n_hops single calls);
		// null output pointers are skipped
		void process_hops(thrust::device_ptr<float> in, std::size_t n_hops, thrust::device_ptr<float> harm,
		                  thrust::device_ptr<float> perc, thrust::device_ptr<float> resid);

		// MI355X extension: the same for HOST buffers (n_hops * hop floats in, n_hops * hop floats into every non-null
		// output) -- what zen/fakert.h:221-247 does hop by hop (host hop in, process_next_hop, copy_*, host hop out), for a
		// whole block, its pieces going up / through the engine / back down on three streams.  Returns when the outputs are
		// in the caller's buffers; buffers from zen::io::IOGPU (or any pinned memory) are copied asynchronously.
		void process_hops_host(const float* in, std::size_t n_hops, float* harm, float* perc, float* resid);

	private:
		zen::internal::hps::HPR<B>* p_impl;
	};

	// Whole-clip, two-pass "HPR-I" separation (Driedger): a large-hop pass for the harmonic part, then a
	// small-hop pass on percussive + residual of the first for the percussive part.
	template <zen::Backend B>
	class HPRIOffline {
	public:
		HPRIOffline(float fs); // 4096 / 256, beta 2.0 / 2.0
		HPRIOffline(float fs, std::size_t hop_h, std::size_t hop_p);
		HPRIOffline(float fs, std::size_t hop_h, std::size_t hop_p, float beta_h, float beta_p);
		HPRIOffline(float fs, std::size_t hop_h, std::size_t hop_p, float beta_h, float beta_p, bool nocopybord);
		~HPRIOffline();
		HPRIOffline(const HPRIOffline&) = delete;
		HPRIOffline& operator=(const HPRIOffline&) = delete;

		void use_sse_filter();
		void use_soft_mask();

		// {harmonic, percussive, residual}, each audio.size() long; throws ZgException if hop_h is not
		// a multiple of hop_p (at construction)
		std::array<std::vector<float>, 3> process(std::vector<float> audio);

	private:
		zen::internal::hps::HPR<B>* p_impl_h; // kept for layout familiarity; the engine handle owns both passes
		zen::internal::hps::HPR<B>* p_impl_p;
		std::size_t hop_h, hop_p;
		void* engine; // zen_hip_hpri_t
	};
} // namespace hps
} // namespace zen

#endif /* ZG_HPS_PUB_H */
