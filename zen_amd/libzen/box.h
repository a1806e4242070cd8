// box.h -- BoxFilterGPU over the C-ABI.  Same surface as the reference's libzen/box.h:30-215 (ctor
// time/frequency/filter_len/dir, filter(...)); throws ZgException("box filter bigger than matrix
// dimension") as box.h:69-77.  Semantics: BoxFilterCPU (box.h:217-288), mean with replicate border.
#ifndef ZG_BOX_INTERNAL_H
#define ZG_BOX_INTERNAL_H

#include <devvec.h>
#include <mfilt.h>

namespace zen {
namespace internal {
	namespace hps {
		namespace box {
			using zen::internal::hps::mfilt::MedianFilterDirection;

			class BoxFilterGPU {
			public:
				MedianFilterDirection mydir;
				int time;
				int frequency;
				int filter_len;

				BoxFilterGPU(int time, int frequency, int filter_len, MedianFilterDirection dir)
				    : mydir(dir)
				    , time(time)
				    , frequency(frequency)
				    , filter_len(filter_len)
				    , h(nullptr)
				{
					int rc = zen_hip_box_create(time, frequency, filter_len, (int)dir, &h);
					if (rc == ZEN_HIP_E_FILTER_TOO_BIG)
						throw zen::ZgException("box filter bigger than matrix dimension");
					throw_or_die(rc, "BoxFilterGPU");
				}
				BoxFilterGPU(const BoxFilterGPU&) = delete;
				~BoxFilterGPU() { zen_hip_box_destroy(h); }

				void filter(zen::internal::device_vector<float>& src, zen::internal::device_vector<float>& dst)
				{
					filter(src.data(), dst.data());
				}

				void filter(thrust::device_ptr<float> src, thrust::device_ptr<float> dst)
				{
					throw_or_die(zen_hip_box_run(h, src.get(), dst.get(), nullptr), "BoxFilterGPU::filter");
				}

			private:
				zen_hip_box_t h;
			};
		} // namespace box
	} // namespace hps
} // namespace internal
} // namespace zen

#endif /* ZG_BOX_INTERNAL_H */
