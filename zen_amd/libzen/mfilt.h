// mfilt.h -- MedianFilterGPU over the C-ABI.  Same surface as the reference's libzen/mfilt.h:27-31
// (direction enum) and :33-268 (ctor time/frequency/filter_len/dir/copy_bord, filter(vec,vec),
// filter(device_ptr,device_ptr)); throws ZgException("median filter bigger than matrix dimension")
// on the same condition (mfilt.h:78-86).  Semantics are the reference CPU path's (MedianFilterCPU,
// mfilt.h:270-342): centred odd mask, replicate border; copy_bord has no effect.
#ifndef ZG_MFILT_INTERNAL_H
#define ZG_MFILT_INTERNAL_H

#include <devvec.h>

namespace zen {
namespace internal {
	namespace hps {
		namespace mfilt {
			enum MedianFilterDirection {
				TimeCausal,
				TimeAnticausal,
				Frequency,
			};

			class MedianFilterGPU {
			public:
				MedianFilterDirection mydir;
				int time;
				int frequency;
				int filter_len;
				bool copy_bord;

				MedianFilterGPU(int time, int frequency, int filter_len, MedianFilterDirection dir,
				                bool copy_bord = false)
				    : mydir(dir)
				    , time(time)
				    , frequency(frequency)
				    , filter_len(filter_len)
				    , copy_bord(copy_bord)
				    , h(nullptr)
				{
					throw_or_die(
					    zen_hip_mfilt_create(time, frequency, filter_len, (int)dir, copy_bord ? 1 : 0, &h),
					    "MedianFilterGPU");
				}
				MedianFilterGPU(const MedianFilterGPU&) = delete;
				~MedianFilterGPU() { zen_hip_mfilt_destroy(h); }

				void filter(zen::internal::device_vector<float>& src, zen::internal::device_vector<float>& dst)
				{
					filter(src.data(), dst.data());
				}

				void filter(thrust::device_ptr<float> src, thrust::device_ptr<float> dst)
				{
					throw_or_die(zen_hip_mfilt_run(h, src.get(), dst.get(), nullptr), "MedianFilterGPU::filter");
				}

			private:
				zen_hip_mfilt_t h;
			};
		} // namespace mfilt
	} // namespace hps
} // namespace internal
} // namespace zen

#endif /* ZG_MFILT_INTERNAL_H */
