// libzen/io.h -- zen::io::IOGPU, signature-compatible with the reference's libzen/libzen/io.h:16-81:
// two pinned, mapped, portable host buffers (host_in also write-combined) and their device aliases,
// obtained through the C-ABI (zen_hip_host_alloc_mapped) instead of cudaHostAlloc.
#ifndef ZG_IO_PUB_H
#define ZG_IO_PUB_H

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <iostream>
#include <numeric>

#include <libzen/device_ptr.h>
#include <zen_hip.h>

namespace zen {
namespace io {
	class IOGPU {
	public:
		float* host_in;
		float* host_out;
		thrust::device_ptr<float> device_in;
		thrust::device_ptr<float> device_out;
		std::size_t size;

		IOGPU(std::size_t size)
		    : size(size)
		{
			void *h = nullptr, *d = nullptr;
			// mapped + write-combined for the input side (io.h:33-35)
			if (zen_hip_host_alloc_mapped(size * sizeof(float), 1, &h, &d) != ZEN_HIP_OK) {
				std::cerr << "IOGPU: hip malloc error: " << zen_hip_last_error() << std::endl;
				std::exit(-1); // io.h:37-41
			}
			host_in = static_cast<float*>(h);
			device_in_raw_ptr = static_cast<float*>(d);
			if (zen_hip_host_alloc_mapped(size * sizeof(float), 0, &h, &d) != ZEN_HIP_OK) {
				std::cerr << "IOGPU: hip malloc error: " << zen_hip_last_error() << std::endl;
				std::exit(-1);
			}
			host_out = static_cast<float*>(h);
			device_out_raw_ptr = static_cast<float*>(d);
			device_in = thrust::device_pointer_cast(device_in_raw_ptr);
			device_out = thrust::device_pointer_cast(device_out_raw_ptr);
		}

		IOGPU(const IOGPU&) = delete;
		IOGPU& operator=(const IOGPU&) = delete;

		~IOGPU()
		{
			zen_hip_host_free(host_in);
			zen_hip_host_free(host_out);
		}

	private:
		float* device_in_raw_ptr;
		float* device_out_raw_ptr;
	};
}; // namespace io
}; // namespace zen

#endif /* ZG_IO_PUB_H */
