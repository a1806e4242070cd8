// host_pipe.h -- what the two host-buffer pipelines share (hpri.hip zen_hip_hpri_process: one clip through the two offline passes;
// hpr.hip zen_hip_hpr_process_host: a block of hops through a realtime engine): is a host range pinned, and pinning a
// caller's pageable buffer for the duration of one call.
#pragma once
#include <hip/hip_runtime.h>

namespace zen_hip_impl {

// Is this host range known to the runtime (hipHostMalloc / hipHostRegister)?  Copies from / to such memory are
// asynchronous; from / to pageable memory hipMemcpyAsync returns when the bytes have moved.
inline bool host_pinned(const void* p)
{
	hipPointerAttribute_t at;
	if (hipPointerGetAttributes(&at, p) == hipSuccess)
		return at.type == hipMemoryTypeHost;
	(void)hipGetLastError();
	return false;
}

struct Registered { // a caller's buffer pinned for the duration of one call
	void* p = nullptr;
	bool pinned = false;
	void take(const void* q, size_t bytes, bool try_register)
	{
		if (!q)
			return;
		if (host_pinned(q)) {
			pinned = true;
			return;
		}
		if (try_register && hipHostRegister(const_cast<void*>(q), bytes, hipHostRegisterDefault) == hipSuccess) {
			p = const_cast<void*>(q);
			pinned = true;
			return;
		}
		(void)hipGetLastError();
	}
	~Registered()
	{
		if (p)
			(void)hipHostUnregister(p);
	}
};

} // namespace zen_hip_impl
