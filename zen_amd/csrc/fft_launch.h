// fft_launch.h -- launch helpers shared by the transform kernels' translation units (stft.hip, istft.hip).
#pragma once
#include "common.h"
#include "fft_dev.h"

namespace zen_hip_impl {

template <int LOG2N>
constexpr size_t lds_bytes()
{
	return sizeof(float2) * (size_t)zfft::Plan<LOG2N>::LDS_FLOAT2 * zfft::Plan<LOG2N>::FRAMES_PER_BLOCK;
}

template <class K>
int set_lds(K kern, size_t bytes)
{
	if (bytes > 64 * 1024)
		ZH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
	return ZEN_HIP_OK;
}

} // namespace zen_hip_impl

#define ZH_DISPATCH_LOG2N(log2n, CALL)                                                     \
	switch (log2n) {                                                                       \
	case 5: return CALL(5);                                                                \
	case 6: return CALL(6);                                                                \
	case 7: return CALL(7);                                                                \
	case 8: return CALL(8);                                                                \
	case 9: return CALL(9);                                                                \
	case 10: return CALL(10);                                                              \
	case 11: return CALL(11);                                                              \
	case 12: return CALL(12);                                                              \
	case 13: return CALL(13);                                                              \
	case 14: return CALL(14);                                                              \
	default:                                                                               \
		ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "nfft = 2^%d outside the supported 32..16384", log2n); \
	}

