// sse_block.h -- launch interface of the fused SSE synthesis for blocks of frames (sse_block.hip).
#pragma once
#include "stft.h"

namespace zen_hip_impl {

struct SseBlockArgs {
	IstftArgs ia;            // spectrum ring, consumed rows, outputs, COLA (H, P and the mask-bit fields are not used)
	const float* mag;        // magnitude ring [n_streams][ring_rows][nfft]; bins 0..nfft/2 of every row are read
	long long clamp_lo, clamp_hi; // time taps are clamped to these absolute rows (FilterArgs)
	int len_t, len_f;        // odd box lengths (time, frequency)
	int causal_self;         // a time tap past the consumed row is that row again (hps.h:265-268)
	float fac_h, fac_p;      // l_harm + 1, l_perc + 1 (hps.cu:599-604)
};

bool sse_block_available(int log2n, int len_t, int len_f, int n_out, long long ring_rows);
int launch_sse_block(int log2n, const SseBlockArgs& b, hipStream_t stream);

} // namespace zen_hip_impl
