// filters.h -- launch interface of the 2-D median / box kernels (median.hip, box.hip).
//
// A "matrix" is rows x cols floats, cols (frequency) contiguous.  The same kernels serve
//   - the drop-in wrappers (MedianFilterGPU::filter / BoxFilterGPU::filter, libzen/mfilt.h:227-267,
//     libzen/box.h:182-214): whole matrix in, whole matrix out;
//   - the streaming engine (hpr.hip): the source is a ring of spectrogram rows (row r lives at
//     r % ring_rows), only the rows that a chunk consumes are produced.
#pragma once
#include <hip/hip_runtime.h>

namespace zen_hip_impl {

struct FilterArgs {
	const float* src;       // row r of stream s at src[s*src_stream_stride + (r % ring_rows)*cols]
	float* dst;             // output row i of stream s at dst[s*dst_stream_stride + i*cols]
	long long src_stream_stride;
	long long dst_stream_stride;
	int n_streams;
	int cols;
	long long ring_rows;    // rows in the source ring (== rows of the matrix for the drop-in call)
	long long first_row;    // absolute source row of output row 0
	int n_out_rows;
	long long clamp_lo;     // time direction: taps are clamped to [clamp_lo, clamp_hi] (replicate border)
	long long clamp_hi;
	int len;                // odd mask length
	int direction;          // ZEN_HIP_TIME_* or ZEN_HIP_FREQUENCY
	int causal_self;        // time direction: additionally clamp taps to <= the output's own row
	                        // (the sliding matrix's last row is the consumed row; hps.h:265-268)
	// box only -- the SSE path's element-wise wrappers (hps.cu:586-604), fused into the filter:
	int sse_pre;            // tap value = 1 / (x*x)    (complex_abs_squared + reciprocal_functor(1))
	int sse_post;           // result    = (1 / mean) * post_factor   (reciprocal_functor(l + 1))
	float post_factor;
	int nonneg;             // every source sample is >= +0 (magnitudes): ordering keys are the raw bits
	int force_general;      // tests: skip the sorting-network fast path, use the general wave kernel
	int hermitian;          // frequency direction, engine only: the source row is the magnitude of a Hermitian
	                        // spectrum of which only bins 0..cols/2 are stored (bin c > cols/2 is read as bin
	                        // cols-c); only bins 0..cols/2 and the last len/2 bins of the output are wanted, the
	                        // rest of the output row is left untouched (P[c] == P[cols-c] there, see stft.h).
	                        // Kernels that do not know the flag are not offered such rows (filter_supports_hermitian)
	int pitch;              // floats between consecutive rows of src and dst (0: cols) -- the time-direction
	                        // kernels filter only the stored half of such rows (cols = nfft/2 + 4, pitch = nfft)
	// Engine only, Hermitian frequency-direction launches with hard masks (stft.h IstftArgs::bits): a kernel that knows how
	// compares its result P with the harmonic estimate H right away and writes two mask bits per bin (natural order, see
	// IstftArgs::bits) INSTEAD of the P row; launch_median(..., &bits_done) tells whether the kernel that ran did (1: `bits`, 2: `bits_t`, 3: soft-mask rows, below).
	// hrows null: H is the source row itself (causal / one-tap time median).
	unsigned* bits;
	long long bits_stream_stride; // words
	int bits_row_words;
	double thr_p, thr_h;          // masks.h hard_mask_threshold
	const float* hrows;           // H row i of stream s at hrows[s*h_stream_stride + i*cols]
	long long h_stream_stride;
	unsigned* bits_t;             // a kernel that holds whole rows writes IstftArgs::bits_t right away (bits_done = 2)
	long long bits_t_stream_stride;
	int need_pm, need_hm;         // which of the two masks some enabled output reads (the other bit stays 0)
	// Soft masks, same launches (bits_done = 3): instead of P the kernel stores the percussive soft mask of every wanted
	// bin where P would have gone (dst), and the harmonic one in mh_dst (same layout), soft_mask_functor hps.h:116-129 with
	// the integer exponent `soft_power`; the synthesis then loads one mask value per bin and output instead of H and P
	// and divides nothing (stft.h IstftArgs::mask_rows).  need_pm / need_hm as above.
	// Engine only, with bits_t: the harmonic estimate is the time median of `time_len` taps of the SAME source rows; a kernel
	// that knows how computes it itself instead of reading hrows (median_tf_herm_bits_kernel; median_tf_fused_available)
	int time_len;
	int soft_rows;
	int soft_power;
	float* mh_dst;                // null unless need_hm
	long long mh_stream_stride;
};

// true if launch_median(a) with a.hermitian = 1 is implemented for this (direction, mask, row length)
bool filter_supports_hermitian(int len, int cols);
bool median_tf_fused_available(int time_len, int freq_len, int cols);

int launch_median(const FilterArgs& a, hipStream_t stream, int* bits_done = nullptr);
int launch_median_net(const FilterArgs& a, hipStream_t stream, bool* handled, int* bits_done = nullptr); // masks <= 63 taps
int launch_median47_dpp(const FilterArgs& a, hipStream_t stream, bool* handled);  // 47 taps, frequency, 4096-bin rows
int launch_median_big(const FilterArgs& a, hipStream_t stream, bool* handled, int* bits_done = nullptr); // 65/85/93/129/171/187/255 taps, frequency
int launch_box(const FilterArgs& a, hipStream_t stream);

} // namespace zen_hip_impl
