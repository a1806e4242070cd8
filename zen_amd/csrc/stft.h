// stft.h -- launch interface of the FFT-based kernels (stft.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace zen_hip_impl {

// Analysis: hps.cu:452-472 (input shift/append, window, zero-pad, forward FFT, append to the STFT) and
// the new row of hps.cu:492-493 (magnitude).  One extra block per stream does the per-chunk
// housekeeping: input tail and overlap-add carries.
struct StftArgs {
	const float* in;        // stream s, hop i: in[s*in_stride + i*hop ...]
	long long in_stride;
	long long in_valid;     // samples of every stream's row that exist: sample i >= in_valid reads as zero (the zero
	                        // padding of hps.cu:116-123 without a padded copy of the clip); >= n_frames*hop: all of them
	const float* tail_prev; // [n_streams][hop] : the hop before this chunk (zeros at stream start)
	float* tail_next;       // [n_streams][hop] : receives the last hop of this chunk
	const float* window;    // nwin
	const float2* tw;       // nfft/2
	float2* S;              // ring: [n_streams][ring_rows][s_stride], bins 0..nfft/2 of each frame
	long long s_stride;     // float2 per ring row (>= nfft/2 + 1)
	float* mag;             // ring: [n_streams][ring_rows][nfft]; bins 0..nfft/2 of every new row are written, the
	int mag_full_from;      // mirrored upper half only for frames >= mag_full_from (0: every frame: full rows)
	long long ring_rows;
	long long row0;         // absolute row of the chunk's first frame
	int n_frames;
	int hop;
	int n_streams;
	// overlap-add carry of the previous chunk: carry[o] = second half of its last frame
	int prev_frames;
	float* carry[3];        // [n_streams][hop]
	const float* Y[3];      // [n_streams][max_frames][nwin]
	long long y_stream_stride;
};

// Synthesis: masks (hps.h:100-140), apply (hps.h:58-66), inverse FFT, *COLA (hps.h:68-80) for the
// consumed rows of a chunk; writes Y[o][frame][0..nwin) = Re(ifft(S*mask))*COLA.
struct IstftArgs {
	const float2* S;
	long long s_stride;
	long long ring_rows;
	long long crow0;        // absolute row of the first consumed frame
	const float* H;         // harmonic estimate of the consumed rows
	long long h_stream_stride;
	int h_is_ring;          // 1: H is the magnitude ring itself (causal median)
	const float* P;         // [n_streams][max_frames][nfft]
	long long p_stream_stride;
	int p_mid;              // half length of the frequency mask: P[k] == P[nfft-k] for p_mid < k < nfft/2, so the
	                        // synthesis reads only bins 0..nfft/2 and the last p_mid bins of a P row (and bins
	                        // 0..nfft/2 of an H row, which is symmetric throughout)
	const float2* tw;
	float* Y[3];
	long long y_stream_stride;
	int n_frames;
	int n_streams;
	// single-frame calls: ready[i] (may be null) receives carry[i][k] + Y[i][k], k < hop: the finished hop
	float* ready[3];
	const float* carry[3];
	int hop;
	unsigned seq;           // see RtFusedArgs
	int publish_seq;
	int n_out;              // enabled outputs
	int out_id[3];          // 0 = percussive, 1 = harmonic, 2 = residual (order of Y[])
	float beta, beta_h;     // hps.cu:505 / :540 (beta - Eps)
	int soft, power, sse;
	int out_h, out_p;       // which masks exist for the residual (hps.cu:562-567)
	float cola;
	// hard masks by exact comparison instead of the division (masks.h HardThr; zeros: divide)
	double thr_p, thr_h;
	int thr_p_inc, thr_h_inc;
	// Hard masks decided once per bin by launch_mask_bits: two bits per bin (bit 0 percussive, bit 1 harmonic), 16 bins
	// per word; entry e of a row: bins 0..nfft/2 at e = bin, the last p_mid bins (whose P differs from its mirror image)
	// at e = nfft/2 + 1 + (bin - (nfft - p_mid)).  Null: the synthesis compares H and P itself.
	const unsigned* bits;
	long long bits_stream_stride; // words
	int bits_row_words;
	// the same bits in the order the synthesis threads want them (launch_mask_bits_transpose): word tf of a row (tf <
	// nfft/16) holds, at bits 2s and 2s+1, the masks of bin tf + s*nfft/16 -- the thread's sixteen inputs of the first pass
	const unsigned* bits_t;
	long long bits_t_stream_stride; // words; a row is nfft/16 words
	int need_pm, need_hm;           // which of the two masks some enabled output reads (the other bit stays 0)
	// Soft masks computed by the frequency-direction median kernel (FilterArgs::soft_rows): P holds the percussive mask
	// of every bin the synthesis reads, Hm the harmonic one (rows laid out as P's); H is not read.
	int mask_rows;
	const float* Hm;
	int grid_map;           // set by launch_istft: blockIdx.x encodes (group of frames, output), see istft_kernel
};

// words of one row of mask bits (a multiple of four: rows stay 16-byte aligned)
inline int mask_bits_row_words(int nfft, int p_mid) { return ((nfft / 2 + 1 + p_mid + 15) / 16 + 3) / 4 * 4; }
// The transposed layout gives the row's last p_mid bins (replicate border: not mirror images) entries of their
// own in slot 15 only: the masks-as-bits path covers geometries whose tail fits there.  mf is about 500 Hz of bins
// (hps.h:229), so this fails below fs of roughly 4 kHz (fs 2000, hop 64: nfft 256, mf 65, p_mid 32 against 16); those take
// the H / P rows path.
inline bool mask_bits_supported(int nfft, int p_mid) { return nfft >= 256 && p_mid <= nfft / 16; }

// overlap-add of consecutive frames (hps.cu:435-449 + :526-528) and copy-out (hps.cu:341-363):
// out[i*hop + n] = (i ? Y[i-1][hop+n] : carry[n]) + Y[i][n]
struct FinalizeArgs {
	const float* Y;
	const float* carry;
	float* out;
	long long y_stream_stride;
	long long out_stride;
	int n_frames, hop, n_streams;
	// launch_finalize_spec only.  A second output whose finished hops are added (hps.cu:153-160: xp1 + xr1; null Y2
	// with add_zero: the reference adds its all-zero accumulator), and where the samples go: the sample at stream
	// position p = pos0 + i*hop + k is written to out[p - shift] if 0 <= p - shift < len (the reference's "drop the
	// lag*hop delay, truncate to the clip length", hps.cu:171-178, folded into the overlap-add), and positions
	// p >= dup_from ALSO to out[p - dup_shift] if that is in [0, dup_len) (the leftovers the in-place shift of
	// hps.cu:171-176 leaves behind, which pass 2 reads: SURVEY Q9).
	const float* Y2;
	const float* carry2;
	int add_zero;
	long long pos0, shift, len, dup_from, dup_shift, dup_len;
};

// Synthesis in runs (istft.hip istft_run_kernel / istft_run_wide_kernel; the passes of the offline driver): hard masks from
// IstftArgs::bits_t; a wavefront (nfft <= 1024) or a workgroup walks a run of consecutive frames of a stream with the
// overlap-add carry in registers and writes the finished hops where FinalizeArgs (second part) says: no Y rows, no
// overlap-add launch.  A GROUP is what one destination receives: the finished hops of one output, or the sum of those of
// two (FinalizeArgs::Y2: P + R of pass 1, hps.cu:153-160).
struct IstftRunGroup {
	int n_out;               // 1 or 2
	int which[2];            // output ids (0 percussive, 1 harmonic, 2 residual)
	const float* carry_prev[2]; // [n_streams][hop]: second half of the frame before frame 0 (the previous call's carry_next)
	float* carry_next[2];       // receives the second half of frame n_frames - 1
	float* out;              // destination of the finished hops, as FinalizeArgs: out, out_stride, shift, len, dup_*
	long long out_stride, shift, len, dup_from, dup_shift, dup_len;
};
struct IstftRunArgs {
	const float2* S;        // the spectrum ring (IstftArgs::S, s_stride, ring_rows): frame i of the call is row crow0 + i
	long long s_stride, ring_rows, crow0;
	const float2* tw;
	const unsigned* bits_t; // row i: the masks of frame i (IstftArgs::bits_t)
	long long bits_t_stream_stride;
	int n_frames, n_streams, hop;
	int out_h, out_p;        // which masks the residual reads (hps.cu:562-567)
	float cola;
	int run;                 // consecutive frames per run (each run but a call's first synthesises one frame more)
	long long pos0;          // FinalizeArgs::pos0
	int n_groups;            // nfft <= 1024: one group of one output
	float* sink;             // [n_streams][hop] floats nobody reads (istft_run_wide_kernel: a turn without a finished hop stores there)
	IstftRunGroup g[3];
};
bool istft_run_available(int log2n, int n_groups, int max_outputs_per_group);
int launch_istft_run(int log2n, const IstftRunArgs& a, hipStream_t stream);

int launch_stft(int log2n, const StftArgs& a, hipStream_t stream);
int launch_istft(int log2n, const IstftArgs& a, hipStream_t stream);
// fills IstftArgs::bits (passed as `bits`, writable) for the frames / streams of `a` from its H and P rows
int launch_mask_bits(int nfft, const IstftArgs& a, unsigned* bits, hipStream_t stream);
// IstftArgs::bits -> IstftArgs::bits_t (passed as `bits_t`, writable)
int launch_mask_bits_transpose(int nfft, const IstftArgs& a, unsigned* bits_t, hipStream_t stream);
int launch_finalize(const FinalizeArgs& a, hipStream_t stream);
int launch_finalize_spec(const FinalizeArgs& a, hipStream_t stream);
// FFTC2CWrapperGPU::forward/backward (fftw.h:35-43), `batch` consecutive transforms in place
int launch_fft(int log2n, float2* data, const float2* tw, size_t batch, int inverse, hipStream_t stream);
// the same for nfft = 32768 (fft_big.hip: two steps through `xch`, batch * nfft float2 of scratch)
int launch_fft_big(int log2n, float2* data, float2* xch, const float2* tw, size_t batch, int inverse, hipStream_t stream);

} // namespace zen_hip_impl
