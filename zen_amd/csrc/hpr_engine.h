// hpr_engine.h -- state of one streaming HPSS engine (zen_hip_hpr_t), shared by hpr.hip (the engine and its
// C-ABI) and hpri.hip (the two-pass offline driver built on two engines).
#pragma once
#include "common.h"

#include <utility>
#include <vector>

struct zen_hip_hpr {
	float fs;
	size_t hop, nwin, nfft;
	float beta;
	int l_harm, l_perc, lag;
	size_t W;
	int causality, log2n, mt, mf;
	float cola;
	bool out_h, out_p, out_r, use_sse, soft;
	size_t n_streams, max_hops;
	long long ring_rows;
	size_t s_stride; // float2 per spectrum-ring row: nfft/2 + 1 bins, padded to a 64-byte multiple
	hipStream_t stream;

	float* d_window = nullptr;
	float2* d_tw = nullptr;
	float* d_tail[2] = {nullptr, nullptr};
	int tail_sel = 0;
	float2* d_S = nullptr;
	float* d_mag = nullptr;
	float* d_H = nullptr;
	float* d_P = nullptr;
	float* d_Y[3] = {nullptr, nullptr, nullptr};     // 0 percussive, 1 harmonic, 2 residual
	float* d_carry[3] = {nullptr, nullptr, nullptr};
	long long abs_frame = 0;
	size_t last_frames = 0;

	// profiling hook (bench.py): HIP events around every launch, per kernel class
	enum { K_STFT = 0, K_FREQ = 1, K_TIME = 2, K_ISTFT = 3, K_FINALIZE = 4, K_FUSED = 5, K_COUNT = 6 };
	bool prof = false;
	double prof_ms[K_COUNT] = {0, 0, 0, 0, 0, 0};
	unsigned long long prof_launches[K_COUNT] = {0, 0, 0, 0, 0, 0};
	unsigned long long prof_elements = 0; // elements filtered by the frequency-direction kernel
	struct Pending {
		int k;
		hipEvent_t e0, e1;
	};
	std::vector<Pending> prof_pending;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
};
