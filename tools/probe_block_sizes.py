#!/usr/bin/env python3
"""What a block call of zen_hip_hpr_process costs as a function of its length (hop 1024, P only, resident input): microseconds
per call, hops/s, and the engine's own per-kernel times.  On the GPU box: python tools/probe_block_sizes.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zen_amd
zen_amd.init(0)
hop = 1024
rng = np.random.default_rng(0)
for M in (16, 64, 128, 256, 512, 1024, 1025, 1100, 1292, 2048, 4096, 25840):
    x = rng.uniform(-1, 1, hop * M).astype(np.float32)
    din, dout = zen_amd.DeviceBuffer.from_host(x), zen_amd.DeviceBuffer(x.size)
    eng = zen_amd.HPR(44100.0, hop, 2.0, zen_amd.OUTPUT_PERCUSSIVE, zen_amd.TIME_CAUSAL, True, 1, 0)
    for _ in range(5):
        eng.process(din.ptr, M, x.size, None, dout.ptr, None, x.size)
    zen_amd.synchronize()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(50):
        eng.process(din.ptr, M, x.size, None, dout.ptr, None, x.size)
    zen_amd.synchronize()
    dt = (time.perf_counter() - t0) / 50
    pr = {k: round(v["ms"] / 50, 4) for k, v in eng.profile_get_all().items() if v["launches"]}
    eng.profile(False)
    print(M, "%.1f us per call, %.2f M hops/s" % (1e6 * dt, M / dt / 1e6), pr, flush=True)
    din.free(); dout.free()
