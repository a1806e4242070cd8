#!/usr/bin/env python3
"""How fast are the configurations whose frequency mask is 257 taps (fs / hop = 7.8125)?  Offline two-pass on a 10-minute clip
resident in HBM, and the realtime block path, at 32 kHz / hop 4096 and 16 kHz / hop 2048, beside 44.1 kHz."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zen_amd  # noqa: E402

zen_amd.init(0)
rng = np.random.default_rng(0)
for fs, hop_h, hop_p in ((44100.0, 4096, 256), (32000.0, 4096, 256), (16000.0, 2048, 128), (8000.0, 1024, 64)):
    n = int(600 * fs)
    x = (0.3 * rng.uniform(-1, 1, n)).astype(np.float32)
    g = zen_amd.HPRIOffline(fs, hop_h, hop_p, 2.0, 2.0)
    din, dh, dp = zen_amd.DeviceBuffer.from_host(x), zen_amd.DeviceBuffer(n), zen_amd.DeviceBuffer(n)
    for _ in range(2):
        g.process_device(din.ptr, n, n, harm=dh.ptr, perc=dp.ptr, resid=None, out_stride=n)
    zen_amd.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g.process_device(din.ptr, n, n, harm=dh.ptr, perc=dp.ptr, resid=None, out_stride=n)
    zen_amd.synchronize()
    dt = (time.perf_counter() - t0) / 3
    e = zen_amd.HPR(fs, hop_h, 2.0, zen_amd.OUTPUT_PERCUSSIVE, zen_amd.TIME_CAUSAL, True, 1, 0)
    M = n // hop_h
    dq = zen_amd.DeviceBuffer(n)
    for _ in range(2):
        e.process(din.ptr, M, M * hop_h, None, dq.ptr, None, M * hop_h)
    zen_amd.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        e.process(din.ptr, M, M * hop_h, None, dq.ptr, None, M * hop_h)
    zen_amd.synchronize()
    dtb = (time.perf_counter() - t0) / 3
    print(json.dumps({"fs": fs, "hop_h": hop_h, "hop_p": hop_p, "freq_len": e.freq_len, "offline_10min_ms": 1e3 * dt, "offline_x_realtime": 600.0 / dt,
                      "realtime_block_ms": 1e3 * dtb, "realtime_block_hops_per_s": M / dtb}), flush=True)
    for b in (din, dh, dp, dq):
        b.free()
