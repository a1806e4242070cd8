#!/usr/bin/env python3
"""The long-mask block-merge kernel (median_big.hip) with and without the shared first merge level: run under two libraries,
    ZEN_HIP_SO=zen_amd/libzen_hip.so python tools/ab_big.py ; ZEN_HIP_SO=zen_amd/libzen_hip_noshare.so python tools/ab_big.py
0.4 s of back-to-back launches per shape through the drop-in wrapper (magnitudes, with the promise)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zen_amd  # noqa: E402

zen_amd.init(0)
for rows, cols, flen in ((6460, 16384, 187), (12920, 8192, 93), (6460, 16384, 171), (12920, 8192, 129), (6460, 16384, 255), (4690, 16384, 257), (12920, 8192, 85)):
    rng = np.random.default_rng(flen)
    src = zen_amd.DeviceBuffer.from_host(rng.random((rows, cols), dtype=np.float32))
    dst = zen_amd.DeviceBuffer(rows * cols)
    f = zen_amd.MedianFilterGPU(rows, cols, flen, zen_amd.FREQUENCY)
    f.assume_nonneg()
    for _ in range(3):
        f.filter(src, dst)
    zen_amd.synchronize()
    e0, e1 = zen_amd.Event(), zen_amd.Event()
    n, t0 = 0, time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(10):
            f.filter(src, dst)
        n += 10
        zen_amd.synchronize()
    e1.record()
    ms = e0.elapsed_ms(e1) / n
    print(json.dumps({"lib": os.environ.get("ZEN_HIP_SO", "default"), "rows": rows, "cols": cols, "taps": flen, "ms": round(ms, 4),
                      "frac": round(8.0 * rows * cols / (1e-3 * ms) / 8e12, 4)}), flush=True)
    src.free()
    dst.free()
