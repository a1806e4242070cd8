#!/usr/bin/env python3
"""Does the time-direction median kernel's rate depend on where its destination lies relative to its source?  One allocation, the
source at its start, the destination at a varying distance behind the source's end; 0.3 s of back-to-back launches each."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zen_amd  # noqa: E402

zen_amd.init(0)
for rows, cols, flen in ((103360, 1024, 11), (25840, 4096, 3), (51680, 2048, 7)):
    n = rows * cols
    big = zen_amd.DeviceBuffer(2 * n + (64 << 20))
    rng = np.random.default_rng(1)
    big.upload(np.concatenate([rng.random(n, dtype=np.float32), np.zeros(n + (64 << 20), np.float32)]))
    f = zen_amd.MedianFilterGPU(rows, cols, flen, zen_amd.TIME_ANTICAUSAL)
    for extra in (0, 64, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096, 3 << 20, 16 << 20):   # floats
        src, dst = big.ptr, big.offset(n + extra)
        for _ in range(3):
            f.filter(src, dst)
        zen_amd.synchronize()
        e0, e1 = zen_amd.Event(), zen_amd.Event()
        k, t0 = 0, time.perf_counter()
        e0.record()
        while time.perf_counter() - t0 < 0.3:
            for _ in range(50):
                f.filter(src, dst)
            k += 50
            zen_amd.synchronize()
        e1.record()
        ms = e0.elapsed_ms(e1) / k
        print(json.dumps({"rows": rows, "cols": cols, "taps": flen, "dst_minus_src_end_bytes": 4 * extra, "src_ptr_mod_2MiB": big.ptr % (2 << 20),
                          "ms": round(ms, 5), "frac": round(8.0 * n / (1e-3 * ms) / 8e12, 4)}), flush=True)
    del f
    big.free()
