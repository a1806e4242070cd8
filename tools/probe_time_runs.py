#!/usr/bin/env python3
"""The time-direction median kernel reads 0.58 of the HBM roof in some processes and 0.65 in others (same build, same box).
One shape, one 0.4 s burst, and where the two buffers lie: run it several times in a row and compare."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zen_amd  # noqa: E402

zen_amd.init(0)
pre = int(os.environ.get("PRE_ALLOC_MB", "0"))
hold = [zen_amd.DeviceBuffer(pre << 18)] if pre else []
rows, cols, flen = 103360, 1024, 11
rng = np.random.default_rng(0)
src = zen_amd.DeviceBuffer.from_host(rng.random((rows, cols), dtype=np.float32))
dst = zen_amd.DeviceBuffer(rows * cols)
out = {"src": hex(src.ptr), "dst": hex(dst.ptr), "src_mod_1GiB_MiB": (src.ptr % (1 << 30)) >> 20, "dst_mod_1GiB_MiB": (dst.ptr % (1 << 30)) >> 20, "pre_alloc_MB": pre}
for name, d in (("time11", zen_amd.TIME_ANTICAUSAL), ("freq13", zen_amd.FREQUENCY)):
    f = zen_amd.MedianFilterGPU(rows, cols, 11 if name == "time11" else 13, d)
    f.assume_nonneg()
    for _ in range(3):
        f.filter(src, dst)
    zen_amd.synchronize()
    e0, e1 = zen_amd.Event(), zen_amd.Event()
    n, t0 = 0, time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(50):
            f.filter(src, dst)
        n += 50
        zen_amd.synchronize()
    e1.record()
    out[name] = round(8.0 * rows * cols / (1e-3 * e0.elapsed_ms(e1) / n) / 8e12, 4)
print(json.dumps(out), flush=True)
