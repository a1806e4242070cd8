#!/usr/bin/env python3
"""ab_sse_block.py -- the SSE block leg of bench.py (BASELINE configs[4]: hop 512, P, nocopybord, 51 680 hops per step) alone, for
A/B runs of variant builds:  ZEN_HIP_SO=zen_amd/libzen_hip_<variant>.so python tools/ab_sse_block.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import zen_amd  # noqa: E402

zen_amd.init(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
hop5, M5 = 512, 2 * 25840
x5 = np.stack([bench.s_music(M5 * hop5, seed=0)])
solo = bench._Solo()
for r in range(reps):
    run = bench.block_run(zen_amd, solo, x5, zen_amd.OUTPUT_PERCUSSIVE, M5, 10, 3, 300.0, zen_amd.synchronize, hop=hop5, sse=True, copy_bord=False)
    per = {k: round(v["ms"] / max(v["launches"], 1), 4) for k, v in run["breakdown"].items() if v["launches"]}
    print("%s  %.2f M hops/s  %.4f ms per step  %s" % (os.environ.get("ZEN_HIP_SO", "shipped"), M5 * 10 / run["dt"] / 1e6, 1e3 * run["dt"] / 10, per), flush=True)
    bench.free_run(run)
