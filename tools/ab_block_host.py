#!/usr/bin/env python3
"""zen_hip_hpr_process_host (the headline block, pinned host memory in and out) at several piece lengths of its three-stream
pipeline (option "host_block_hops"; 0 = the default, ~8 MiB of input per piece): wall clock per 25 840-hop block and the
link measured with the same buffers.  On the GPU box: python tools/ab_block_host.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zen_amd, bench
zen_amd.init(0)
M = 25840
x = bench.s_music(M * 1024, seed=0)
for piece in (0, 4096, 1024, 512, 256, 3230, 1292):
    zen_amd.set_option("host_block_hops", piece)
    r = bench.block_host_run(zen_amd, x, M, steps=10, warmup=3)
    print(piece, "wall %.3f ms min %.3f  hops/s %.2f M  link up %.3f down %.3f both %.3f" % (r["wall_ms"], r["wall_ms_min"], r["value"] / 1e6, r["link"]["h2d_ms"], r["link"]["d2h_ms"], r["link"]["both_ms"]), flush=True)
