#!/usr/bin/env python3
"""Micro-benchmark of the standalone median kernel (zen_hip_mfilt_run) on the reference's bench shapes
(libzen/mfilt.bench.cu: dim x dim, filter 11) and on the BASELINE path shapes.  Prints one JSON line per
case: achieved algorithmic GB/s (8 B per element) and fraction of the 8 TB/s HBM roof."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zen_amd  # noqa: E402


def run(rows, cols, flen, direction, iters, general=False, nonneg=False):
    zen_amd.set_option("median_general", int(general))
    rng = np.random.default_rng(0)
    d = rng.uniform(0, 1, (rows, cols)).astype(np.float32)
    if os.environ.get("ZEN_BENCH_SIGNED"):
        d -= 0.5
    src, dst = zen_amd.DeviceBuffer.from_host(d), zen_amd.DeviceBuffer(d.size)
    f = zen_amd.MedianFilterGPU(rows, cols, flen, direction)
    if nonneg:
        f.assume_nonneg()
    for _ in range(3):
        f.filter(src, dst)
    zen_amd.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f.filter(src, dst)
    zen_amd.synchronize()
    dt = (time.perf_counter() - t0) / iters
    gbs = 8.0 * rows * cols / dt / 1e9
    return {"rows": rows, "cols": cols, "filter_len": flen,
            "direction": "frequency" if direction == zen_amd.FREQUENCY else "time",
            "kernel": "general" if general else "auto", "assume_nonneg": bool(nonneg),
            "data": "signed" if os.environ.get("ZEN_BENCH_SIGNED") else "non-negative", "ms": 1e3 * dt, "GBps": gbs, "frac_of_8TBps": gbs / 8000.0}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--suite", default="path", choices=["path", "squares", "one", "sustained"])
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--opt", action="append", default=[], help="zen_hip_set_option name=value (A/B switches), repeatable")
    ap.add_argument("--shapes", default="", help='--suite sustained: "rows,cols,taps,t|f;..." instead of bench.py\'s list')
    ap.add_argument("--rows", type=int, default=25840)
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--len", type=int, default=47)
    ap.add_argument("--dir", default="frequency")
    ap.add_argument("--nonneg", action="store_true",
                    help="zen_hip_mfilt_assume_nonneg on the handle: the input is a magnitude matrix (>= +0), ordering keys are "
                         "the raw bits -- the kernel builds the engine launches (the 47-tap kernel on 4096-bin rows finds "
                         "that out by itself)")
    args = ap.parse_args()
    zen_amd.init(0)
    for opt in ("median47_blocks", "median47_shared"):
        if os.environ.get("ZEN_" + opt.upper()):
            zen_amd.set_option(opt, int(os.environ["ZEN_" + opt.upper()]))
    for kv in args.opt:
        zen_amd.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    F, Tm = zen_amd.FREQUENCY, zen_amd.TIME_ANTICAUSAL
    if args.suite == "sustained":   # bench.py's protocol (>= 1 s of back-to-back launches per shape), every listed shape
        import bench
        shapes = bench.MEDIAN_SHAPES
        if args.shapes:
            shapes = [(int(a), int(b), int(c), d) for a, b, c, d in (x.split(",") for x in args.shapes.split(";") if x)]
        for rec in bench.median_shapes(zen_amd, args.seconds, shapes=shapes, nonneg=args.nonneg):
            print(json.dumps(rec), flush=True)
        sys.exit(0)
    if args.suite == "one":
        cases = [(args.rows, args.cols, args.len, F if args.dir == "frequency" else Tm)]
    elif args.suite == "path":   # SURVEY 8(d) path shapes
        cases = [(25840, 4096, 47, F), (25840, 4096, 3, Tm), (103360, 1024, 13, F), (103360, 1024, 11, Tm),
                 (51680, 2048, 23, F), (51680, 2048, 7, Tm), (6460, 16384, 187, F), (12920, 8192, 93, F)]
    else:                        # libzen/mfilt.bench.cu:222-262
        cases = [(1 << k, 1 << k, 11, d) for k in range(5, 15) for d in (F, Tm)]
    for rows, cols, flen, direction in cases:
        print(json.dumps(run(rows, cols, flen, direction, args.iters, nonneg=args.nonneg)), flush=True)
