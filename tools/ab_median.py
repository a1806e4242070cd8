#!/usr/bin/env python3
"""A/B of median-kernel variants on the GPU box, interleaved so that clock and box drift hit both alike:
    python tools/ab_median.py --shapes "25840,4096,3,t;..." --a "" --b "median_time_variant=1" [--nonneg-b] [--rounds 4]
For every shape: `rounds` x (0.3 s of back-to-back launches with the options of A, then of B); prints the mean launch time and
fraction of 8 TB/s (8 B per element) per variant.  Options are zen_hip_set_option names ("name=value,name=value")."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zen_amd  # noqa: E402


def set_opts(spec, on):
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        zen_amd.set_option(k, int(v) if on else 0)


def burst(f, src, dst, seconds):
    e0, e1 = zen_amd.Event(), zen_amd.Event()
    n, t0 = 0, time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            f.filter(src, dst)
        n += 50
        zen_amd.synchronize()
    e1.record()
    return e0.elapsed_ms(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", required=True)
    ap.add_argument("--a", default="")
    ap.add_argument("--b", default="")
    ap.add_argument("--nonneg-a", action="store_true")
    ap.add_argument("--nonneg-b", action="store_true")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=0.3)
    args = ap.parse_args()
    zen_amd.init(0)
    for sh in filter(None, args.shapes.split(";")):
        rows, cols, flen, d = sh.split(",")
        rows, cols, flen = int(rows), int(cols), int(flen)
        rng = np.random.default_rng(flen)
        src = zen_amd.DeviceBuffer.from_host(rng.random((rows, cols), dtype=np.float32))
        dst = zen_amd.DeviceBuffer(rows * cols)
        fa = zen_amd.MedianFilterGPU(rows, cols, flen, zen_amd.FREQUENCY if d == "f" else zen_amd.TIME_ANTICAUSAL)
        fb = zen_amd.MedianFilterGPU(rows, cols, flen, zen_amd.FREQUENCY if d == "f" else zen_amd.TIME_ANTICAUSAL)
        if args.nonneg_a:
            fa.assume_nonneg()
        if args.nonneg_b:
            fb.assume_nonneg()
        ta, tb = [], []
        for _ in range(args.rounds):
            set_opts(args.a, True)
            ta.append(burst(fa, src, dst, args.seconds))
            set_opts(args.a, False)
            set_opts(args.b, True)
            tb.append(burst(fb, src, dst, args.seconds))
            set_opts(args.b, False)
        fr = lambda ms: 8.0 * rows * cols / (1e-3 * ms) / 8e12
        print(json.dumps({"shape": sh, "a_ms": round(float(np.mean(ta)), 5), "b_ms": round(float(np.mean(tb)), 5), "a_frac": round(fr(np.mean(ta)), 4),
                          "b_frac": round(fr(np.mean(tb)), 4), "a_each": [round(fr(x), 3) for x in ta], "b_each": [round(fr(x), 3) for x in tb]}), flush=True)
        del fa, fb
        src.free()
        dst.free()


if __name__ == "__main__":
    main()
