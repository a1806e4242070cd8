#!/usr/bin/env python3
"""Randomised differential test: HIP engine vs CPU oracle on random configurations (sample rate, hop, output
flags, causality, mask type, blocking, streams).  Test infrastructure; run on the GPU box:
    python tools/fuzz_parity.py --seconds 120 --seed 1
Prints every mismatch with its configuration and exits non-zero if there was one."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zen_amd  # noqa: E402
from oracle import oracle as o  # noqa: E402


def run(seconds, seed):
    """Returns (ok, mismatches, skipped, distinct configurations)."""
    rng = np.random.default_rng(seed)
    zen_amd.init(0)
    t_end = time.time() + seconds
    n_ok = n_bad = n_skip = 0
    seen = set()
    while time.time() < t_end:
        fs = float(rng.choice([2000, 3000, 4000, 8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000]))
        hop = int(rng.choice([32, 64, 128, 256, 512, 1024, 2048, 4096]))
        beta = float(rng.choice([1.5, 2.0, 2.5, 3.0]))
        flags = int(rng.integers(1, 8))
        causal = bool(rng.integers(0, 2))
        mode = rng.choice(["hard", "soft", "sse"])
        streams = int(rng.choice([1, 1, 2, 3]))
        try:
            h = o.HPR(fs, hop, beta, flags, o.TIME_CAUSAL if causal else o.TIME_ANTICAUSAL)
        except Exception:
            n_skip += 1
            continue
        time_len = h.l_harm | 1 if not causal else h.stft_width | 1     # odd mask lengths (mfilt.h:89)
        freq_len = h.l_perc | 1
        if freq_len > 255 or time_len > 255:
            n_skip += 1
            continue
        n_hops = int(min(max(2 * h.stft_width + 5, 12), 40 if hop >= 2048 else 400))
        if hop * n_hops * h.stft_width > 3e7:       # keep the oracle (O(W) work per hop) quick
            n_hops = max(6, int(3e7 / (hop * h.stft_width)))
        x = rng.uniform(-1, 1, (streams, hop * n_hops)).astype(np.float32)
        x *= (rng.uniform(0, 1, x.shape) < 0.7)       # some exact zeros / ties
        refs = []
        for s in range(streams):
            hh = o.HPR(fs, hop, beta, flags, o.TIME_CAUSAL if causal else o.TIME_ANTICAUSAL)
            if mode == "soft":
                hh.use_soft_mask()
            if mode == "sse":
                hh.use_sse_filter()
            refs.append(hh.process_stream(x[s]))
        block = int(rng.choice([1, 3, 7, n_hops]))
        chunk = int(rng.choice([0, 4, 16]))
        try:
            g = zen_amd.HPR(fs, hop, beta, flags, zen_amd.TIME_CAUSAL if causal else zen_amd.TIME_ANTICAUSAL, True,
                            streams, chunk)
        except zen_amd.ZenHipError as e:
            print("GPU refused", fs, hop, e)
            n_skip += 1
            continue
        if mode == "soft":
            g.use_soft_mask()
        if mode == "sse":
            g.use_sse_filter()
        got = g.process_stream_host(x if streams > 1 else x[0], block=block)
        ok = True
        for s in range(streams):
            for k in "PHR":
                a = got[k][s] if streams > 1 else got[k]
                if not np.array_equal(a, refs[s][k], equal_nan=True):
                    ok = False
        key = (fs, hop, time_len, freq_len, causal, mode)
        seen.add(key)
        if ok:
            n_ok += 1
        else:
            n_bad += 1
            print("MISMATCH", dict(fs=fs, hop=hop, beta=beta, flags=flags, causal=causal, mode=str(mode), streams=streams,
                                   block=block, chunk=chunk, time_len=time_len, freq_len=freq_len), flush=True)
    return n_ok, n_bad, n_skip, len(seen)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    n_ok, n_bad, n_skip, n_seen = run(args.seconds, args.seed)
    print("ok %d  mismatches %d  skipped %d  distinct (fs, hop, masks, causality, mode): %d" % (n_ok, n_bad, n_skip, n_seen))
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
