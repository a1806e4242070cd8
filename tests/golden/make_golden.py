#!/usr/bin/env python3
"""Generates the committed fixtures of tests/golden/ (inputs + expected outputs, data only).

  mfilt_stripes.npz   -- the reference's median-filter known-answer vectors, libzen/mfilt.test.cu:315-591:
                         zero matrices with the middle row = 5 and the middle column = 8 (9x9 / f 3,
                         10x20 / f 5, 1024x17 / f 5, 1024x128 / f 5) and the results the reference asserts
                         (time median keeps the column and erases the row, frequency median the converse;
                         with the replicate border everywhere, as its copy-border tests :701-886 assert).
                         Built from that description alone; the oracle is NOT consulted.
  hpr_waveforms.npz   -- frozen outputs of the CPU oracle (oracle/zen_oracle.c) for small seeded inputs on
                         the BASELINE configurations: regression vectors.  The reference has no waveform
                         vectors, so these pin the oracle against drift and give the
                         GPU tests a second, oracle-independent-at-run-time target.

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def stripes(x, y):
    d = np.zeros((x, y), np.float32)
    d[x // 2, :] = 5
    d[:, y // 2] = 8          # assigned last in the reference loop: the crossing holds 8
    return d


def make_stripes():
    out = {}
    for x, y, f in ((9, 9, 3), (10, 20, 5), (1024, 17, 5), (1024, 128, 5)):
        tag = "%dx%d_f%d" % (x, y, f)
        exp_t = np.zeros((x, y), np.float32)
        exp_t[:, y // 2] = 8
        exp_f = np.zeros((x, y), np.float32)
        exp_f[x // 2, :] = 5
        out["in_" + tag] = stripes(x, y)
        out["time_" + tag] = exp_t
        out["freq_" + tag] = exp_f
    np.savez_compressed(os.path.join(HERE, "mfilt_stripes.npz"), **out)


def signal(n, seed, fs=44100.0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    x = 0.25 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1320 * t) + 0.02 * rng.uniform(-1, 1, n)
    x += 0.6 * rng.uniform(-1, 1, n) * ((np.arange(n) % 3000) < 120)
    return x.astype(np.float32)


# (name, fs, hop, beta, flags, causal, sse, soft, n_hops)
HPR_CASES = [
    ("rt_1024_P", 44100.0, 1024, 2.0, 2, True, False, False, 8),         # BASELINE configs[1]
    ("rt_1024_HPR", 44100.0, 1024, 2.0, 7, True, False, False, 8),
    ("rt_256_P_soft", 44100.0, 256, 2.5, 2, True, False, True, 30),
    ("rt_512_sse", 44100.0, 512, 2.0, 2, True, True, False, 16),           # BASELINE configs[4]
    ("ac_256_HPR", 48000.0, 256, 2.0, 7, False, False, False, 40),
    ("ac_4096_HPR", 44100.0, 4096, 2.5, 7, False, False, False, 4),
]
# (name, hop_h, hop_p, beta, soft, n)
HPRI_CASES = [("off_1024_256", 1024, 256, 2.0, False, 20011), ("off_4096_256_soft", 4096, 256, 2.5, True, 30000)]


def make_waveforms():
    from oracle import oracle as o
    out = {}
    for i, (name, fs, hop, beta, flags, causal, sse, soft, n_hops) in enumerate(HPR_CASES):
        x = signal(hop * n_hops, 100 + i, fs)
        h = o.HPR(fs, hop, beta, flags, o.TIME_CAUSAL if causal else o.TIME_ANTICAUSAL)
        if sse:
            h.use_sse_filter()
        if soft:
            h.use_soft_mask()
        res = h.process_stream(x)
        out[name + "_in"] = x
        for k in "PHR":
            out[name + "_" + k] = res[k]
    for i, (name, hop_h, hop_p, beta, soft, n) in enumerate(HPRI_CASES):
        x = signal(n, 200 + i)
        e = o.HPRIOffline(44100.0, hop_h, hop_p, beta, beta)
        if soft:
            e.use_soft_mask()
        hh, pp, rr = e.process(x)
        out[name + "_in"] = x
        out[name + "_H"], out[name + "_P"], out[name + "_R"] = hh, pp, rr
    np.savez_compressed(os.path.join(HERE, "hpr_waveforms.npz"), **out)


if __name__ == "__main__":
    make_stripes()
    make_waveforms()
    for f in ("mfilt_stripes.npz", "hpr_waveforms.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
