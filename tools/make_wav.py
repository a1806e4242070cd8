#!/usr/bin/env python3
"""Writes a synthetic PCM16 WAV (S-music of BASELINE.md) for trying the `zen` CLI: make_wav.py out.wav [seconds] [channels]"""
import struct
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bench import s_music  # noqa: E402

path = sys.argv[1]
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.6637
ch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fs = 44100
n = int(seconds * fs)
x = np.stack([s_music(n, seed=c) for c in range(ch)], axis=1).reshape(-1)
pcm = np.round(np.clip(x, -1, 1) * 20000).astype("<i2").tobytes()
with open(path, "wb") as f:
    f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " +
            struct.pack("<IHHIIHH", 16, 1, ch, fs, fs * 2 * ch, 2 * ch, 16) + b"data" + struct.pack("<I", len(pcm)) + pcm)
