#!/usr/bin/env python3
"""Shape sweeps of the reference's three bench harnesses, one JSON line per case:

  mfilt -- libzen/mfilt.bench.cu:222-262: dim x dim matrices, dim = 2^5 .. 2^14, filter 11, iota data, both
           directions; device-resident, and the "MEM" variants through mapped host memory (dim <= 2^12).
  fft   -- libzen/fftw.bench.cu:231-282: one C2C transform of n = 2^8 .. 2^15 (the reference's whole sweep),
           forward, inverse, round trip; per-call latency and batched throughput.
  hpr   -- libzen/hps.bench.cu:62-64: HPRRealtime<GPU>(48000, hop, 2.0, P) for hop = 2^5 .. 2^12: per-hop
           call path through mapped memory (the figure docs/cpu_vs_gpu.png plots), and block-mode hops/s.
  sse   -- BASELINE configs[4]: the SSE (box-filter) variant at hop 512 and 2048, per-hop and block mode.
  copy  -- device-to-device copy bandwidth of the box: the practical HBM roof next to the nominal 8 TB/s.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zen_amd  # noqa: E402


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    zen_amd.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    zen_amd.synchronize()
    return (time.perf_counter() - t0) / iters


def suite_mfilt():
    for k in range(5, 15):
        dim = 1 << k
        d = np.arange(dim * dim, dtype=np.float32).reshape(dim, dim)      # iota (mfilt.bench.cu:17-32)
        src, dst = zen_amd.DeviceBuffer.from_host(d), zen_amd.DeviceBuffer(d.size)
        for name, direction in (("time", zen_amd.TIME_ANTICAUSAL), ("frequency", zen_amd.FREQUENCY)):
            f = zen_amd.MedianFilterGPU(dim, dim, 11, direction)
            dt = timed(lambda: f.filter(src, dst), 50 if k < 12 else 10)
            yield {"suite": "mfilt", "dim": dim, "filter_len": 11, "direction": name, "memory": "device",
                   "us": 1e6 * dt, "GBps": 8.0 * dim * dim / dt / 1e9}
            if k <= 12:                                                     # "MEM": mapped host memory both ways
                io = zen_amd.IOGPU(dim * dim)
                io.host_in[:] = d.ravel()
                dt = timed(lambda: f.filter(io.device_in, io.device_out), 20 if k < 11 else 5)
                yield {"suite": "mfilt", "dim": dim, "filter_len": 11, "direction": name, "memory": "mapped host",
                       "us": 1e6 * dt, "GBps": 8.0 * dim * dim / dt / 1e9}
        src.free()
        dst.free()


def suite_fft():
    rng = np.random.default_rng(0)
    for k in range(8, 16):
        n = 1 << k
        f = zen_amd.FFTC2CWrapperGPU(n)
        f.fft_vec.upload((rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64))
        fwd = timed(f.forward, 200)
        inv = timed(f.backward, 200)

        def rt():
            f.forward()
            f.backward()

        both = timed(rt, 200)
        batch = max(1, (1 << 26) // n)
        buf = zen_amd.DeviceBuffer(batch * n, np.complex64)
        buf.zero()
        bt = timed(lambda: f.exec_batched(buf.ptr, batch), 10)
        yield {"suite": "fft", "n": n, "forward_us": 1e6 * fwd, "inverse_us": 1e6 * inv, "round_trip_us": 1e6 * both,
               "batched_transforms": batch, "batched_ms": 1e3 * bt, "batched_GBps": 16.0 * batch * n / bt / 1e9,
               "batched_GFLOPs": 5.0 * n * k * batch / bt / 1e9}
        buf.free()


def hpr_rows(fs):
    """libzen/hps.bench.cu:62-64 (hop 2^5 .. 2^12 at `fs`), and -- round 6 -- the corners of the engine's domain beside it: the
    lowest rate each long hop still runs at (fs / hop = 7.8125: a 257-tap frequency mask) and the shortest hop at the
    highest rates (96 kHz: a 400-row sliding matrix, 128 kHz: 534 rows).  (fs, hop) pairs; none may be refused."""
    rows = [(fs, 1 << k) for k in range(5, 13)]
    rows += [(32000.0, 4096), (16000.0, 2048), (8000.0, 1024), (22050.0, 2048), (16000.0, 1024), (96000.0, 32), (128000.0, 32)]
    return rows


def suite_hpr(fs):
    rng = np.random.default_rng(1)
    for fs, hop in hpr_rows(fs):
        n_hops = 300 if fs / hop < 2000 else 60
        x = rng.uniform(-1, 1, hop * n_hops).astype(np.float32)
        rt = zen_amd.HPRRealtime(fs, hop, 2.0, zen_amd.OUTPUT_PERCUSSIVE)
        io = zen_amd.IOGPU(hop)
        for i in range(30):
            io.host_in[:] = x[i * hop:(i + 1) * hop]
            rt.process_next_hop(io.device_in)
            rt.copy_percussive(io.device_out)
        t0 = time.perf_counter()
        for i in range(n_hops):
            io.host_in[:] = x[i * hop:(i + 1) * hop]
            rt.process_next_hop(io.device_in)
            rt.copy_percussive(io.device_out)
            _ = io.host_out[0]
        per_hop = (time.perf_counter() - t0) / n_hops
        p = rt.p_impl
        M = max(64, (1 << 25) // (4 * hop))                                 # block mode, resident input
        if p.freq_len > 255 or p.stft_width > 255:
            M = min(M, 2048)                                                # (the general kernels: slow, see DESIGN)
        xb = rng.uniform(-1, 1, hop * M).astype(np.float32)
        eng = zen_amd.HPR(fs, hop, 2.0, zen_amd.OUTPUT_PERCUSSIVE, zen_amd.TIME_CAUSAL, True, 1, M)
        din, dout = zen_amd.DeviceBuffer.from_host(xb), zen_amd.DeviceBuffer(xb.size)
        bt = timed(lambda: eng.process(din.ptr, M, xb.size, None, dout.ptr, None, xb.size), 10)
        yield {"suite": "hpr", "fs": fs, "hop": hop, "nfft": 4 * hop, "time_mask": p.time_len, "freq_mask": p.freq_len,
               "per_hop_us": 1e6 * per_hop, "hop_period_us": 1e6 * hop / fs, "pct_of_hop_period": 100 * per_hop * fs / hop,
               "block_hops": M, "block_hops_per_s": M / bt, "block_x_realtime": (M / bt) * hop / fs}
        din.free()
        dout.free()


def suite_sse(fs=44100.0):
    """BASELINE configs[4]: HPRRealtime<GPU>(44100, 512, 2.0, P, nocopybord) + use_sse_filter(): box filters
    7 / 23 taps instead of medians; also hop 2048 as the config's second reading."""
    rng = np.random.default_rng(2)
    for hop in (512, 2048):
        n_hops = 300
        x = rng.uniform(-1, 1, hop * n_hops).astype(np.float32)
        rt = zen_amd.HPRRealtime(fs, hop, 2.0, zen_amd.OUTPUT_PERCUSSIVE, True)
        rt.use_sse_filter()
        io = zen_amd.IOGPU(hop)
        for i in range(30):
            io.host_in[:] = x[i * hop:(i + 1) * hop]
            rt.process_next_hop(io.device_in)
            rt.copy_percussive(io.device_out)
        t0 = time.perf_counter()
        for i in range(n_hops):
            io.host_in[:] = x[i * hop:(i + 1) * hop]
            rt.process_next_hop(io.device_in)
            rt.copy_percussive(io.device_out)
            _ = io.host_out[0]
        per_hop = (time.perf_counter() - t0) / n_hops
        M = (1 << 25) // (4 * hop)
        xb = rng.uniform(-1, 1, hop * M).astype(np.float32)
        eng = zen_amd.HPR(fs, hop, 2.0, zen_amd.OUTPUT_PERCUSSIVE, zen_amd.TIME_CAUSAL, False, 1, M)
        eng.use_sse_filter()
        din, dout = zen_amd.DeviceBuffer.from_host(xb), zen_amd.DeviceBuffer(xb.size)
        bt = timed(lambda: eng.process(din.ptr, M, xb.size, None, dout.ptr, None, xb.size), 10)
        yield {"suite": "sse", "fs": fs, "hop": hop, "nfft": 4 * hop, "time_box": rt.p_impl.time_len,
               "freq_box": rt.p_impl.freq_len, "per_hop_us": 1e6 * per_hop, "block_hops": M,
               "block_hops_per_s": M / bt, "block_x_realtime": (M / bt) * hop / fs}
        din.free()
        dout.free()


def suite_copy():
    n = 1 << 28                                                             # 1 GiB of floats each way
    a, b = zen_amd.DeviceBuffer(n), zen_amd.DeviceBuffer(n)
    a.zero()
    lib = zen_amd.load()
    dt = timed(lambda: lib.zen_hip_memcpy_d2d(C.c_void_p(b.ptr), C.c_void_p(a.ptr), C.c_size_t(4 * n), None), 10)
    yield {"suite": "copy", "bytes": 4 * n, "ms": 1e3 * dt, "GBps_read_plus_write": 8.0 * n / dt / 1e9}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--suite", default="all", choices=["all", "mfilt", "fft", "hpr", "sse", "copy"])
    ap.add_argument("--fs", type=float, default=48000.0)
    args = ap.parse_args()
    zen_amd.init(0)
    gens = {"copy": suite_copy, "mfilt": suite_mfilt, "fft": suite_fft, "hpr": lambda: suite_hpr(args.fs), "sse": suite_sse}
    for name, g in gens.items():
        if args.suite in ("all", name):
            for rec in g():
                print(json.dumps(rec), flush=True)
