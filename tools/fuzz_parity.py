#!/usr/bin/env python3
"""Randomised differential test: HIP engine vs CPU oracle on random configurations (sample rate, hop, output
flags, causality, mask type, blocking, streams).  Test infrastructure; run on the GPU box:
    python tools/fuzz_parity.py --seconds 120 --seed 1          (--offline: the two-pass driver; --resident: the per-hop API;
                                                                 --filters: the drop-in median / box filter classes)
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
    """Returns (ok, mismatches, skipped, refused, distinct configurations).  `skipped`: the ORACLE's constructor threw (a
    configuration the reference rejects or cannot run: l_harm or l_perc rounds to 0, a mask longer than its dimension);
    `refused`: the oracle accepted and the GPU engine did not -- counted as a failure."""
    rng = np.random.default_rng(seed)
    zen_amd.init(0)
    t_end = time.time() + seconds
    n_ok = n_bad = n_skip = n_refused = 0
    seen = set()
    while time.time() < t_end:
        fs = float(rng.choice([2000, 3000, 4000, 8000, 11025, 16000, 22050, 24000, 31000, 32000, 44100, 48000, 64000, 88200, 96000,
                               128000]))
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
        n_hops = int(min(max(2 * h.stft_width + 5, 12), 40 if hop >= 2048 else 400))
        if hop * n_hops * h.stft_width > 3e7:       # keep the oracle (O(W) work per hop) quick
            n_hops = max(6, int(3e7 / (hop * h.stft_width)))
        # (the oracle filters the whole sliding matrix every hop, a sliding insertion per sample: ~W nfft (mt + mf) / 2 per hop)
        per_hop = h.stft_width * 4 * hop * ((h.l_harm | 1) + freq_len) / 2
        if per_hop * n_hops > 3e9:
            n_hops = max(6, int(3e9 / per_hop), (0 if causal else h.l_harm + 6))   # (anticausal: past the lag, or all is zeros)
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
        except zen_amd.ZenHipError as e:              # the oracle accepted the configuration: a refusal is a failure
            print("GPU REFUSED", dict(fs=fs, hop=hop, causal=causal, time_len=time_len, freq_len=freq_len), e, flush=True)
            n_refused += 1
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
        if zen_amd.memcheck()["corrupt_words"]:
            ok = False
        key = (fs, hop, time_len, freq_len, causal, mode)
        seen.add(key)
        if ok:
            n_ok += 1
        else:
            n_bad += 1
            print("MISMATCH", dict(fs=fs, hop=hop, beta=beta, flags=flags, causal=causal, mode=str(mode), streams=streams,
                                   block=block, chunk=chunk, time_len=time_len, freq_len=freq_len), flush=True)
    return n_ok, n_bad, n_skip, n_refused, len(seen)


def run_offline(seconds, seed):
    """The two-pass offline driver (HPRIOffline::process on host vectors) on random clips, hops, sample rates and mask
    types, with the engines' chunk size and the run lengths of the synthesis-in-runs kernels drawn at random too (0 = the
    library's own choice).  Returns (ok, mismatches, skipped, refused, distinct configurations) as run() does."""
    rng = np.random.default_rng(seed)
    zen_amd.init(0)
    t_end = time.time() + seconds
    n_ok = n_bad = n_skip = n_refused = 0
    seen = set()
    opts = ("offline_chunk_hops", "istft_run", "istft_run_wide", "no_istft_runs", "offline_range")
    while time.time() < t_end:
        fs = float(rng.choice([4000, 8000, 16000, 22050, 31000, 32000, 44100, 48000, 96000]))
        hop_p = int(rng.choice([32, 64, 128, 256, 512]))
        hop_h = hop_p * int(rng.choice([1, 2, 4, 8, 16]))
        if hop_h > 4096:
            continue
        beta_h, beta_p = float(rng.choice([1.5, 2.0, 3.0])), float(rng.choice([2.0, 2.5]))
        mode = str(rng.choice(["hard", "hard", "soft", "sse"]))
        n = int(rng.integers(1, 30)) * hop_h + int(rng.integers(0, hop_h))
        try:
            ro = o.HPRIOffline(fs, hop_h, hop_p, beta_h, beta_p)
            eh = o.HPR(fs, hop_h, beta_h, 7, o.TIME_ANTICAUSAL)
            ep = o.HPR(fs, hop_p, beta_p, 1, o.TIME_ANTICAUSAL)
        except Exception:
            n_skip += 1
            continue
        if mode == "soft":
            ro.use_soft_mask()
        if mode == "sse":
            ro.use_sse_filter()
        clips = int(rng.choice([1, 1, 1, 2, 5]))      # > 1: zen_hip_hpri_process_device, one stream per clip
        xs = rng.uniform(-1, 1, (clips, n)).astype(np.float32)
        xs *= (rng.uniform(0, 1, xs.shape) < 0.8)
        x = xs[0]
        ref = ro.process(x)
        refs = [ref]
        for c in range(1, clips):
            rc = o.HPRIOffline(fs, hop_h, hop_p, beta_h, beta_p)
            if mode == "soft":
                rc.use_soft_mask()
            if mode == "sse":
                rc.use_sse_filter()
            refs.append(rc.process(xs[c]))
        cfg = {"offline_chunk_hops": int(rng.choice([0, 0, 1, 3, 8, 24, 100])), "istft_run": int(rng.choice([0, 0, 1, 5, 33])),
               "istft_run_wide": int(rng.choice([0, 0, 1, 4, 9])), "no_istft_runs": int(rng.choice([0, 0, 0, 2])),
               "offline_range": int(rng.choice([0, 0, 4 * hop_h, 9 * hop_h]))}
        for k in opts:
            zen_amd.set_option(k, cfg[k])
        try:
            g = zen_amd.HPRIOffline(fs, hop_h, hop_p, beta_h, beta_p, n_clips=clips)
            if mode == "soft":
                g.use_soft_mask()
            if mode == "sse":
                g.use_sse_filter()
            if clips == 1:
                got = g.process(x)
                ok = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(got, ref))
            else:
                din, dh, dp, dr = (zen_amd.DeviceBuffer(clips * n) for _ in range(4))
                din.upload(xs.reshape(-1))
                g.process_device(din.ptr, n, n, harm=dh.ptr, perc=dp.ptr, resid=dr.ptr, out_stride=n)
                zen_amd.synchronize()
                H, P, R = (b.download().reshape(clips, n) for b in (dh, dp, dr))
                ok = all(np.array_equal(H[c], refs[c][0], equal_nan=True) and np.array_equal(P[c], refs[c][1], equal_nan=True)
                         and np.array_equal(R[c], refs[c][2], equal_nan=True) for c in range(clips))
                for b in (din, dh, dp, dr):
                    b.free()
        except zen_amd.ZenHipError as e:      # the oracle accepted the configuration: a refusal is a failure
            print("GPU REFUSED", dict(fs=fs, hop_h=hop_h, hop_p=hop_p, mode=mode, n=n, clips=clips, **cfg), e, flush=True)
            ok = False
            n_refused += 1
        finally:
            for k in opts:
                zen_amd.set_option(k, 0)
        if zen_amd.memcheck()["corrupt_words"]:
            ok = False
        seen.add((fs, hop_h, hop_p, mode))
        if ok:
            n_ok += 1
        else:
            n_bad += 1
            print("MISMATCH offline", dict(fs=fs, hop_h=hop_h, hop_p=hop_p, beta_h=beta_h, beta_p=beta_p, mode=mode, n=n, clips=clips, **cfg), flush=True)
    return n_ok, n_bad, n_skip, n_refused, len(seen)


def run_resident(seconds, seed):
    """The per-hop API of the reference (HPRRealtime::process_next_hop + copy_* through IOGPU, zen/fakert.h:221-247) with the
    resident kernels (zen_hip_hpr_set_resident): random sample rate, hop, output, mask, idle time-out, pauses longer than it,
    resets, per-launch hops and block calls in between.  Returns (ok, mismatches, skipped, refused, distinct configurations)."""
    rng = np.random.default_rng(seed)
    zen_amd.init(0)
    t_end = time.time() + seconds
    n_ok = n_bad = n_skip = n_refused = 0
    seen = set()
    while time.time() < t_end:
        fs = float(rng.choice([8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000]))
        hop = int(rng.choice([128, 256, 512, 1024, 2048, 4096]))
        beta = float(rng.choice([1.5, 2.0, 2.5, 3.0]))
        key, flag = [("P", o.OUTPUT_PERCUSSIVE), ("H", o.OUTPUT_HARMONIC), ("R", o.OUTPUT_RESIDUAL)][int(rng.integers(0, 3))]
        mode = str(rng.choice(["hard", "hard", "soft", "sse"]))
        if mode == "sse" and key == "R":
            key, flag = "P", o.OUTPUT_PERCUSSIVE                  # (the SSE path has no residual, hps.cu:582-652)
        try:
            ho = o.HPR(fs, hop, beta, flag, o.TIME_CAUSAL)
        except Exception:
            n_skip += 1
            continue
        if mode == "soft":
            ho.use_soft_mask()
        if mode == "sse":
            ho.use_sse_filter()
        n_hops = int(rng.integers(20, 60 if hop <= 1024 else 30))
        x = rng.uniform(-1, 1, hop * n_hops).astype(np.float32)
        x *= (rng.uniform(0, 1, x.shape) < 0.7)
        ref = ho.process_stream(x)[key]
        try:
            io = zen_amd.IOGPU(hop)
            rt = zen_amd.HPRRealtime(fs, hop, beta, flag)
        except Exception as e:
            print("GPU REFUSED", dict(fs=fs, hop=hop), e, flush=True)
            n_refused += 1
            continue
        if mode == "soft":
            rt.use_soft_mask()
        if mode == "sse":
            rt.use_sse_filter()
        copy = {"P": rt.copy_percussive, "H": rt.copy_harmonic, "R": rt.copy_residual}[key]
        eng = rt.p_impl
        idle_ms = int(rng.choice([2, 5, 50, 200]))
        eng.set_resident(idle_ms)
        out = np.zeros_like(x)
        i = 0
        while i < n_hops:
            what = int(rng.integers(0, 12))
            if what == 0 and i + 3 <= n_hops:                    # a block call in mid-stream (the kernel goes home first)
                blk = eng.process_stream_host(x[i * hop:(i + 3) * hop])[key]
                out[i * hop:(i + 3) * hop] = blk
                i += 3
                continue
            if what == 1:
                eng.set_resident(0 if rng.integers(0, 2) else idle_ms)   # per-launch hops for a while, or back
            if what == 2:
                time.sleep(idle_ms / 1e3 * 2.5)                   # longer than the idle time-out: the kernel leaves
            io.host_in[:] = x[i * hop:(i + 1) * hop]
            rt.process_next_hop(io.device_in)
            copy(io.device_out)
            out[i * hop:(i + 1) * hop] = io.host_out
            i += 1
        seen.add((fs, hop, key, mode))
        if np.array_equal(out, ref, equal_nan=True):
            n_ok += 1
        else:
            n_bad += 1
            print("MISMATCH resident", dict(fs=fs, hop=hop, beta=beta, out=key, mode=mode, idle_ms=idle_ms, n_hops=n_hops), flush=True)
        del rt, eng, io
    return n_ok, n_bad, n_skip, n_refused, len(seen)


def run_filters(seconds, seed):
    """The drop-in filter classes (MedianFilterGPU / BoxFilterGPU: libzen/mfilt.h:33-268, box.h:30-215) on random matrices:
    any shape, ANY filter_len the reference accepts (1 .. the filtered dimension, mfilt.h:296-305), both directions, data with
    ties / ramps / signs, with and without the non-negativity promise.  Returns (ok, mismatches, rejected, refused, distinct)."""
    rng = np.random.default_rng(seed)
    zen_amd.init(0)
    t_end = time.time() + seconds
    n_ok = n_bad = n_skip = n_refused = 0
    seen = set()
    while time.time() < t_end:
        kind = str(rng.choice(["median", "median", "median", "box"]))
        direction = str(rng.choice(["f", "t"]))
        shape = str(rng.choice(["small", "wide", "tall", "aligned"]))
        if shape == "small":
            rows, cols = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        elif shape == "wide":
            rows, cols = int(rng.integers(1, 12)), int(rng.integers(40, 6000))
        elif shape == "tall":
            rows, cols = int(rng.integers(40, 3000)), int(rng.integers(1, 24))
        else:
            rows, cols = int(rng.integers(1, 40)), int(rng.choice([256, 512, 1024, 2048, 4096, 8192]))
        dim = cols if direction == "f" else rows
        pick = str(rng.choice(["short", "any", "near", "special"]))
        if pick == "short":
            flen = int(rng.integers(1, min(dim, 64) + 1))
        elif pick == "any":
            flen = int(rng.integers(1, dim + 1))
        elif pick == "near":
            flen = max(1, dim - int(rng.integers(0, 3)))
        else:
            flen = min(dim, int(rng.choice([47, 63, 64, 65, 93, 187, 255, 256, 257, 258, 511, 1024, 2047, 2048, 2049, 4097])))
        if rows * cols * flen > 6e8:                      # (the oracle: a sliding insertion per sample)
            continue
        data = str(rng.choice(["uniform", "ties", "ramp", "nonneg", "nonneg"]))
        if data == "uniform":
            a = rng.uniform(-1, 1, (rows, cols)).astype(np.float32)
        elif data == "ties":
            a = rng.integers(-3, 4, (rows, cols)).astype(np.float32)
        elif data == "ramp":
            a = (np.arange(rows * cols, dtype=np.float32).reshape(rows, cols) * (1 if rng.integers(0, 2) else -1)).astype(np.float32)
        else:
            a = rng.random((rows, cols), dtype=np.float32)
            a *= (rng.uniform(0, 1, a.shape) < 0.8)
        promise = data == "nonneg" and kind == "median" and bool(rng.integers(0, 2))
        od = o.FREQUENCY if direction == "f" else (o.TIME_CAUSAL if rng.integers(0, 2) else o.TIME_ANTICAUSAL)
        zd = zen_amd.FREQUENCY if direction == "f" else (zen_amd.TIME_CAUSAL if od == o.TIME_CAUSAL else zen_amd.TIME_ANTICAUSAL)
        try:
            ref = (o.median_filter if kind == "median" else o.box_filter)(a, flen, od)
        except Exception:
            n_skip += 1
            continue
        try:
            f = (zen_amd.MedianFilterGPU if kind == "median" else zen_amd.BoxFilterGPU)(rows, cols, flen, zd)
            if promise:
                f.assume_nonneg()
            got = f.filter_host(a)
        except zen_amd.ZenHipError as e:
            print("GPU REFUSED", dict(kind=kind, rows=rows, cols=cols, flen=flen, direction=direction), e, flush=True)
            n_refused += 1
            continue
        ok = np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and not zen_amd.memcheck()["corrupt_words"]
        seen.add((kind, direction, flen | 1, cols if direction == "f" else rows))
        if ok:
            n_ok += 1
        else:
            n_bad += 1
            print("MISMATCH filter", dict(kind=kind, rows=rows, cols=cols, flen=flen, direction=direction, data=data, promise=promise), flush=True)
    return n_ok, n_bad, n_skip, n_refused, len(seen)


def memcheck_line():
    """Red zones (ZEN_HIP_REDZONE, set by tools/fuzz_final.sh) of everything still alive, plus what the frees found."""
    r = zen_amd.memcheck()
    if not r["redzone_bytes"]:
        return "memcheck off (no ZEN_HIP_REDZONE)", 0
    bad = r["corrupt_words"] + r["bounds_violations"]
    return ("memcheck clean (%d allocations, %d-byte red zones)" % (r["allocations"], r["redzone_bytes"]) if not bad
            else "MEMCHECK: " + r["first_message"]), bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--offline", action="store_true", help="fuzz the two-pass offline driver instead of the streaming engine")
    ap.add_argument("--resident", action="store_true", help="fuzz the per-hop API with the resident kernels (zen_hip_hpr_set_resident)")
    ap.add_argument("--filters", action="store_true", help="fuzz the drop-in MedianFilterGPU / BoxFilterGPU classes (any shape, any length)")
    args = ap.parse_args()
    if args.filters:
        n_ok, n_bad, n_skip, n_ref, n_seen = run_filters(args.seconds, args.seed)
        mc, bad = memcheck_line()
        print("filters: ok %d  mismatches %d  oracle-rejected %d  GPU-refused %d  distinct (kind, direction, taps, dimension): %d  %s"
              % (n_ok, n_bad, n_skip, n_ref, n_seen, mc))
        return 1 if n_bad or bad or n_ref else 0
    if args.resident:
        n_ok, n_bad, n_skip, n_ref, n_seen = run_resident(args.seconds, args.seed)
        mc, bad = memcheck_line()
        print("resident: ok %d  mismatches %d  oracle-rejected %d  GPU-refused %d  distinct (fs, hop, output, mode): %d  %s"
              % (n_ok, n_bad, n_skip, n_ref, n_seen, mc))
        return 1 if n_bad or bad or n_ref else 0
    if args.offline:
        n_ok, n_bad, n_skip, n_ref, n_seen = run_offline(args.seconds, args.seed)
        mc, bad = memcheck_line()
        print("offline: ok %d  mismatches %d  oracle-rejected %d  GPU-refused %d  distinct (fs, hop_h, hop_p, mode): %d  %s"
              % (n_ok, n_bad, n_skip, n_ref, n_seen, mc))
        return 1 if n_bad or bad or n_ref else 0
    n_ok, n_bad, n_skip, n_ref, n_seen = run(args.seconds, args.seed)
    mc, bad = memcheck_line()
    print("ok %d  mismatches %d  oracle-rejected %d  GPU-refused %d  distinct (fs, hop, masks, causality, mode): %d  %s"
          % (n_ok, n_bad, n_skip, n_ref, n_seen, mc))
    return 1 if n_bad or bad or n_ref else 0


if __name__ == "__main__":
    sys.exit(main())
