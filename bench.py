#!/usr/bin/env python3
"""bench.py -- hops/sec of the 1024-hop HPR hot path on MI355X, with the median kernel's HBM roofline.

Metric (BASELINE.json): "hops/sec (1024-hop HPR, 44.1 kHz mono) + median-filter HBM GB/s vs roofline".

Workload at N = 1 (BASELINE configs[1]): HPRRealtime<GPU> semantics -- hop 1024 (nwin 2048, transform
size 4096), beta 2.0, OUTPUT_PERCUSSIVE, hard mask, causal -- on a synthetic 44.1 kHz mono stream
(S-music of BASELINE.md) that is already resident in HBM.  One "step" pushes the next `--hops` hops
(default 25 840 = 10 minutes of audio) of the stream through zen_hip_hpr_process, percussive output
included (STFT -> frequency median -> mask -> iSTFT -> overlap-add); state carries over between steps,
exactly as consecutive process_next_hop calls would leave it, and the samples are bit-identical to
per-hop calls (tests/test_gpu_parity.py::test_hpr_blocking_is_invisible).

N > 1: the realtime stream is a sequential recurrence and does not shard ("replicas only"):
every rank runs its own independent stream of the same size (weak scaling), no data-path collective;
torch.distributed (RCCL) is used for the barrier and the max-over-ranks time only.

Also in the JSON line:
  roofline     -- the frequency-direction median kernel (47 taps over the 25 840 x 4096 magnitude
                  matrix), algorithmic bytes = 8 B/element (4 read + 4 written) divided by
                  its mean launch duration, measured with HIP events on the engine's stream inside the
                  timed region (zen_hip_hpr_profile).
  cpu_baseline -- the CPU oracle (restatement of the reference's CPU/IPP path, "port") timed on this
                  box's host, one thread, on a bounded prefix of the same stream; rank 0, N = 1 only.
  realtime     -- the single-stream per-hop call path (process_next_hop + copy_percussive through mapped
                  host memory, as zen/fakert.h:221-247 times it), outside the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FS = 44100.0
HOP = 1024
BETA = 2.0


def s_music(n, seed=0, fs=FS):
    """BASELINE.md S-music: 4 sines + decaying noise clicks every 0.25 s + 0.01 noise (float32)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / fs
    x = sum(0.2 * np.sin(2 * np.pi * f * t) for f in (220.0, 440.0, 660.0, 1320.0))
    step = int(0.25 * fs)
    env = np.exp(-np.arange(int(0.005 * fs)) / (0.001 * fs))
    for s in range(0, n, step):
        m = min(env.size, n - s)
        x[s:s + m] += 0.9 * env[:m] * rng.uniform(-1, 1, m)
    return (x + 0.01 * rng.uniform(-1, 1, n)).astype(np.float32)


def cpu_baseline(x, budget_s=12.0):
    """Oracle HPR<CPU> (hop 1024, P only, causal) on a prefix of x; returns dict for the JSON line."""
    from oracle import oracle as o
    h = o.HPR(FS, HOP, BETA, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL)
    probe = 40
    t0 = time.perf_counter()
    for i in range(probe):
        h.process_next_hop(x[i * HOP:(i + 1) * HOP])
    per_hop = (time.perf_counter() - t0) / probe
    n = int(max(100, min(x.size // HOP - probe, budget_s / per_hop)))
    h.reset_buffers()
    t0 = time.perf_counter()
    for i in range(n):
        h.process_next_hop(x[i * HOP:(i + 1) * HOP])
        _ = h.percussive_out            # copy_percussive
    dt = time.perf_counter() - t0
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n / dt, "unit": "hops/s", "cores": 1, "kind": "port",
            "sample": "first %d hops (%.1f s of audio) of the same S-music stream, oracle/zen_oracle.c "
                      "HPR<CPU> hop 1024 P-only causal, 1 thread" % (n, n * HOP / FS),
            "ms_per_hop": 1e3 * dt / n, "host_cpu": cpu, "host_cores_available": os.cpu_count()}


def realtime_leg(zen_amd, x, n_hops=400):
    """Per-hop call path through mapped memory; returns dict."""
    rt = zen_amd.HPRRealtime(FS, HOP, BETA, zen_amd.OUTPUT_PERCUSSIVE, False, 1)
    io = zen_amd.IOGPU(HOP)
    for i in range(50):                              # warm-up
        io.host_in[:] = x[i * HOP:(i + 1) * HOP]
        rt.process_next_hop(io.device_in)
        rt.copy_percussive(io.device_out)
    t0 = time.perf_counter()
    for i in range(n_hops):
        io.host_in[:] = x[i * HOP:(i + 1) * HOP]
        rt.process_next_hop(io.device_in)
        rt.copy_percussive(io.device_out)            # synchronises
        _ = io.host_out[0]
    dt = time.perf_counter() - t0
    return {"us_per_hop": 1e6 * dt / n_hops, "hops_per_s": n_hops / dt, "launches_per_hop": 4,
            "note": "process_next_hop + copy_percussive via mapped host memory, host-timed incl. copies "
                    "(zen/fakert.h:221-247); latency-bound, no roofline quoted"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--hops", type=int, default=25840, help="hops per step per stream (25840 = 10 min)")
    ap.add_argument("--streams", type=int, default=1, help="independent streams per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-realtime", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    import zen_amd
    zen_amd.init(local_rank)

    M, S = args.hops, args.streams
    n = M * HOP
    x = np.stack([s_music(n, seed=1000 * rank + s) for s in range(S)])
    d_in = zen_amd.DeviceBuffer.from_host(x)
    d_out = zen_amd.DeviceBuffer(S * n)
    eng = zen_amd.HPR(FS, HOP, BETA, zen_amd.OUTPUT_PERCUSSIVE, zen_amd.TIME_CAUSAL, True, S, M)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        eng.process(d_in.ptr, M, n, None, d_out.ptr, None, n)

    for _ in range(args.warmup):
        step()
    barrier()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    med_ms, med_launches, med_elems = eng.profile_get()
    eng.profile(False)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        chk = torch.tensor([float(np.abs(d_out.download(4096)).sum())], dtype=torch.float64, device="cuda")
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)   # bytes-sized: a liveness checksum, not data path

    if rank == 0:
        total_hops = world * S * M * args.steps
        value = total_hops / dt
        bytes_per_launch = 8.0 * med_elems / max(med_launches, 1)
        t_launch = 1e-3 * med_ms / max(med_launches, 1)
        achieved = bytes_per_launch / t_launch / 1e9 if t_launch > 0 else 0.0
        out = {
            "metric": "hops/sec (1024-hop HPR, 44.1 kHz mono)",
            "value": value,
            "unit": "hops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "HPRRealtime<GPU> semantics: hop 1024, nwin 2048, transform 4096, beta 2.0, "
                            "OUTPUT_PERCUSSIVE, hard mask, causal; S-music 44.1 kHz mono stream resident "
                            "in HBM; block mode (zen_hip_hpr_process), %d hops/step/stream" % M,
                "hops_per_step": M, "streams_per_gpu": S, "fs": FS, "hop": HOP,
                "time_mask": 3, "freq_mask": 47, "parallelism": "replicas x%d" % world,
            },
            "x_realtime": value * HOP / FS,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": None,
                "kernel": "median_wave_kernel<1,0> (frequency direction, 47 taps)",
                "elements_per_launch": med_elems / max(med_launches, 1),
                "algorithmic_bytes_per_element": 8, "avg_launch_ms": 1e3 * t_launch,
                "launches": med_launches,
                "share_of_step": (med_ms / 1e3) / dt if dt > 0 else None,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(x[0])
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if world == 1 and not args.no_realtime:
            out["realtime"] = realtime_leg(zen_amd, x[0])
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
