#!/usr/bin/env python3
"""bench.py -- hops/sec of the 1024-hop HPR hot path on MI355X, with the HBM roofline of its kernels.

Metric (BASELINE.json): "hops/sec (1024-hop HPR, 44.1 kHz mono) + median-filter HBM GB/s vs roofline".

Default workload (BASELINE configs[1], the one `value` is quoted on): HPRRealtime<GPU> semantics -- hop
1024 (nwin 2048, transform size 4096), beta 2.0, OUTPUT_PERCUSSIVE, hard mask, causal -- on a synthetic
44.1 kHz mono stream (S-music of BASELINE.md) already resident in HBM.  One "step" pushes the next
`--hops` hops (default 25 840 = 10 minutes of audio) of the stream through zen_hip_hpr_process with the
percussive output written (STFT -> 47-tap frequency median -> hard mask -> iSTFT -> overlap-add).  State
carries over between steps exactly as consecutive process_next_hop calls leave it, and the samples are
bit-identical to per-hop calls (tests/test_gpu_parity.py::test_hpr_blocking_is_invisible,
::test_block_fused_matches_three_kernel_path).

N > 1: a realtime stream is a sequential recurrence and does not shard ("replicas only"):
every rank runs its own independent stream of the same size (weak scaling), no data-path collective;
torch.distributed (RCCL) carries the barrier and the max-over-ranks time only.

Also in the JSON line:
  roofline        -- the dominant kernel of the timed region, timed with HIP events on the engine's stream
                     (zen_hip_hpr_profile).  Default path: rt_fused_kernel (one workgroup per hop), priced
                     with SURVEY 8(d)'s per-frame minimum 24*(nfft/2+1) + 8*hop bytes per hop.  With
                     --no-block-fused: the 47-tap frequency median kernel, 8 B/element.
  roofline_median -- BASELINE's second metric: the stand-alone frequency-direction median kernel (47 taps
                     over the 25 840 x 4096 magnitude matrix, 8 B/element: 4 read + 4 written), timed
                     with HIP events in a second leg that sends the same stream through the STFT /
                     median / iSTFT kernels (outside the timed region of `value`).
  cpu_baseline    -- the CPU oracle (restatement of the reference's CPU/IPP path, kind "port") timed on
                     this box's host, one thread, on a bounded prefix of the same stream; rank 0, N = 1.
  realtime        -- the single-stream per-hop call path (process_next_hop + copy_percussive through
                     mapped host memory, timed like zen/fakert.h:221-247), outside the timed region.

Other workloads (not the headline; `--workload`):
  offline_batch -- BASELINE configs[3]: independent 30 s mono clips, HPRIOffline<GPU> 4096/256 hard mask,
                   clips sharded over the ranks (64 per GPU by default), two passes resident in HBM.
  offline_long  -- BASELINE configs[2]: one 10-minute stereo clip, soft mask; with N ranks each channel is
                   cut into N time ranges computed independently (strong scaling).
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
# HBM bytes per launch measured with rocprofv3 --pmc (tools/pmc_cmd.sh); re-collected whenever a kernel changes
FUSED_TRAFFIC_FILE = "r02_fused_hbm_traffic.json"
MEDIAN_TRAFFIC_FILE = "r02_median47_hbm_traffic.json"


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


def usable_cores():
    """Hardware threads this process may really use: the smaller of the affinity mask and the cgroup CPU quota
    (a container on a 256-thread host may be allowed a few CPUs' worth of time only)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def host_cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_realtime(x, budget_s=12.0):
    """Oracle HPR<CPU> (hop 1024, P only, causal) on a prefix of x."""
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
    return {"value": n / dt, "unit": "hops/s", "cores": 1, "kind": "port",
            "sample": "first %d hops (%.1f s of audio) of the same S-music stream, oracle/zen_oracle.c "
                      "HPR<CPU> hop 1024 P-only causal, 1 thread" % (n, n * HOP / FS),
            "ms_per_hop": 1e3 * dt / n, "host_cpu": host_cpu_name(), "host_cores_available": os.cpu_count()}


def cpu_baseline_offline(x, hop_h, hop_p, total_hops_per_clip, seconds=6.0, beta=BETA, soft=False):
    """Oracle HPRIOffline on a prefix of one clip (what zen/offline.h:141-147 times), one thread."""
    from oracle import oracle as o
    n = int(min(x.size, seconds * FS))
    eng = o.HPRIOffline(FS, hop_h, hop_p, beta, beta)
    if soft:
        eng.use_soft_mask()
    t0 = time.perf_counter()
    eng.process(x[:n])
    dt = time.perf_counter() - t0
    n1, _ = o.chunk_padder(n, hop_h, 1)
    n2, _ = o.chunk_padder(n, hop_p, 11)
    return {"value": (n1 + n2) / dt, "unit": "hops/s", "cores": 1, "kind": "port",
            "sample": "first %.1f s of clip 0, oracle HPRIOffline %d/%d %s mask, 1 thread"
                      % (n / FS, hop_h, hop_p, "soft" if soft else "hard"),
            "x_realtime": (n / FS) / dt, "host_cpu": host_cpu_name(), "host_cores_available": os.cpu_count()}


def _cpu_clip_proc(clip_id, seconds, hop_h, hop_p, barrier, q):
    """One host core: synthesise the clip, wait for every other core, then time one oracle HPRIOffline run
    (forked worker, CPU only)."""
    from oracle import oracle as o
    n = int(seconds * FS)
    x = s_music(n, seed=7000 + clip_id)
    eng = o.HPRIOffline(FS, hop_h, hop_p, BETA, BETA)
    eng.process(x[:int(0.3 * FS)])           # page in the library and the engine's buffers
    barrier.wait()
    t0 = time.perf_counter()
    eng.process(x)
    t1 = time.perf_counter()
    n1, _ = o.chunk_padder(n, hop_h, 1)
    n2, _ = o.chunk_padder(n, hop_p, 11)
    q.put((n1 + n2, t0, t1))


def cpu_baseline_offline_all_cores(hop_h, hop_p, seconds=4.0):
    """one clip per host core, all cores at once.  Must run BEFORE the GPU is
    initialised: the workers are plain forks of this process.  Input synthesis and start-up are outside the
    timed region: every worker prepares its clip, all meet at a barrier, and the wall time is taken from the
    first start to the last finish of the process() calls (time.perf_counter is system-wide on Linux)."""
    import multiprocessing as mp
    cores = usable_cores()
    ctx = mp.get_context("fork")
    barrier, q = ctx.Barrier(cores), ctx.Queue()
    procs = [ctx.Process(target=_cpu_clip_proc, args=(i, seconds, hop_h, hop_p, barrier, q)) for i in range(cores)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    hops = sum(r[0] for r in res)
    wall = max(r[2] for r in res) - min(r[1] for r in res)
    return {"value": hops / wall, "unit": "hops/s", "cores": cores, "kind": "port",
            "sample": "first %.1f s of %d clips, one per core at the same time, oracle HPRIOffline %d/%d hard mask; "
                      "input synthesis outside the timed region" % (seconds, cores, hop_h, hop_p),
            "x_realtime": cores * seconds / wall, "wall_s": wall, "host_cpu": host_cpu_name(),
            "host_threads_visible": os.cpu_count(), "host_threads_usable": cores}


def offline_rooflines(prof, steps, frames, nfft, hop, n_out1, copy_bw, freq_mask):
    """Per-kernel HBM rooflines of the two offline passes from the engines' HIP-event timings.

    Algorithmic bytes per frame (SURVEY 8(d) accounting: compulsory traffic of each
    kernel as a stand-alone stage, half spectrum nfft/2+1 bins where the data is Hermitian):
      stft        : 4*hop in + 8*(nfft/2+1) spectrum + 4*(nfft/2+1) magnitude out
      freq_filter : 8 B per element (4 read + 4 written) of the frames x nfft matrix; where the engine filters
                    half rows of the frames x (nfft/2 + 1 + mask/2) it needs
      time_filter : 8 B per element of frames x nfft, or of frames x (nfft/2 + 1) with half rows
      istft       : per output 8*(nfft/2+1) spectrum + 8*(nfft/2+1) H and P in + 4*nwin out
      finalize    : per output 12*hop (two half frames in, one hop out)"""
    out = {}
    for ps in ("pass1", "pass2"):
        N, h, F = nfft[ps], hop[ps], frames[ps]
        nout = n_out1 if ps == "pass1" else 1
        mf = freq_mask[ps]
        half = mf <= 63 or mf in (65, 85, 93, 129, 171, 187, 255)  # hpr.hip run_chunk: half rows on the median path
        per_frame = {"stft": 4 * h + 12 * (N // 2 + 1),
                     "freq_filter": 8 * (N // 2 + 1 + mf // 2) if half else 8 * N,
                     "time_filter": 8 * (N // 2 + 1) if half else 8 * N,
                     "istft": nout * (16 * (N // 2 + 1) + 8 * h), "finalize": nout * 12 * h}
        for k, v in prof[ps].items():
            if not v["launches"] or k not in per_frame:
                continue
            ms = v["ms"] / steps
            ach = per_frame[k] * F / (ms * 1e-3) / 1e9
            out["%s.%s" % (ps, k)] = {"ms_per_step": ms, "launches_per_step": v["launches"] / steps,
                                      "algorithmic_bytes_per_frame": per_frame[k], "frames_per_step": F,
                                      "achieved": ach, "frac": ach / 8000.0}
    dom = max(out, key=lambda k: out[k]["ms_per_step"])
    d = out[dom]
    roof = {"bound": "hbm", "achieved": d["achieved"], "peak": 8000.0, "unit": "GB/s", "frac": d["frac"],
            "traffic": None, "kernel": dom, "avg_launch_ms": d["ms_per_step"] / max(d["launches_per_step"], 1),
            "algorithmic_bytes_per_frame": d["algorithmic_bytes_per_frame"], "frames_per_step": d["frames_per_step"],
            "device_copy_GBps": copy_bw, "frac_of_device_copy": d["achieved"] / copy_bw,
            "share_of_kernel_time": d["ms_per_step"] / sum(v["ms_per_step"] for v in out.values()),
            "note": "dominant kernel of the step by HIP-event time; every kernel's line is in `kernels`"}
    return roof, out


def device_copy_bandwidth(zen_amd, n_floats=1 << 28, iters=10):
    """Device-to-device copy of 1 GiB (read + write bytes per second): the practical HBM roof of this box,
    quoted next to the nominal 8 TB/s (roofline denominators)."""
    import ctypes as C
    a, b = zen_amd.DeviceBuffer(n_floats), zen_amd.DeviceBuffer(n_floats)
    a.zero()
    lib = zen_amd.load()

    def cp():
        lib.zen_hip_memcpy_d2d(C.c_void_p(b.ptr), C.c_void_p(a.ptr), C.c_size_t(4 * n_floats), None)

    for _ in range(3):
        cp()
    zen_amd.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        cp()
    zen_amd.synchronize()
    dt = (time.perf_counter() - t0) / iters
    a.free()
    b.free()
    return 8.0 * n_floats / dt / 1e9


def realtime_leg(zen_amd, x, n_hops=400):
    """Per-hop call path through mapped memory, timed like zen/fakert.h:221-247: this interpreter's loop, and
    the same loop without an interpreter (tools/rt_latency.cpp through the C-ABI) for hops 256...4096."""
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
        rt.copy_percussive(io.device_out)            # returns when the hop is in host_out
        _ = io.host_out[0]
    dt = time.perf_counter() - t0
    res = {"us_per_hop_python_loop": 1e6 * dt / n_hops, "launches_per_hop": 1,
           "note": "process_next_hop + copy_percussive via IOGPU buffers, host-timed incl. the host copies "
                   "(zen/fakert.h:221-247); one launch per hop, the overlap-add happens in the kernel and copy_* "
                   "polls a sequence word behind the finished hop; latency-bound, no roofline quoted"}
    # the same loop in C++ through the C-ABI (no interpreter between the calls), hops 256...4096 + SSE
    try:
        import subprocess
        import tempfile
        exe = os.path.join(tempfile.gettempdir(), "zen_rt_latency_%d" % os.getpid())
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tools", "rt_latency.cpp"), "-o", exe, "-L",
                               os.path.join(ROOT, "zen_amd"), "-lzen_hip", "-Wl,-rpath," + os.path.join(ROOT, "zen_amd")],
                              stderr=subprocess.DEVNULL)
        lines = subprocess.run([exe, "2000"], capture_output=True, text=True, timeout=120).stdout.splitlines()
        os.remove(exe)
        sweep = [json.loads(ln) for ln in lines if ln.startswith("{")]
        res["per_hop_us_by_hop"] = {("sse_" if r["sse"] else "") + str(r["hop"]): r["us_per_hop"] for r in sweep}
        at = [r for r in sweep if r["hop"] == HOP and not r["sse"]]
        if at:
            res["us_per_hop"] = at[0]["us_per_hop"]
            res["hops_per_s"] = 1e6 / at[0]["us_per_hop"]
            res["x_realtime"] = (1e6 / at[0]["us_per_hop"]) * HOP / FS
            res["timed_by"] = "tools/rt_latency.cpp (C++ loop through the C-ABI, 2000 hops after 200 warm-up hops)"
    except Exception as exc:                         # no compiler on the box: keep the interpreter's figure
        res["cpp_loop_error"] = str(exc)[:200]
    if "us_per_hop" not in res:
        res["us_per_hop"] = res["us_per_hop_python_loop"]
        res["hops_per_s"] = 1e6 / res["us_per_hop"]
        res["x_realtime"] = res["hops_per_s"] * HOP / FS
        res["timed_by"] = "this interpreter's loop"
    return res


def dry_main(args, zdist):
    """The N-rank plumbing without a GPU: rendezvous, sharding of the workload's units, empty timed steps
    between barriers, max-over-ranks time, summed counters, one JSON line from rank 0.  What tests/ run on
    CPU with gloo; never a measurement."""
    rank, _, world = zdist.env_world()
    grp = zdist.Group(args.backend if args.backend == "gloo" else "gloo")
    if args.workload == "offline_batch":
        units = len(zdist.shard_units(args.clips * world, world, rank))
        par = "clips sharded x%d, no data-path collective" % world
    elif args.workload == "offline_long":
        b, e = zdist.time_shards(int(600 * FS), world, 4096)[rank]
        units = (e - b) // 4096
        par = "each channel time-sharded x%d with warm-up halos, no exchange" % world
    else:
        units = args.hops * args.streams
        par = "replicas x%d" % world
    grp.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    grp.barrier()
    dt = grp.max(time.perf_counter() - t0)
    tot_units, ranks = grp.sum([units, 1])
    if rank == 0:
        print(json.dumps({"metric": "dry run (launch / sharding plumbing only)", "dry": True, "value": None,
                          "unit": "hops/s", "n_gpus": world, "ranks_reported": int(ranks), "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1),
                          "higher_is_better": True, "scaling": "strong" if args.workload == "offline_long" else "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "none",
                          "config": {"workload": args.workload, "units_all_ranks": int(tot_units),
                                     "units_rank0": units, "parallelism": par, "backend": "gloo"}}))
    grp.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="realtime_block", choices=["realtime_block", "offline_batch", "offline_long"])
    ap.add_argument("--hops", type=int, default=25840, help="hops per step per stream (25840 = 10 min)")
    ap.add_argument("--streams", type=int, default=1, help="independent streams per GPU")
    ap.add_argument("--clips", type=int, default=64, help="offline_batch: clips per GPU")
    ap.add_argument("--clip-seconds", type=float, default=30.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-realtime", action="store_true")
    ap.add_argument("--outputs", default="P", choices=["P", "HPR"],
                    help="realtime_block: percussive only (the headline config) or all three outputs")
    ap.add_argument("--no-block-fused", action="store_true",
                    help="realtime_block: STFT / median / iSTFT kernels instead of the fused per-hop kernel")
    ap.add_argument("--fused-minb", type=int, default=0, help="tuning: occupancy the fused kernel is built for")
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed steps run for this long before the W warm-up steps, so that the clocks have "
                         "settled when the timed region starts (the timed region is still exactly K steps)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL over xGMI; gloo for the CPU plumbing test)")
    ap.add_argument("--dry", action="store_true",
                    help="no GPU: run only the launch / sharding / aggregation plumbing with empty steps "
                         "(CPU tests; the line says dry: true and is not a measurement)")
    args = ap.parse_args()

    from zen_amd import dist as zdist
    if args.gpus > 1 and not zdist.launched_by_torchrun():
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above has imported torch, loaded the
        # HIP library or touched a GPU; the ranks are fresh interpreters (one per GPU), rank 0's line is relayed.
        sys.exit(zdist.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    rank, local_rank, world = zdist.env_world()
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d from the launcher; using WORLD_SIZE" % (args.gpus, world),
              file=sys.stderr)
    if args.dry:
        return dry_main(args, zdist)

    cpu_all = None
    if args.workload == "offline_batch" and world == 1 and not args.no_cpu_baseline:
        cpu_all = cpu_baseline_offline_all_cores(4096, 256)      # forks: before anything loads or touches the GPU
    import torch
    if os.environ.get("ZEN_ALLOW_GPU_SHARING") and torch.cuda.device_count() > 0:
        # testing the N-rank path on a box with fewer GPUs than ranks (use --backend gloo: RCCL refuses two ranks
        # on one device); never a measurement configuration
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    grp = zdist.Group(args.backend, torch.device("cuda", local_rank) if args.backend == "nccl" else None)

    import zen_amd
    zen_amd.init(local_rank)
    if args.no_block_fused:
        zen_amd.set_option("no_block_fused", 1)
    if args.fused_minb:
        zen_amd.set_option("block_fused_minb", args.fused_minb)

    def barrier():
        torch.cuda.synchronize()
        grp.barrier()
        torch.cuda.synchronize()

    def settle(step_fn, ms):
        t_end = time.perf_counter() + ms / 1e3
        while time.perf_counter() < t_end:
            for _ in range(5):
                step_fn()
            zen_amd.synchronize()

    out = {"metric": "hops/sec (1024-hop HPR, 44.1 kHz mono)", "unit": "hops/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic"}

    if args.workload == "realtime_block":
        M, S = args.hops, args.streams
        n = M * HOP
        x = np.stack([s_music(n, seed=1000 * rank + s) for s in range(S)])
        d_in = zen_amd.DeviceBuffer.from_host(x)
        d_out = zen_amd.DeviceBuffer(S * n)
        all_out = args.outputs == "HPR"
        flags = (zen_amd.OUTPUT_PERCUSSIVE | zen_amd.OUTPUT_HARMONIC | zen_amd.OUTPUT_RESIDUAL) if all_out \
            else zen_amd.OUTPUT_PERCUSSIVE
        d_h = zen_amd.DeviceBuffer(S * n) if all_out else None
        d_r = zen_amd.DeviceBuffer(S * n) if all_out else None
        eng = zen_amd.HPR(FS, HOP, BETA, flags, zen_amd.TIME_CAUSAL, True, S, M)

        def step():
            eng.process(d_in.ptr, M, n, d_h.ptr if all_out else None, d_out.ptr, d_r.ptr if all_out else None, n)

        settle(step, args.settle_ms)
        for _ in range(args.warmup):
            step()
        barrier()
        eng.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = grp.max(time.perf_counter() - t0)
        med_ms, med_launches, med_elems = eng.profile_get()
        breakdown = eng.profile_get_all()
        eng.profile(False)
        chk = grp.sum([float(np.abs(d_out.download(4096)).sum())])[0]   # liveness only; bytes, not data path
        fused = breakdown["rt_fused"]["launches"] > 0
        three = None
        if fused and rank == 0:
            # second leg, outside the timed region: the same stream through the general engine (STFT / median /
            # iSTFT kernels) for the stand-alone median kernel's roofline, BASELINE's second metric
            zen_amd.set_option("no_block_fused", 1)

            def leg():
                for _ in range(3):
                    step()
                zen_amd.synchronize()
                eng.profile(True)
                t1 = time.perf_counter()
                for _ in range(10):
                    step()
                zen_amd.synchronize()
                dt3 = (time.perf_counter() - t1) / 10
                r = eng.profile_get()
                kern = {k: v["ms"] / 10 for k, v in eng.profile_get_all().items() if v["launches"]}
                eng.profile(False)
                return dt3, r, kern

            # the path as the engine runs it (half rows: bins 0..nfft/2 and the last 23 of every magnitude / P row)
            dt3, _, kern3 = leg()
            three = {"ms_per_step": 1e3 * dt3, "hops_per_s": S * M / dt3, "kernel_ms_per_step": kern3,
                     "rows": "half (bins 0..2048 and 4073..4095 filtered)"}
            # BASELINE's median metric is the kernel over the whole 25 840 x 4096 matrix: whole rows
            zen_amd.set_option("no_half_rows", 1)
            dt3f, (med_ms, med_launches, med_elems), kern3f = leg()
            three["whole_rows"] = {"ms_per_step": 1e3 * dt3f, "hops_per_s": S * M / dt3f, "kernel_ms_per_step": kern3f}
            zen_amd.set_option("no_half_rows", 0)
            zen_amd.set_option("no_block_fused", 0)
        if rank == 0:
            total_hops = world * S * M * args.steps
            value = total_hops / dt
            nfft = 4 * HOP

            def traffic_of(fname, kernel, elems):
                """HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this
                process); only quoted for the shape it was measured on."""
                try:
                    tj = json.load(open(os.path.join(ROOT, "profiles", fname)))
                    if tj["shape"]["elements"] == elems:
                        return tj["kernels"][kernel]["hbm_bytes_per_launch"]
                except (OSError, KeyError, ValueError):
                    pass
                return None

            copy_bw = device_copy_bandwidth(zen_amd)
            t_med = 1e-3 * med_ms / max(med_launches, 1)
            el_med = med_elems // max(med_launches, 1)
            ach_med = 8.0 * el_med / t_med / 1e9 if t_med > 0 else 0.0
            tr_med = traffic_of(MEDIAN_TRAFFIC_FILE, "median47_dpp_kernel<nonneg>", el_med)
            roof_median = {
                "bound": "hbm", "achieved": ach_med, "peak": 8000.0, "unit": "GB/s", "frac": ach_med / 8000.0,
                "device_copy_GBps": copy_bw, "frac_of_device_copy": ach_med / copy_bw,
                "traffic": tr_med,
                "traffic_source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in passes of their own, FETCH "
                                  "doubled per the gfx950 correction)" % MEDIAN_TRAFFIC_FILE if tr_med else None,
                "kernel": "median47_dpp_kernel<nonneg> (frequency direction, 47 taps, whole 4096-bin rows)",
                "elements_per_launch": el_med, "algorithmic_bytes_per_element": 8, "avg_launch_ms": 1e3 * t_med,
                "launches": med_launches,
                "note": "stand-alone kernel of the three-kernel path (second leg, outside the timed region of `value`); "
                        "its data movement alone (same loads, LDS image, transposed stores, no sorting) takes 0.145 ms, "
                        "the nontemporal copy of the same bytes 0.139 ms (profiles/r02_median47_variants.txt, "
                        "profiles/r02_ubench_copy.txt)"}
            if fused:
                fl = breakdown["rt_fused"]
                t_f = 1e-3 * fl["ms"] / fl["launches"]
                bytes_per_hop = 24 * (nfft // 2 + 1) + 8 * HOP       # per-frame minimum, P-only hard mask
                ach = bytes_per_hop * S * M / t_f / 1e9
                tr = traffic_of(FUSED_TRAFFIC_FILE, "rt_fused_kernel<12,47>", S * M * nfft)
                moved = 4 * HOP + 8 * HOP                              # what the kernel itself must move per hop
                roof = {
                    "bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                    "limiter": "valu-issue",
                    "device_copy_GBps": copy_bw, "frac_of_device_copy": ach / copy_bw,
                    "traffic": tr,
                    "traffic_source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in passes of their own, "
                                      "FETCH doubled per the gfx950 correction)" % FUSED_TRAFFIC_FILE if tr else None,
                    "kernel": "rt_fused_kernel<12, 47> (one workgroup per hop: STFT, |S|, 47-tap median, hard mask, iSTFT)",
                    "hops_per_launch": S * M, "algorithmic_bytes_per_hop": bytes_per_hop,
                    "algorithmic_bytes_formula": "24*(nfft/2+1) + 8*hop (SURVEY 8(d): per-frame minimum of the batched "
                                                 "pipeline, percussive-only hard mask)",
                    "hbm_bytes_moved_per_hop_by_design": moved,
                    "moved_GBps": moved * S * M / t_f / 1e9,
                    "avg_launch_ms": 1e3 * t_f, "launches": fl["launches"],
                    "share_of_step": (fl["ms"] / 1e3) / dt if dt > 0 else None,
                    "note": "`achieved`/`frac` price the launch with SURVEY 8(d)'s ALGORITHMIC bytes as the bench contract "
                            "asks; the kernel does not move them: the spectrum, |S| and P stay in registers/LDS, its own HBM "
                            "traffic is 4*hop read + 8*hop written per hop (`moved_GBps`, `traffic`), far from the HBM roof. "
                            "Its limiter is VALU issue (FFT butterflies at full rate; median min/max, DPP moves, double-"
                            "precision |z| at half rate): see DESIGN.md section 5 and profiles/r02_*fused*"}
            else:
                roof = dict(roof_median, share_of_step=(med_ms / 1e3) / dt if dt > 0 else None)
            out.update({
                "value": value, "ms_per_step": 1e3 * dt / args.steps,
                "config": {
                    "workload": "HPRRealtime<GPU> semantics: hop 1024, nwin 2048, transform 4096, beta 2.0, "
                                "%s, hard mask, causal; S-music 44.1 kHz mono stream resident "
                                "in HBM; block mode (zen_hip_hpr_process), %d hops/step/stream"
                                % ("OUTPUT_HARMONIC|PERCUSSIVE|RESIDUAL" if all_out else "OUTPUT_PERCUSSIVE", M),
                    "hops_per_step": M, "streams_per_gpu": S, "fs": FS, "hop": HOP,
                    "time_mask": 3, "freq_mask": 47, "parallelism": "replicas x%d" % world,
                    "path": "fused per-hop kernel + overlap-add" if fused else "STFT / median / iSTFT kernels + overlap-add"},
                "x_realtime": value * HOP / FS,
                "checksum": chk,
                "kernel_ms_per_step": {k: v["ms"] / args.steps for k, v in breakdown.items() if v["launches"]},
                "roofline": roof})
            if fused:
                out["roofline_median"] = roof_median
                out["three_kernel_path"] = three
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_realtime(x[0])
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            if world == 1 and not args.no_realtime:
                out["realtime"] = realtime_leg(zen_amd, x[0])
    elif args.workload == "offline_batch":
        C = args.clips
        n = int(args.clip_seconds * FS)
        hop_h, hop_p = 4096, 256
        ids = zdist.shard_units(C * world, world, rank)          # clip ids of this rank (C each)
        x = np.stack([s_music(n, seed=7000 + i) for i in ids])
        d_in = zen_amd.DeviceBuffer.from_host(x)
        d_h, d_p = zen_amd.DeviceBuffer(C * n), zen_amd.DeviceBuffer(C * n)
        eng = zen_amd.HPRIOffline(FS, hop_h, hop_p, BETA, BETA, False, C)
        n1, n2 = eng.hop_counts(n)

        def step():
            eng.process_device(d_in.ptr, n, n, d_h.ptr, d_p.ptr, None, n)

        settle(step, args.settle_ms)
        for _ in range(args.warmup):
            step()
        barrier()
        eng.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = grp.max(time.perf_counter() - t0)
        prof = eng.profile_get_all()
        eng.profile(False)
        chk = grp.sum([float(np.abs(d_p.download(4096)).sum())])[0]
        if rank == 0:
            total_hops = world * C * (n1 + n2) * args.steps
            value = total_hops / dt
            roof, kern = offline_rooflines(prof, args.steps, {"pass1": C * n1, "pass2": C * n2},
                                           {"pass1": 4 * hop_h, "pass2": 4 * hop_p}, {"pass1": hop_h, "pass2": hop_p},
                                           3, device_copy_bandwidth(zen_amd), {"pass1": 187, "pass2": 13})
            out.update({"roofline": roof, "kernels": kern})
            out.update({
                "metric": "hops/sec (HPR-I offline, hops of both passes)", "value": value,
                "ms_per_step": 1e3 * dt / args.steps,
                "config": {
                    "workload": "HPRIOffline<GPU> 4096/256 beta 2.0 hard mask, %d x %.0f s mono S-music clips per "
                                "GPU resident in HBM, both passes + harmonic/percussive outputs" % (C, args.clip_seconds),
                    "clips_per_gpu": C, "clip_samples": n, "hops_pass1": n1, "hops_pass2": n2,
                    "parallelism": "clips sharded x%d, no data-path collective" % world},
                "x_realtime": world * C * args.clip_seconds * args.steps / dt,
                "checksum": chk})
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_offline(x[0], hop_h, hop_p, n1 + n2)
                out["cpu_baseline_all_cores"] = cpu_all
    if args.workload == "offline_long":
        # BASELINE configs[2]: one 10-minute stereo clip = 2 mono channels, HPR-I 4096/256, soft mask p = 2.
        # N > 1: every channel is cut into N time ranges, rank r computes range r of both
        # channels from its own halo of input; strong scaling, no exchange.
        n = int(600 * FS)
        hop_h, hop_p = 4096, 256
        b, e = zdist.time_shards(n, world, hop_h)[rank]
        chans = [s_music(n, seed=9000 + c) for c in range(2)]
        d_in = [zen_amd.DeviceBuffer.from_host(c) for c in chans]
        d_h, d_p = zen_amd.DeviceBuffer(max(e - b, 1)), zen_amd.DeviceBuffer(max(e - b, 1))
        eng = zen_amd.HPRIOffline(FS, hop_h, hop_p, 2.5, 2.5, False, 1)
        eng.use_soft_mask()
        n1, n2 = eng.hop_counts(n)

        def step():
            for c in range(2):
                if e > b:
                    eng.process_range(d_in[c].ptr, n, b, e, d_h.ptr, d_p.ptr)

        settle(step, args.settle_ms)
        for _ in range(args.warmup):
            step()
        barrier()
        eng.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = grp.max(time.perf_counter() - t0)
        prof = eng.profile_get_all()
        eng.profile(False)
        chk = grp.sum([float(np.abs(d_p.download(min(4096, max(e - b, 1)))).sum())])[0]
        if rank == 0:
            if world == 1:
                roof, kern = offline_rooflines(prof, args.steps, {"pass1": 2 * n1, "pass2": 2 * n2},
                                               {"pass1": 4 * hop_h, "pass2": 4 * hop_p},
                                               {"pass1": hop_h, "pass2": hop_p}, 2, device_copy_bandwidth(zen_amd),
                                               {"pass1": 187, "pass2": 13})
                out.update({"roofline": roof, "kernels": kern})
                if not args.no_cpu_baseline:
                    out["cpu_baseline"] = cpu_baseline_offline(chans[0], hop_h, hop_p, n1 + n2, seconds=4.0, beta=2.5,
                                                               soft=True)
            out.update({
                "metric": "hops/sec (HPR-I offline, hops of both passes)", "scaling": "strong",
                "value": 2 * (n1 + n2) * args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
                "config": {"workload": "HPRIOffline<GPU> 4096/256 beta 2.5 soft mask (p = 2), one 10-minute stereo "
                                       "clip (2 mono channels of 26 460 000 samples) resident in HBM",
                           "hops_pass1": n1, "hops_pass2": n2,
                           "parallelism": "each channel time-sharded x%d with warm-up halos, no exchange" % world},
                "x_realtime": 600.0 * args.steps / dt, "checksum": chk})
    if rank == 0:
        print(json.dumps(out))
    grp.close()


if __name__ == "__main__":
    main()
