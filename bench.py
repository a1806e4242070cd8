#!/usr/bin/env python3
"""bench.py -- hops/sec of the 1024-hop HPR hot path on MI355X, with the HBM roofline of its kernels.

Metric (BASELINE.json): "hops/sec (1024-hop HPR, 44.1 kHz mono) + median-filter HBM GB/s vs roofline".

`value` (BASELINE configs[1]): HPRRealtime<GPU> semantics -- hop 1024 (nwin 2048, transform size 4096), beta 2.0,
OUTPUT_PERCUSSIVE, hard mask, causal -- on a synthetic 44.1 kHz mono stream (S-music of BASELINE.md) already
resident in HBM.  One "step" pushes the next `--hops` hops (default 25 840 = 10 minutes of audio) of the stream
through zen_hip_hpr_process (the BLOCK form of the reference's process_next_hop: an MI355X extension of the API;
the per-hop reference API is timed in `realtime`) with the percussive output written.  State carries over between
steps exactly as consecutive process_next_hop calls leave it, and the samples are bit-identical to per-hop calls and
to the oracle (tests/test_gpu_round3.py::test_headline_block_windows_vs_oracle checks this very block).

Everything else in the JSON line is measured OUTSIDE the timed region of `value`, each leg between its own
synchronisation points:
  roofline        -- the dominant kernel of the timed region, timed with HIP events on the engine's stream
                     (zen_hip_hpr_profile): rt_fused_kernel (one workgroup per hop), priced with SURVEY 8(d)'s
                     per-frame minimum 24*(nfft/2+1) + 8*hop bytes per hop.
  roofline_median -- BASELINE's second metric: the stand-alone frequency-direction median kernel (47 taps over
                     whole rows of the 25 840 x 4096 magnitude matrix, 8 B/element).  `burst`: 10 launches inside
                     the three-kernel path (input freshly written: Infinity-Cache help); `sustained`: >= 1 s of
                     back-to-back launches; `cold`: a 512 MiB write between launches.  frac = sustained.
  all_outputs     -- the same stream with H + P + R written (the second case SURVEY 8(d) names for config 2)
  s_noise         -- the headline configuration on the S-noise seed
  offline_batch   -- BASELINE configs[3] per GPU: 64 x 30 s clips, HPRIOffline<GPU> 4096/256, hard masks
  offline_long    -- BASELINE configs[2]: the 10-minute stereo clip, soft mask p = 2
  sse_block       -- BASELINE configs[4]: SSE path, hop 512, nocopybord, blocks of hops
  cpu_baseline    -- the CPU oracle (kind "port") on this box's host, one thread, bounded prefix; rank 0, N = 1
  realtime        -- the reference API: process_next_hop + copy_percussive per hop through mapped host memory,
                     timed like zen/fakert.h:221-247

N > 1 (`--gpus N`, one process per GPU): a realtime stream is a sequential recurrence and does not shard
("replicas only"): `value` is N independent streams (weak scaling).  The path north_star scales --
the batched offline HPR-I, clips sharded over the ranks with no data-path collective -- is timed in the same run
and reported as `offline_batch_sharded` next to it; `ranks_reported` is an all-reduce of ones.

`--workload offline_batch | offline_long` run those as the main workload with the full step count.
"""
import argparse
import json
import os
import re
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
TRAFFIC_FILE = "r06_hbm_traffic.json"   # {demangled kernel name: {elements, rows, cols, hbm_bytes_per_launch, ...}}
K_FUSED_P = "rt_fused_kernel<12, 47, 3, true, true, true>"
K_FUSED_HPR = "rt_fused_kernel<12, 47, 3, false, true, true>"
K_MEDIAN_WHOLE = "median47_dpp_kernel<false, 0, false>"   # through plain zen_hip_mfilt_run: the build that checks sign bits
K_MEDIAN_WHOLE_ENGINE = "median47_dpp_kernel<true, 0, false>"   # the engine's own launches on whole rows (burst leg)
K_MEDIAN_HALF = "median47_dpp_kernel<true, 0, true>"


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


def s_noise(n, seed=0):
    """BASELINE.md S-noise: i.i.d. uniform(-1, 1), the reference tests' distribution (hps.test.cu:24-36)."""
    return np.random.default_rng(seed).uniform(-1, 1, n).astype(np.float32)


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
    from oracle import ipp_probe                      # "if the box has ipp.h / libipp*, say so"
    return {"value": n / dt, "unit": "hops/s", "cores": 1, "kind": "port",
            "sample": "first %d hops (%.1f s of audio) of the same S-music stream, oracle/zen_oracle.c "
                      "HPR<CPU> hop 1024 P-only causal, 1 thread" % (n, n * HOP / FS),
            "ms_per_hop": 1e3 * dt / n, "host_cpu": host_cpu_name(), "host_cores_available": os.cpu_count(),
            "ipp": ipp_probe.check()}


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


def offline_rooflines(prof, steps, frames, nfft, hop, n_out1, copy_bw, freq_mask, soft=False):
    """Per-kernel HBM rooflines of the two offline passes from the engines' HIP-event timings.

    Algorithmic bytes per frame (SURVEY 8(d) accounting: compulsory traffic of each
    kernel as a stand-alone stage, half spectrum nfft/2+1 bins where the data is Hermitian):
      stft        : 4*hop in + 8*(nfft/2+1) spectrum + 4*(nfft/2+1) magnitude out
      freq_filter : 8 B per element (4 read + 4 written) of the frames x nfft matrix; where the engine filters
                    half rows of the frames x (nfft/2 + 1 + mask/2) it needs
      time_filter : 8 B per element of frames x nfft, or of frames x (nfft/2 + 1) with half rows
      istft       : what the kernel moves since the masks travel as bits (round 3): the spectrum ONCE 8*(nfft/2+1), one
                    word of mask bits per 16 bins (nfft/4 bytes), and per output 4*nwin = 8*hop written.  Soft masks: pass 1
                    loads one mask value per bin and output (4*(nfft/2+1) each) instead of the bits, pass 2 loads H and P
                    (8*(nfft/2+1)).  (Rounds 1-3 priced it as S + H + P per OUTPUT, which the kernels no longer read.)
      finalize    : per output 12*hop (two half frames in, one hop out); with hard masks the passes of the offline batch have none
                    (round 4): the synthesis (istft_run_kernel / istft_run_wide_kernel) adds the halves in registers and writes
                    4*hop per frame and destination instead of an 8*hop Y row per output"""
    out = {}
    for ps in ("pass1", "pass2"):
        N, h, F = nfft[ps], hop[ps], frames[ps]
        nout = n_out1 if ps == "pass1" else 1
        mf = freq_mask[ps]
        half = mf <= 63 or mf in (65, 85, 93, 129, 171, 187, 255, 257)  # hpr.hip run_chunk: half rows on the median path
        per_frame = {"stft": 4 * h + 12 * (N // 2 + 1),
                     "freq_filter": 8 * (N // 2 + 1 + mf // 2) if half else 8 * N,
                     "time_filter": 8 * (N // 2 + 1) if half else 8 * N,
                     "istft": 8 * (N // 2 + 1) + nout * 8 * h
                     + ((nout * 4 * (N // 2 + 1) if ps == "pass1" else 8 * (N // 2 + 1)) if soft else N // 4),
                     "finalize": nout * 12 * h}
        if mf <= 63 and not prof[ps].get("time_filter", {}).get("launches") and prof[ps].get("freq_filter", {}).get("launches") \
                and ps == "pass2":
            # median_tf_herm_bits_kernel: both medians in one launch.  SURVEY 8(d): "fused H+P stage (one read of abs(S), two
            # writes): 12 B per element" -- the kernel writes two mask bits per bin instead of the two rows
            per_frame["freq_filter"] = 12 * (N // 2 + 1)
        if not soft and prof[ps].get("istft", {}).get("launches") and not prof[ps].get("finalize", {}).get("launches"):
            # istft_run_kernel / istft_run_wide_kernel: synthesis + overlap-add in one launch; all it writes is the finished hop of
            # every destination (pass 2: P; pass 1 with three outputs: H, and P + R summed) -- no Y rows
            per_frame["istft"] = 8 * (N // 2 + 1) + N // 4 + (1 if nout == 1 else nout - 1) * 4 * h
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
            "device_copy_GBps": copy_bw, "frac_of_device_copy": d["achieved"] / copy_bw if copy_bw else None,
            "share_of_kernel_time": d["ms_per_step"] / sum(v["ms_per_step"] for v in out.values()),
            "note": "dominant kernel of the step by HIP-event time; every kernel's line is in `kernels`"}
    return roof, out


OFFLINE_PMC_FILE = "r06_offline_batch_pmc.json"   # tools/pmc_cmd.sh _kernel python3 bench.py --workload offline_batch ... (collect_profiles.sh)
OFFLINE_PMC_KERNELS = {"pass1.stft": "stft_real_kernel<14>", "pass1.freq_filter": "median_big_kernel<187", "pass1.istft": "istft_run_wide_kernel<14, 2>",
                       "pass2.stft": "stft_real_kernel<10>", "pass2.freq_filter": "median_tf_herm_bits_kernel<11, 13>",
                       "pass2.istft": "istft_run_kernel<10>"}


def offline_valu_issue(kern, clips, clip_seconds):
    """valu_issue_frac of the offline batch's VALU-bound kernels (SQ_INSTS_VALU per launch from the committed PMC pass of the
    SAME workload x 2 cycles over 1024 SIMDs x 2.4 GHz x the launch time measured here); an HBM fraction is the wrong ruler
    for the long-mask median (196 instructions per output).  Silent when the record is missing or for another batch."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", OFFLINE_PMC_FILE)))
    except (OSError, ValueError):
        return
    if rec.get("clips") != clips or rec.get("clip_seconds") != clip_seconds:
        return
    for line, prefix in OFFLINE_PMC_KERNELS.items():
        hit = [v for k, v in rec.get("kernels", {}).items() if k.startswith(prefix)]
        if line in kern and hit and hit[0].get("SQ_INSTS_VALU"):
            n = sum(h["SQ_INSTS_VALU"] for h in hit)          # per step: every matching kernel runs once per step
            kern[line]["valu_issue_frac"] = 2.0 * n / (1024 * 2.4e9 * 1e-3 * kern[line]["ms_per_step"])
            kern[line]["valu_instructions_per_step"] = n


def whole_step_roofline(frames, nfft, hop, n_out, ms_per_step):
    """The whole offline step on SURVEY 8(d)'s per-frame minimum (fused_bytes_per_hop: 24*(nfft/2+1) + 8*hop for one
    output, every further output reads the spectrum again and writes its hop), both passes, over the step's wall time."""
    tot = sum(frames[ps] * fused_bytes_per_hop(nfft[ps], hop[ps], n_out[ps]) for ps in frames)
    ach = tot / (ms_per_step * 1e-3) / 1e9
    return {"bound": "hbm", "algorithmic_bytes_per_step": tot, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
            "per_frame_bytes": {ps: fused_bytes_per_hop(nfft[ps], hop[ps], n_out[ps]) for ps in frames},
            "note": "SURVEY 8(d) per-frame minimum of the batched pipeline x frames of both passes / wall time of the step"}


COPY_DETAIL = {}        # what device_copy_bandwidth measured, for the detail record


def tuned_copy_record():
    """tools/ubench_copy --json: the best streaming copy KERNEL of this box (4 x 16 B per thread, nontemporal loads and stores)
    over the median kernel's working set and over 1 GiB, next to hipMemcpy device-to-device.  The binary is built by
    __graft_entry__.build() (tools/bin/, travels with the tree); compiled here with hipcc if it is missing."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "tools", "bin", "ubench_copy")
    tmp = None
    try:
        if not os.path.exists(exe):
            tmp = exe = os.path.join(tempfile.gettempdir(), "zen_ubench_copy_%d" % os.getpid())
            subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3",
                                   os.path.join(ROOT, "tools", "ubench_copy.hip"), "-o", exe], stderr=subprocess.DEVNULL, timeout=300)
        txt = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=120).stdout
        return next(json.loads(ln) for ln in txt.splitlines() if ln.startswith("{"))
    except Exception as exc:
        return {"error": str(exc)[:200]}
    finally:
        if tmp and os.path.exists(tmp):
            os.remove(tmp)


def device_copy_bandwidth(zen_amd, n_floats=1 << 28, iters=10):
    """The practical HBM roof of this box (read + write bytes per second), quoted next to the nominal 8 TB/s (BASELINE.md,
    roofline denominators): the TUNED copy kernel over the median kernel's working set -- a kernel that streams at
    0.72 of 8 TB/s is measured against the best a copy kernel does here, not against hipMemcpy, which it beats.  hipMemcpy
    device-to-device of 1 GiB is kept beside it (`hipMemcpy_d2d_GBps`); it is the fallback if the tool cannot run."""
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
    memcpy_bw = 8.0 * n_floats / dt / 1e9
    rec = tuned_copy_record()
    COPY_DETAIL.clear()
    COPY_DETAIL.update(rec)
    COPY_DETAIL["hipMemcpy_d2d_GBps"] = memcpy_bw
    tuned = max(rec.get("tuned_copy_median_shape_GBps", 0.0), rec.get("tuned_copy_1GiB_GBps", 0.0))
    COPY_DETAIL["device_copy_GBps_is"] = "tuned copy kernel" if tuned > 0 else "hipMemcpy device-to-device (the tuned copy tool did not run)"
    return tuned if tuned > 0 else memcpy_bw


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
        # the same loop with the opt-in resident kernel (zen_hip_hpr_set_resident): no launch per hop
        res_lines = subprocess.run([exe, "2000"], capture_output=True, text=True, timeout=120,
                                   env=dict(os.environ, ZEN_RT_RESIDENT="100")).stdout.splitlines()
        # both again with the light publication (write-through sample stores + relaxed flag: the default of rounds 4-5, opt-in
        # since round 6 -- ZEN_HIP_PUBLISH_LIGHT=1; the default is the system-scope release form)
        light_lines = subprocess.run([exe, "2000"], capture_output=True, text=True, timeout=120,
                                     env=dict(os.environ, ZEN_HIP_PUBLISH_LIGHT="1")).stdout.splitlines()
        light_res_lines = subprocess.run([exe, "2000"], capture_output=True, text=True, timeout=120,
                                         env=dict(os.environ, ZEN_HIP_PUBLISH_LIGHT="1", ZEN_RT_RESIDENT="100")).stdout.splitlines()
        os.remove(exe)
        sweep = [json.loads(ln) for ln in lines if ln.startswith("{")]
        rsweep = [json.loads(ln) for ln in res_lines if ln.startswith("{")]
        lsweep = [json.loads(ln) for ln in light_lines if ln.startswith("{")]
        lrsweep = [json.loads(ln) for ln in light_res_lines if ln.startswith("{")]
        res["publication"] = ("default: system-scope release fence + release store; *_light_*: ZEN_HIP_PUBLISH_LIGHT=1 (write-through "
                              "sample stores + relaxed flag)")
        res["light_us_by_hop"] = {("sse_" if r["sse"] else "") + str(r["hop"]): r["us_per_hop"] for r in lsweep}
        res["light_resident_us_by_hop"] = {("sse_" if r["sse"] else "") + str(r["hop"]): r["us_per_hop"] for r in lrsweep if r.get("resident")}
        lat = [r for r in lsweep if r["hop"] == HOP and not r["sse"]]
        lrat = [r for r in lrsweep if r.get("resident") and r["hop"] == HOP and not r["sse"]]
        if lat:
            res["light_us_per_hop"] = lat[0]["us_per_hop"]
        if lrat:
            res["light_resident_us_per_hop"] = lrat[0]["us_per_hop"]
        res["resident_us_by_hop"] = {("sse_" if r["sse"] else "") + str(r["hop"]): r["us_per_hop"] for r in rsweep if r.get("resident")}
        rat = [r for r in rsweep if r.get("resident") and r["hop"] == HOP and not r["sse"]]
        if rat:
            res["resident_us_per_hop"] = rat[0]["us_per_hop"]
            res["resident_hops_per_s"] = 1e6 / rat[0]["us_per_hop"]
            res["resident_note"] = ("opt-in zen_hip_hpr_set_resident(idle 100 ms): one workgroup stays on the device and takes each "
                                    "hop from a mailbox; same samples (tests/test_gpu_round4.py)")
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


class _Solo:
    """A world of one: what the legs that only rank 0 runs hand to the helpers in place of the process group."""

    def max(self, v):
        return float(v)

    def sum(self, vs):
        return [float(v) for v in vs]

    def barrier(self):
        pass


def traffic_record(kernel, elems):
    """HBM bytes per launch of `kernel` from the committed PMC passes (rocprofv3 cannot run inside this process).
    The record is keyed by the demangled kernel name the counters were collected under and carries the elements
    that launch processed: a record for another kernel or another shape is REFUSED (null + the reason), never quoted."""
    path = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    try:
        tj = json.load(open(path))
    except (OSError, ValueError):
        return None, "no profiles/%s" % TRAFFIC_FILE
    rec = tj.get("kernels", {}).get(kernel)
    if rec is None:
        return None, "profiles/%s holds no record for %s" % (TRAFFIC_FILE, kernel)
    if int(rec.get("elements", -1)) != int(elems):
        return None, "profiles/%s: the record of %s is for %s elements per launch, this launch has %d" % (
            TRAFFIC_FILE, kernel, rec.get("elements"), elems)
    return rec["hbm_bytes_per_launch"], ("committed PMC record profiles/%s, build %s -- replayed, NOT measured in this run (rocprofv3 --pmc "
                                         "cannot run inside this process; FETCH_SIZE and WRITE_SIZE in passes of their own, FETCH doubled per "
                                         "the gfx950 correction; keyed by exact kernel name and elements per launch)" % (TRAFFIC_FILE, tj.get("build", "?")))


def valu_issue_frac(kernel, elems, launch_s):
    """Share of the chip's VALU issue slots a launch uses: SQ_INSTS_VALU (wave instructions per launch, from the committed
    PMC pass of this exact kernel and shape) x 2 cycles per wave64 instruction (MI355X_MICROARCH.md: 32 lanes per cycle)
    over 1024 SIMDs x 2.4 GHz x the launch duration measured in THIS run.  Half-rate instructions (v_min / v_max / v_med3,
    f64) take two such slots each, so a kernel made of them saturates well below 1.0; null without a record."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE))).get("kernels", {}).get(kernel)
    except (OSError, ValueError):
        return None
    if not rec or int(rec.get("elements", -1)) != int(elems) or launch_s <= 0:
        return None
    n = rec.get("sq_counters_per_launch", {}).get("SQ_INSTS_VALU")
    return None if not n else 2.0 * n / (1024 * 2.4e9 * launch_s)


def fused_bytes_per_hop(nfft, hop, n_out):
    """SURVEY 8(d) per-frame minimum of the batched pipeline: 4*hop in, spectrum written + read 2*8*(nfft/2+1),
    magnitude written + read 2*4*(nfft/2+1), 4*hop out -- 24*(nfft/2+1) + 8*hop for one output; every further
    output reads the spectrum again and writes its hop."""
    return 24 * (nfft // 2 + 1) + 4 * hop + n_out * 4 * hop + (n_out - 1) * 8 * (nfft // 2 + 1)


def block_run(zen_amd, grp, x2d, flags, M, steps, warmup, settle_ms, barrier, hop=HOP, sse=False, copy_bord=True):
    """`steps` timed zen_hip_hpr_process calls of M hops per stream between two barriers; returns the engine too."""
    S, n = x2d.shape
    d_in = zen_amd.DeviceBuffer.from_host(x2d)
    want = {"P": bool(flags & zen_amd.OUTPUT_PERCUSSIVE), "H": bool(flags & zen_amd.OUTPUT_HARMONIC),
            "R": bool(flags & zen_amd.OUTPUT_RESIDUAL) and not sse}
    bufs = {k: (zen_amd.DeviceBuffer(S * n) if w else None) for k, w in want.items()}
    eng = zen_amd.HPR(FS, hop, BETA, flags, zen_amd.TIME_CAUSAL, copy_bord, S, M)
    if sse:
        eng.use_sse_filter()
    ptr = {k: (b.ptr if b else None) for k, b in bufs.items()}

    def step():
        eng.process(d_in.ptr, M, n, ptr["H"], ptr["P"], ptr["R"], n)

    t_end = time.perf_counter() + settle_ms / 1e3
    while time.perf_counter() < t_end:
        for _ in range(5):
            step()
        zen_amd.synchronize()
    for _ in range(warmup):
        step()
    barrier()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = grp.max(time.perf_counter() - t0)
    med = eng.profile_get()
    breakdown = eng.profile_get_all()
    eng.profile(False)
    first = next(b for b in (bufs["P"], bufs["H"], bufs["R"]) if b is not None)
    chk = float(np.abs(first.download(4096)).sum())            # liveness only; bytes, not data path
    return {"dt": dt, "breakdown": breakdown, "median": med, "checksum": chk, "eng": eng, "step": step,
            "bufs": [d_in] + [b for b in bufs.values() if b is not None]}


def free_run(run):
    run["eng"] = None
    run["step"] = None
    for b in run["bufs"]:
        b.free()
    run["bufs"] = []


def fused_roofline(run, S, M, steps, n_out, kernel, copy_bw):
    nfft = 4 * HOP
    fl = run["breakdown"]["rt_fused"]
    t_f = 1e-3 * fl["ms"] / fl["launches"]
    bph = fused_bytes_per_hop(nfft, HOP, n_out)
    ach = bph * S * M / t_f / 1e9
    direct = run["breakdown"]["finalize"]["launches"] == fl["launches"] and \
        run["breakdown"]["finalize"]["ms"] < 0.5 * 0.05 * fl["launches"]   # the kernel wrote the finished hops itself (one fix-up launch only)
    # HBM-side bytes the launch asks for per hop: the hop's input (4*hop; the previous hop it also reads is the neighbouring
    # workgroup's and comes from L2), per output the Y row of the frame (8*hop written) and -- where the kernel finishes the
    # hops itself -- the two half rows of the hop 128 items back read again (8*hop) and the finished hop written (4*hop):
    # 4*hop + n_out*20*hop (24 576 B for one output; DESIGN.md section 5 uses the same figure)
    moved = 4 * HOP + n_out * 8 * HOP + (n_out * 12 * HOP if direct else 0)
    tr, src = traffic_record(kernel, S * M * nfft)
    return {
        "bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
        "limiter": "valu-issue", "valu_issue_frac": valu_issue_frac(kernel, S * M * nfft, t_f),
        "device_copy_GBps": copy_bw, "frac_of_device_copy": ach / copy_bw if copy_bw else None,
        "traffic": tr, "traffic_source": src,
        "kernel": kernel + " (one workgroup per hop: STFT, |S|, 47-tap median, hard mask%s, iSTFT%s)"
                  % ("s" if n_out > 1 else "", " x%d" % n_out if n_out > 1 else ""),
        "hops_per_launch": S * M, "algorithmic_bytes_per_hop": bph,
        "algorithmic_bytes_formula": "24*(nfft/2+1) + 4*hop + n_out*4*hop + (n_out-1)*8*(nfft/2+1) (SURVEY 8(d): per-frame "
                                     "minimum of the batched pipeline; 24*(nfft/2+1) + 8*hop for one output)",
        "hbm_bytes_moved_per_hop_by_design": moved, "moved_GBps": moved * S * M / t_f / 1e9,
        "finishes_hops_itself": direct,
        "avg_launch_ms": 1e3 * t_f, "launches": fl["launches"],
        "share_of_step": (fl["ms"] / 1e3) / run["dt"] if run["dt"] > 0 else None,
        "note": "`achieved`/`frac` price the launch with SURVEY 8(d)'s ALGORITHMIC bytes as the bench contract asks; the "
                "kernel does not move them: spectrum, |S| and P stay in registers / LDS, its own HBM-side traffic is 4*hop read "
                "+ 8*hop written per hop and output, plus 8*hop read back and 4*hop written per output where it finishes the "
                "hops itself (`hbm_bytes_moved_per_hop_by_design`, `moved_GBps`, `traffic`).  Its limiter is VALU issue "
                "(`valu_issue_frac`: wave instructions x 2 cycles over the SIMD-cycles of the launch; half-rate min/max/med3 "
                "and f64 instructions take two slots): DESIGN.md section 5"}


def median_rooflines(zen_amd, run, S, M, copy_bw):
    """BASELINE's second metric.  burst: the kernel inside the three-kernel path of the engine (whole rows), 10 launches;
    sustained: >= 1 s of back-to-back launches through the plain drop-in wrapper (zen_hip_mfilt_run, no option, no
    promise: the kernel checks the sign bits of every row it stages) on a magnitude matrix; cold: a 512 MiB write
    between launches, each launch timed on its own."""
    eng, step = run["eng"], run["step"]
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

    dt3, (h_ms, h_launches, h_elems), kern3 = leg()      # as the engine runs it: half rows
    three = {"ms_per_step": 1e3 * dt3, "hops_per_s": S * M / dt3, "kernel_ms_per_step": kern3,
             "rows": "half (bins 0..2048 and 4073..4095 filtered)", "median_kernel": K_MEDIAN_HALF,
             "median_elements_per_launch": h_elems // max(h_launches, 1)}
    tr_h, src_h = traffic_record(K_MEDIAN_HALF, h_elems // max(h_launches, 1))
    three["median_traffic"], three["median_traffic_source"] = tr_h, src_h
    zen_amd.set_option("no_half_rows", 1)                # BASELINE's metric: the kernel over whole rows
    dt3f, (med_ms, med_launches, med_elems), kern3f = leg()
    three["whole_rows"] = {"ms_per_step": 1e3 * dt3f, "hops_per_s": S * M / dt3f, "kernel_ms_per_step": kern3f}
    zen_amd.set_option("no_half_rows", 0)
    zen_amd.set_option("no_block_fused", 0)
    t_b = 1e-3 * med_ms / max(med_launches, 1)
    el = med_elems // max(med_launches, 1)
    burst = 8.0 * el / t_b / 1e9 if t_b > 0 else 0.0

    # ---- sustained and cold, through the drop-in wrapper on the same shape
    rows, cols = S * M, 4 * HOP
    rng = np.random.default_rng(3)
    mat = rng.random((rows, cols), dtype=np.float32)       # magnitudes: >= +0
    src, dst = zen_amd.DeviceBuffer.from_host(mat), zen_amd.DeviceBuffer(rows * cols)
    del mat
    f = zen_amd.MedianFilterGPU(rows, cols, 47, zen_amd.FREQUENCY)
    for _ in range(5):
        f.filter(src, dst)
    zen_amd.synchronize()
    e0, e1 = zen_amd.Event(), zen_amd.Event()
    n_l, t_host = 0, time.perf_counter()
    e0.record()
    while time.perf_counter() - t_host < 1.2:
        for _ in range(200):
            f.filter(src, dst)
        n_l += 200
        zen_amd.synchronize()                             # (bounds the queue; 200 launches = ~30 ms of device work)
    e1.record()
    ms_sus = e0.elapsed_ms(e1) / n_l
    evs = [(zen_amd.Event(), zen_amd.Event()) for _ in range(60)]   # 60 consecutive launches, each timed on its own
    for a, b in evs:
        a.record()
        f.filter(src, dst)
        b.record()
    each = sorted(a.elapsed_ms(b) for a, b in evs)
    flush = zen_amd.DeviceBuffer(128 << 20)               # 512 MiB
    cold = []
    for _ in range(12):
        flush.zero()
        a, b = zen_amd.Event(), zen_amd.Event()
        a.record()
        f.filter(src, dst)
        b.record()
        cold.append(a.elapsed_ms(b))
    for bfr in (src, dst, flush):
        bfr.free()
    cold.sort()
    sus = 8.0 * rows * cols / (1e-3 * ms_sus) / 1e9
    cold_med = cold[len(cold) // 2]
    tr, tsrc = traffic_record(K_MEDIAN_WHOLE, el)
    roof = {
        "bound": "hbm", "achieved": sus, "peak": 8000.0, "unit": "GB/s", "frac": sus / 8000.0,
        "frac_is": "sustained (the conservative figure); burst and cold-cache beside it",
        "sustained": {"avg_launch_ms": ms_sus, "launches": n_l, "seconds": 1e-3 * ms_sus * n_l, "GBps": sus, "frac": sus / 8000.0,
                      "each_of_60_ms": {"min": each[0], "median": each[30], "max": each[-1]}},
        "burst": {"avg_launch_ms": 1e3 * t_b, "launches": med_launches, "GBps": burst, "frac": burst / 8000.0,
                  "note": "inside the engine's three-kernel path: the STFT kernel has just written the matrix"},
        "cold": {"median_launch_ms": cold_med, "min": cold[0], "max": cold[-1], "launches": len(cold),
                 "GBps": 8.0 * rows * cols / (1e-3 * cold_med) / 1e9, "frac": 8.0 * rows * cols / (1e-3 * cold_med) / 1e9 / 8000.0,
                 "note": "512 MiB written elsewhere before every launch: nothing of the input is left in the Infinity Cache"},
        "device_copy_GBps": copy_bw, "frac_of_device_copy": sus / copy_bw if copy_bw else None,
        "traffic": tr, "traffic_source": tsrc,
        "kernel": K_MEDIAN_WHOLE + " (frequency direction, 47 taps, whole 4096-bin rows; rows without a set sign bit keep raw-bit keys)",
        "elements_per_launch": el, "rows": rows, "cols": cols, "algorithmic_bytes_per_element": 8}
    roof["long_masks"] = long_mask_shapes(zen_amd)
    roof["device_copy"] = dict(COPY_DETAIL)
    return roof, three


# BASELINE's second metric on every shape BASELINE.md / SURVEY 8(d) list: (rows, cols, taps, direction).  Path shapes: 10 minutes of
# 44.1 kHz audio at hop 1024 / 256 / 512 / 4096 / 2048 (time and frequency masks of each); the reference's own bench:
# dim x dim, 11 taps, both directions (libzen/mfilt.bench.cu:222-262), at its two largest sizes.
MEDIAN_SHAPES = ((25840, 4096, 3, "t"), (25840, 4096, 47, "f"), (103360, 1024, 11, "t"), (103360, 1024, 13, "f"),
                 (51680, 2048, 7, "t"), (51680, 2048, 23, "f"), (12920, 8192, 93, "f"), (6460, 16384, 187, "f"),
                 (8192, 8192, 11, "t"), (8192, 8192, 11, "f"), (16384, 16384, 11, "t"), (16384, 16384, 11, "f"))


def median_shapes(zen_amd, seconds=1.0, shapes=MEDIAN_SHAPES, nonneg=False):
    """ONE protocol for all of them, the one `roofline.median47` is quoted on: >= `seconds` of back-to-back launches of the plain
    drop-in wrapper (zen_hip_mfilt_run; no option, no promise unless `nonneg`) on a matrix of magnitudes, timed with HIP events
    around the whole run on the stream the launches go to; 8 B per element."""
    out = []
    for rows, cols, flen, d in shapes:
        rng = np.random.default_rng(flen)
        src = zen_amd.DeviceBuffer.from_host(rng.random((rows, cols), dtype=np.float32))
        dst = zen_amd.DeviceBuffer(rows * cols)
        f = zen_amd.MedianFilterGPU(rows, cols, flen, zen_amd.FREQUENCY if d == "f" else zen_amd.TIME_ANTICAUSAL)
        if nonneg:
            f.assume_nonneg()
        for _ in range(3):
            f.filter(src, dst)
        zen_amd.synchronize()
        e0, e1 = zen_amd.Event(), zen_amd.Event()
        t_probe = time.perf_counter()
        f.filter(src, dst)
        zen_amd.synchronize()
        per = max(time.perf_counter() - t_probe, 1e-5)
        batch = int(min(200, max(2, 0.03 / per)))          # ~30 ms of device work between the host's looks at the clock
        n_l, t_host = 0, time.perf_counter()
        e0.record()
        while time.perf_counter() - t_host < seconds:
            for _ in range(batch):
                f.filter(src, dst)
            n_l += batch
            zen_amd.synchronize()
        e1.record()
        ms = e0.elapsed_ms(e1) / n_l
        gb = 8.0 * rows * cols / (1e-3 * ms) / 1e9
        out.append({"rows": rows, "cols": cols, "taps": flen, "direction": "frequency" if d == "f" else "time", "avg_launch_ms": ms,
                    "launches": n_l, "seconds": 1e-3 * ms * n_l, "GBps": gb, "frac": gb / 8000.0, "assume_nonneg": bool(nonneg)})
        del f
        src.free()
        dst.free()
    return out


def long_mask_shapes(zen_amd, iters=30):
    """The long frequency masks through the plain wrapper, so that every instantiation of median_big_kernel a user can reach has
    a number in the line: 93 taps on 8192-bin rows (hop 2048 at 44.1 kHz), 187 taps on 16384-bin rows (hop 4096: pass 1 of the
    offline path) and 255 taps -- the longest mask the API accepts, 1.1 KB of scratch per lane (kernel_resources.json)."""
    out = []
    for rows, cols, flen in ((12920, 8192, 93), (6460, 16384, 187), (6460, 16384, 255)):
        rng = np.random.default_rng(flen)
        src = zen_amd.DeviceBuffer.from_host(rng.random((rows, cols), dtype=np.float32))
        dst = zen_amd.DeviceBuffer(rows * cols)
        f = zen_amd.MedianFilterGPU(rows, cols, flen, zen_amd.FREQUENCY)
        for _ in range(3):
            f.filter(src, dst)
        zen_amd.synchronize()
        e0, e1 = zen_amd.Event(), zen_amd.Event()
        e0.record()
        for _ in range(iters):
            f.filter(src, dst)
        e1.record()
        ms = e0.elapsed_ms(e1) / iters
        gb = 8.0 * rows * cols / (1e-3 * ms) / 1e9
        out.append({"rows": rows, "cols": cols, "taps": flen, "ms": ms, "GBps": gb, "frac": gb / 8000.0, "launches": iters})
        src.free()
        dst.free()
    return out


def sse_rooflines(prof, steps, frames, nfft, hop):
    """Per-kernel lines of the SSE block path (BASELINE configs[4]); whole rows: the box mean is not mirror symmetric."""
    fused = not prof.get("time_filter", {}).get("launches")     # sse_block.hip: both boxes + masks + synthesis in one launch
    per_frame = {"stft": 4 * hop + 8 * (nfft // 2 + 1) + 4 * nfft, "freq_filter": 8 * nfft, "time_filter": 8 * nfft,
                 "istft": 8 * (nfft // 2 + 1) + 8 * nfft + 8 * hop, "finalize": 12 * hop}
    if fused:    # the spectrum row and the magnitude row once each (the neighbours' rows of the time box are re-reads), one Y row
        per_frame.update({"stft": 4 * hop + 12 * (nfft // 2 + 1), "istft": 12 * (nfft // 2 + 1) + 8 * hop})
    out = {}
    for k, v in prof.items():
        if v["launches"] and k in per_frame:
            ms = v["ms"] / steps
            ach = per_frame[k] * frames / (ms * 1e-3) / 1e9
            out[k] = {"ms_per_step": ms, "algorithmic_bytes_per_frame": per_frame[k], "achieved": ach, "frac": ach / 8000.0}
    dom = max(out, key=lambda k: out[k]["ms_per_step"])
    return {"bound": "hbm", "achieved": out[dom]["achieved"], "peak": 8000.0, "unit": "GB/s", "frac": out[dom]["frac"],
            "traffic": None, "kernel": dom + (" (sse_synth_kernel: time box + frequency box + Wiener mask + inverse transform)" if fused and dom == "istft" else ""),
            "launches_per_step": sum(1 for v in prof.values() if v["launches"]),
            "note": "dominant kernel of the step by HIP-event time"}, out


def offline_batch_run(zen_amd, zdist, grp, rank, world, C, clip_seconds, steps, warmup, settle_ms, barrier, rooflines=True):
    """BASELINE configs[3]: C clips per GPU, clips sharded over the ranks, both passes resident in HBM."""
    n = int(clip_seconds * FS)
    hop_h, hop_p = 4096, 256
    ids = zdist.shard_units(C * world, world, rank)          # clip ids of this rank (C each)
    x = np.stack([s_music(n, seed=7000 + i) for i in ids])
    d_in = zen_amd.DeviceBuffer.from_host(x)
    d_h, d_p = zen_amd.DeviceBuffer(C * n), zen_amd.DeviceBuffer(C * n)
    eng = zen_amd.HPRIOffline(FS, hop_h, hop_p, BETA, BETA, False, C)
    n1, n2 = eng.hop_counts(n)

    def step():
        eng.process_device(d_in.ptr, n, n, d_h.ptr, d_p.ptr, None, n)

    t_end = time.perf_counter() + settle_ms / 1e3
    step()
    zen_amd.synchronize()
    while time.perf_counter() < t_end:
        step()
        zen_amd.synchronize()
    for _ in range(warmup):
        step()
    barrier()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = grp.max(time.perf_counter() - t0)
    prof = eng.profile_get_all()
    eng.profile(False)
    chk, ranks = grp.sum([float(np.abs(d_p.download(4096)).sum()), 1.0])
    res = {"metric": "hops/sec (HPR-I offline, hops of both passes)", "unit": "hops/s",
           "value": world * C * (n1 + n2) * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
           "x_realtime": world * C * clip_seconds * steps / dt, "ranks_reported": int(round(ranks)),
           "config": {"workload": "HPRIOffline<GPU> 4096/256 beta 2.0 hard mask, %d x %.0f s mono S-music clips per GPU "
                                  "resident in HBM, both passes + harmonic/percussive outputs" % (C, clip_seconds),
                      "clips_per_gpu": C, "clips_total": C * world, "clip_samples": n, "hops_pass1": n1, "hops_pass2": n2,
                      "parallelism": "clips sharded x%d, no data-path collective" % world},
           "checksum": chk}
    if rooflines and rank == 0:
        fr, nf, hp = {"pass1": C * n1, "pass2": C * n2}, {"pass1": 4 * hop_h, "pass2": 4 * hop_p}, {"pass1": hop_h, "pass2": hop_p}
        roof, kern = offline_rooflines(prof, steps, fr, nf, hp, 3, None, {"pass1": 187, "pass2": 13})
        offline_valu_issue(kern, C, clip_seconds)
        if "valu_issue_frac" in kern.get(roof["kernel"], {}):
            roof["valu_issue_frac"] = kern[roof["kernel"]]["valu_issue_frac"]
        res.update({"roofline": roof, "kernels": kern,
                    "whole_step": whole_step_roofline(fr, nf, hp, {"pass1": 3, "pass2": 1}, res["ms_per_step"])})
    first_clip = x[0].copy()
    eng = None
    for b in (d_in, d_h, d_p):
        b.free()
    return res, first_clip, (n1, n2)


def offline_long_run(zen_amd, zdist, grp, rank, world, steps, warmup, settle_ms, barrier):
    """BASELINE configs[2]: one 10-minute stereo clip = 2 mono channels, HPR-I 4096/256, soft mask p = 2.  N > 1: every
    channel is cut into N time ranges, rank r computes range r of both channels from its own halo of
    input; strong scaling, no exchange."""
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

    step()
    zen_amd.synchronize()
    t_end = time.perf_counter() + settle_ms / 1e3
    while time.perf_counter() < t_end:
        step()
        zen_amd.synchronize()
    for _ in range(warmup):
        step()
    barrier()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = grp.max(time.perf_counter() - t0)
    prof = eng.profile_get_all()
    eng.profile(False)
    chk, ranks = grp.sum([float(np.abs(d_p.download(min(4096, max(e - b, 1)))).sum()), 1.0])
    res = {"metric": "hops/sec (HPR-I offline, hops of both passes)", "unit": "hops/s", "scaling": "strong",
           "value": 2 * (n1 + n2) * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
           "x_realtime": 600.0 * steps / dt, "ranks_reported": int(round(ranks)),
           "config": {"workload": "HPRIOffline<GPU> 4096/256 beta 2.5 soft mask (p = 2), one 10-minute stereo clip (2 mono "
                                  "channels of 26 460 000 samples) resident in HBM, zen_hip_hpri_process_range",
                      "hops_pass1": n1, "hops_pass2": n2,
                      "parallelism": "each channel time-sharded x%d with warm-up halos, no exchange" % world},
           "checksum": chk}
    if rank == 0 and world == 1:
        fr, nf, hp = {"pass1": 2 * n1, "pass2": 2 * n2}, {"pass1": 4 * hop_h, "pass2": 4 * hop_p}, {"pass1": hop_h, "pass2": hop_p}
        roof, kern = offline_rooflines(prof, steps, fr, nf, hp, 2, None, {"pass1": 187, "pass2": 13}, soft=True)
        res.update({"roofline": roof, "kernels": kern,
                    "whole_step": whole_step_roofline(fr, nf, hp, {"pass1": 2, "pass2": 1}, res["ms_per_step"])})
    ch0 = chans[0]
    eng = None
    for bfr in d_in + [d_h, d_p]:
        bfr.free()
    return res, ch0, (n1, n2)


def block_host_run(zen_amd, x, M, steps=8, warmup=3):
    """The headline block with HOST buffers on both sides -- the reference's timed region (zen/fakert.h:221-247: host hop in,
    process_next_hop, copy_percussive, host hop out) for the whole 25 840-hop block: zen_hip_hpr_process_host, pieces of the block
    going up / through the fused kernel / back down on three streams.  Pinned buffers; 4 bytes per sample each way, so the
    roof is the slower direction of the host link, measured here with the same buffers (one copy up, one down, at once)."""
    import ctypes as C
    n = M * HOP
    pin_in, pin_out = zen_amd.PinnedHost(n), zen_amd.PinnedHost(n)
    pin_in.array[:] = x[:n]
    eng = zen_amd.HPR(FS, HOP, BETA, zen_amd.OUTPUT_PERCUSSIVE, zen_amd.TIME_CAUSAL, True, 1, 0)
    for _ in range(warmup):
        eng.process_host(pin_in.array, perc=pin_out.array)
    walls = []
    for _ in range(steps):
        t0 = time.perf_counter()
        eng.process_host(pin_in.array, perc=pin_out.array)
        walls.append(1e3 * (time.perf_counter() - t0))
    chk = float(np.abs(pin_out.array[:4096]).sum())
    # the link with the same buffers: n floats up and n floats down at once on two streams
    lib = zen_amd.load()
    d_a, d_b = zen_amd.DeviceBuffer(n), zen_amd.DeviceBuffer(n)
    d_b.zero()
    s1, s2 = C.c_void_p(), C.c_void_p()
    lib.zen_hip_stream_create(C.byref(s1))
    lib.zen_hip_stream_create(C.byref(s2))
    best = {"up": 1e30, "down": 1e30, "both": 1e30}
    for what in ("up", "down", "both"):
        for _ in range(4):
            zen_amd.synchronize()
            t0 = time.perf_counter()
            if what != "down":
                lib.zen_hip_memcpy_h2d_async(d_a.ptr, pin_in.array.ctypes.data, 4 * n, s1)
            if what != "up":
                lib.zen_hip_memcpy_d2h_async(pin_out.array.ctypes.data, d_b.ptr, 4 * n, s2)
            zen_amd.synchronize(s1)
            zen_amd.synchronize(s2)
            best[what] = min(best[what], 1e3 * (time.perf_counter() - t0))
    lib.zen_hip_stream_destroy(s1)
    lib.zen_hip_stream_destroy(s2)
    eng = None
    d_a.free()
    d_b.free()
    pin_in.free()
    pin_out.free()
    wall = float(np.median(walls))
    return {"value": M / (1e-3 * wall), "unit": "hops/s", "wall_ms": wall, "wall_ms_min": min(walls), "steps": steps,
            "x_realtime": M / (1e-3 * wall) * HOP / FS, "hops_per_step": M, "checksum": chk,
            "link": {"h2d_ms": best["up"], "d2h_ms": best["down"], "both_ms": best["both"], "h2d_GBps": 4e-6 * n / best["up"],
                     "d2h_GBps": 4e-6 * n / best["down"]},
            # the roof: the slower direction alone (a full-duplex link moves the other one at the same time); `both_ms` is what
            # two whole-block copies issued at once on two streams take on this box (1.2-2x the roof: they share the link's
            # engines unevenly), the practical floor of ANY schedule of these bytes
            "roofline": {"bound": "host link (full duplex: 4 B per sample up, 4 B down)", "roof_ms": max(best["up"], best["down"]),
                         "achieved_ms": wall, "frac": max(best["up"], best["down"]) / wall, "both_directions_at_once_ms": best["both"]},
            "config": {"workload": "the headline block (hop 1024, P only, hard mask, causal; %d hops) from pinned HOST memory back to "
                                   "pinned host memory: zen_hip_hpr_process_host, wall clock of the call" % M,
                       "timed_region": "zen/fakert.h:221-247 for the whole block: host in -> process -> host out"}}


def link_roof(zen_amd, n, reps=5):
    """The host link's share of HPRIOffline::process on an n-sample clip, measured on this box with pinned buffers
    through the library's own copies: 4n bytes up, 8n bytes down -- each alone, and both at once on two streams (the
    roof of `offline_host`)."""
    import ctypes as C
    lib = zen_amd.load()
    up, dn = zen_amd.IOGPU(n), [zen_amd.IOGPU(n), zen_amd.IOGPU(n)]     # host_out: plain pinned host memory
    up.host_out[:] = 0.25
    for d in dn:
        d.host_out[:] = 0
    d_up, d_dn = zen_amd.DeviceBuffer(n), zen_amd.DeviceBuffer(2 * n)
    d_dn.zero()
    s1, s2 = C.c_void_p(), C.c_void_p()
    lib.zen_hip_stream_create(C.byref(s1))
    lib.zen_hip_stream_create(C.byref(s2))

    def run(do_up, do_down):
        best = 1e30
        for _ in range(reps):
            zen_amd.synchronize()
            t0 = time.perf_counter()
            if do_up:
                lib.zen_hip_memcpy_h2d_async(d_up.ptr, up.host_out.ctypes.data, 4 * n, s1)
            if do_down:
                for i, d in enumerate(dn):
                    lib.zen_hip_memcpy_d2h_async(d.host_out.ctypes.data, d_dn.offset(i * n), 4 * n, s2)
            zen_amd.synchronize(s1)
            zen_amd.synchronize(s2)
            best = min(best, time.perf_counter() - t0)
        return 1e3 * best

    res = {"h2d_ms": run(True, False), "d2h_ms": run(False, True), "both_ms": run(True, True)}
    res["h2d_GBps"] = 4e-6 * n / res["h2d_ms"]
    res["d2h_GBps"] = 8e-6 * n / res["d2h_ms"]
    res["both_GBps"] = 12e-6 * n / res["both_ms"]
    lib.zen_hip_stream_destroy(s1)
    lib.zen_hip_stream_destroy(s2)
    d_up.free()
    d_dn.free()
    del up, dn
    return res


def offline_host_sharded_run(zen_amd, grp, rank, world, seconds, barrier, numa_cpus, reps=3):
    """N ranks, each with its own pageable host clip of `seconds` (mono S-music, another seed per rank), all calling
    HPRIOffline<GPU>::process (zen_hip_hpri_process: upload, both passes, two downloads, pipelined) at the same time: what
    `zen batch --gpus N` does to the host.  Per rank the wall time of its own call; for the job the audio of all ranks over the
    slowest rank's time, and the bytes that crossed the host links in that time (12 per sample and rank) -- on one node the
    ranks share the host's DRAM bandwidth, so this curve, not the HBM-resident one, is what a user's batch scales like."""
    n = int(seconds * FS)
    base = s_music(int(30 * FS), seed=7000 + rank)
    x = np.tile(base, -(-n // base.size))[:n].copy()
    outs = [np.zeros(n, np.float32) for _ in range(3)]
    eng = zen_amd.HPRIOffline(FS, 4096, 256, BETA, BETA)
    eng.process(x, out=tuple(outs))                                    # warm-up: staging buffers, engine growth, registration
    walls = []
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        eng.process(x, out=tuple(outs))
        walls.append(time.perf_counter() - t0)
    mine = min(walls)
    slowest = grp.max(mine)
    one_hot = [0.0] * world
    one_hot[rank] = 1e3 * mine
    per_rank = grp.sum(one_hot)
    chk, ranks = grp.sum([float(np.abs(outs[1][:4096]).sum()), 1.0])
    return {"metric": "x real time, N ranks x HPRIOffline<GPU>::process on host vectors at the same time", "unit": "x_realtime",
            "value": world * seconds / slowest, "per_rank_wall_ms": per_rank, "slowest_rank_wall_ms": 1e3 * slowest,
            "per_rank_x_realtime_min": seconds / slowest, "clip_seconds_per_rank": seconds, "ranks_reported": int(round(ranks)),
            "host_link_GBps_aggregate": 12e-9 * n * world / slowest, "bytes_per_sample": 12, "checksum": chk,
            "numa_cpus_rank0": (numa_cpus[:4] + ["..."] + numa_cpus[-1:]) if numa_cpus and len(numa_cpus) > 6 else numa_cpus,
            "host_stats_rank0": eng.host_stats(),
            "config": {"workload": "%d ranks, each: HPRIOffline<GPU>(44100, 4096, 256, 2.0, 2.0).process on its own %.0f s mono "
                                   "clip in pageable host memory, concurrently; no exchange between ranks" % (world, seconds)}}


def cpp_process_ms(seconds, reps=3):
    """HPRIOffline<GPU>::process(std::vector<float>) through the C++ host mirror (zen_amd/libzen), timed like
    zen/offline.h:141-147 by tools/offline_host.cpp: the by-value copy of the clip and the three result vectors included."""
    import subprocess
    import tempfile
    exe = os.path.join(tempfile.gettempdir(), "zen_offline_host_%d" % os.getpid())
    zdir = os.path.join(ROOT, "zen_amd")
    try:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(zdir, "libzen"),
                               os.path.join(ROOT, "tools", "offline_host.cpp"), "-o", exe, "-L", zdir, "-lzen", "-lzen_hip",
                               "-Wl,-rpath," + zdir], stderr=subprocess.DEVNULL)
        txt = subprocess.run([exe, str(seconds), str(reps)], capture_output=True, text=True, timeout=300).stdout
        os.remove(exe)
        return next(json.loads(ln) for ln in txt.splitlines() if ln.startswith("{"))
    except Exception as exc:
        return {"error": str(exc)[:200]}


def offline_host_run(zen_amd, seconds=3600.0, reps=4, cpu_baseline=True, variants=False):
    """What the reference's `zen offline` times (zen/offline.h:141-147): HPRIOffline<GPU>::process on HOST vectors,
    wall clock, copies included -- zen_hip_hpri_process on a `seconds`-long mono S-music clip in plain (pageable) numpy
    arrays, 4096/256, beta 2, hard masks.  Beside it: the link's roof measured on this box (4 B up + 8 B down per sample,
    pinned, both directions at once), the kernels alone on the same time ranges (resident), the C++ mirror's
    process(std::vector<float>) with its allocations, and the oracle timed the same way on a bounded sample."""
    n = int(seconds * FS)
    base = s_music(int(60 * FS), seed=4242)
    x = np.tile(base, -(-n // base.size))[:n].copy()
    outs = [np.zeros(n, np.float32) for _ in range(3)]                 # allocated and touched, like the caller's vectors
    eng = zen_amd.HPRIOffline(FS, 4096, 256, BETA, BETA)
    n1, n2 = eng.hop_counts(n)

    def timed(opt=None, bufs=None):
        for k, v in (opt or {}).items():
            zen_amd.set_option(k, v)
        try:
            src, dst = (x, outs) if bufs is None else bufs
            eng.process(src, out=tuple(dst))                           # warm-up: staging buffers, engine growth
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                eng.process(src, out=tuple(dst))
                ts.append(1e3 * (time.perf_counter() - t0))
            return {"wall_ms_min": min(ts), "wall_ms_median": sorted(ts)[len(ts) // 2], "x_realtime": seconds / (1e-3 * min(ts)),
                    "stats": eng.host_stats()}
        finally:
            for k in (opt or {}):
                zen_amd.set_option(k, 0)

    main = timed()
    chk = float(np.abs(outs[1][:4096]).sum())
    res = {"metric": "x real time, HPRIOffline<GPU>::process on host vectors (zen/offline.h:141-147)", "unit": "x_realtime",
           "clip_seconds": seconds, "value": main["x_realtime"], "wall_ms": main["wall_ms_min"], "wall_ms_median": main["wall_ms_median"], "reps": reps,
           "hops_per_s": (n1 + n2) / (1e-3 * main["wall_ms_min"]), "host_stats": main["stats"], "checksum": chk,
           "config": {"workload": "HPRIOffline<GPU>(44100, 4096, 256, 2.0, 2.0).process on a %.0f s mono S-music clip in pageable host "
                                  "memory (numpy), three host outputs; zen_hip_hpri_process" % seconds,
                      "samples": n, "hops_pass1": n1, "hops_pass2": n2, "bytes_up": 4 * n, "bytes_down": 8 * n}}
    # ---- the serial path of rounds 1-3 (one range: up, both passes, down), same build
    res["serial_one_range"] = timed({"offline_range": 1 << 30})
    res["speedup_over_serial"] = res["serial_one_range"]["wall_ms_min"] / main["wall_ms_min"]
    if variants:
        res["no_register"] = timed({"offline_no_register": 1})
        for r in (1 << 20, 1 << 21, 1 << 22, 1 << 24):
            res["range_%d" % r] = timed({"offline_range": r})
        try:
            io = [zen_amd.IOGPU(n) for _ in range(3)]
            io[0].host_out[:] = x
            res["pinned_caller_buffers"] = timed(bufs=(io[0].host_out, [io[1].host_out, io[2].host_out, None]))
            del io
        except Exception as exc:
            res["pinned_caller_buffers"] = {"error": str(exc)[:200]}
    # ---- the kernels alone, same ranges, everything resident
    st = main["stats"]
    d_in, d_h, d_p = zen_amd.DeviceBuffer.from_host(x), zen_amd.DeviceBuffer(n), zen_amd.DeviceBuffer(n)
    rng = int(st["range_samples"])

    def compute():
        for b in range(0, n, rng):
            e = min(b + rng, n)
            eng.process_range(d_in.ptr, n, b, e, d_h.offset(b), d_p.offset(b))
        zen_amd.synchronize()

    compute()
    tc = []
    for _ in range(reps):
        t0 = time.perf_counter()
        compute()
        tc.append(1e3 * (time.perf_counter() - t0))
    for b in (d_in, d_h, d_p):
        b.free()
    eng = None
    res["compute_ms"] = min(tc)
    link = link_roof(zen_amd, n)
    res["link"] = link
    # The link is full duplex: the roof is the slower direction at the rate a pinned copy of that size gets on this box
    # (8 B down per sample).  Two whole-clip copies issued at once on two streams do NOT overlap on this runtime (both_ms =
    # the sum); the pipeline's 32 / 64 MB pieces do, which is why the call beats that figure.
    roof_ms = max(link["h2d_ms"], link["d2h_ms"])
    res["roofline"] = {"bound": "host link (PCIe, full duplex), pinned copies on this box", "roof_ms": roof_ms,
                       "achieved_ms": main["wall_ms_min"], "frac": roof_ms / main["wall_ms_min"],
                       "bytes_per_sample": {"up": 4, "down": 8}, "down_GBps_at_roof": link["d2h_GBps"],
                       "achieved_down_GBps": 8e-6 * n / main["wall_ms_min"],
                       "split_ms": {"h2d_alone": link["h2d_ms"], "d2h_alone": link["d2h_ms"], "both_issued_at_once": link["both_ms"],
                                    "kernels_alone": res["compute_ms"]},
                       "note": "roof = max(4 B up, 8 B down per sample) through pinned buffers at the rates measured in this run; "
                               "the call itself works on the caller's pageable vectors (registered for the call)"}
    res["cpp_process"] = cpp_process_ms(seconds)
    if cpu_baseline:
        from oracle import oracle as o
        m = int(20 * FS)
        ro = o.HPRIOffline(FS, 4096, 256, BETA, BETA)
        t0 = time.perf_counter()
        ro.process(x[:m])
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": (m / FS) / dt, "unit": "x_realtime", "cores": 1, "kind": "port",
                               "sample": "first 20 s of the same clip, oracle HPRIOffline 4096/256 hard mask process() on host "
                                         "vectors, 1 thread", "wall_s": dt, "host_cpu": host_cpu_name()}
        res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
    return res


def short_source(src):
    """The provenance of a replayed counter record in under 160 characters (what the driver's record keeps of a string)."""
    if not src:
        return src
    mm = re.search(r"profiles/(\S+?), build (\S+)", src)
    if mm and src.startswith("committed PMC record"):
        return "REPLAYED from committed PMC record profiles/%s (build %s), not measured in this run" % (mm.group(1), mm.group(2))
    return src[:150]


def compact_line(full):
    """The one line the driver keeps (it stores the contract keys, `config`, `roofline`, `cpu_baseline` and only the NAMES of
    anything else, and cuts the tail of long lines): everything both BASELINE metrics and the north-star targets need sits
    inside those four; the per-kernel tables of every leg go to the detail file (`--detail`: print them instead)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "x_realtime", "ranks_reported", "checksum")
    line = {k: full[k] for k in keep if k in full}
    cfg = dict(full.get("config", {}))
    legs = {}
    for name in ("offline_batch", "offline_long", "offline_batch_sharded"):
        leg = full.get(name)
        if leg:
            cfg[name + "_x_realtime"] = leg.get("x_realtime")
            legs[name] = {"x_realtime": leg.get("x_realtime"), "ms_per_step": leg.get("ms_per_step"), "hops_per_s": leg.get("value"),
                          "whole_step_frac_of_hbm_roof": leg.get("whole_step", {}).get("frac")}
            if name == "offline_batch_sharded" and full.get("offline_host_sharded"):
                hs = full["offline_host_sharded"]
                cfg["offline_host_sharded_x_realtime"] = hs.get("value")
                cfg["offline_host_sharded_link_GBps"] = hs.get("host_link_GBps_aggregate")
                legs["offline_host_sharded"] = {k: hs.get(k) for k in ("value", "per_rank_wall_ms", "slowest_rank_wall_ms", "ranks_reported",
                                                                       "host_link_GBps_aggregate", "clip_seconds_per_rank", "numa_cpus_rank0")}
            if name == "offline_batch_sharded":
                cfg["offline_batch_sharded_ranks_reported"] = leg.get("ranks_reported")
                cfg["offline_batch_sharded_clips_total"] = leg.get("config", {}).get("clips_total")
    oh = full.get("offline_host")
    if oh:
        cfg["offline_host_x_realtime"] = oh.get("value")
        # what a `zen offline` user sees: the reference's literal C++ signature process(std::vector<float>) (by-value clip, three
        # new result vectors), by copy and from a caller that moves its clip in
        cfg["offline_host_cpp_x_realtime"] = oh.get("cpp_process", {}).get("x_realtime")
        cfg["offline_host_cpp_moved_x_realtime"] = oh.get("cpp_process", {}).get("moved_x_realtime")
        legs["offline_host"] = {"x_realtime": oh.get("value"), "wall_ms": oh.get("wall_ms"), "clip_seconds": oh.get("clip_seconds"),
                                "frac_of_link_roof": oh.get("roofline", {}).get("frac"),
                                "link_roof_ms": oh.get("roofline", {}).get("roof_ms"), "kernels_alone_ms": oh.get("compute_ms"),
                                "serial_one_range_ms": oh.get("serial_one_range", {}).get("wall_ms_min"),
                                "cpp_process_ms": oh.get("cpp_process", {}).get("ms_min"),
                                "cpp_process_moved_ms": oh.get("cpp_process", {}).get("moved_ms_min"),
                                "cpu_oracle_x_realtime": oh.get("cpu_baseline", {}).get("value")}
    for name in ("all_outputs", "s_noise", "sse_block"):
        if name in full:
            cfg[name + "_hops_per_s"] = full[name].get("value")
    bh = full.get("block_host")
    if bh:     # `value` with the host copies inside the timed region (zen/fakert.h:221-247), PCIe-bound
        cfg["block_host_hops_per_s"] = bh.get("value")
        cfg["block_host_frac_of_link_roof"] = bh.get("roofline", {}).get("frac")
        legs["block_host"] = {"hops_per_s": bh.get("value"), "wall_ms": bh.get("wall_ms"), "x_realtime": bh.get("x_realtime"),
                              "frac_of_link_roof": bh.get("roofline", {}).get("frac"), "link_roof_ms": bh.get("roofline", {}).get("roof_ms"),
                              "h2d_GBps": bh.get("link", {}).get("h2d_GBps"), "d2h_GBps": bh.get("link", {}).get("d2h_GBps"),
                              "api": "zen_hip_hpr_process_host (pinned host in -> pinned host out)"}
    if "realtime" in full:
        cfg["per_hop_api_us"] = full["realtime"].get("us_per_hop")
        cfg["per_hop_api_hops_per_s"] = full["realtime"].get("hops_per_s")
        cfg["per_hop_api_resident_us"] = full["realtime"].get("resident_us_per_hop")
        cfg["per_hop_api_light_us"] = full["realtime"].get("light_us_per_hop")            # (opt-in publication, ZEN_HIP_PUBLISH_LIGHT=1)
        cfg["per_hop_api_light_resident_us"] = full["realtime"].get("light_resident_us_per_hop")
        cfg["per_hop_publication"] = "release (default); *_light_*: ZEN_HIP_PUBLISH_LIGHT=1"
        # the same call at the other hops of the reference's sweep and on the SSE path (BASELINE configs[4]: hop 512), C++ loop
        by, rby = full["realtime"].get("per_hop_us_by_hop", {}), full["realtime"].get("resident_us_by_hop", {})
        cfg["per_hop_api_us_by_hop"] = {k: by[k] for k in ("256", "512", "2048", "4096", "sse_512") if k in by}
        cfg["per_hop_api_resident_us_by_hop"] = {k: rby[k] for k in ("256", "512", "2048", "4096", "sse_512") if k in rby}
    line["config"] = cfg
    if legs:
        line["legs"] = legs
    if "roofline" in full:
        r = full["roofline"]
        roof = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "limiter", "valu_issue_frac",
                                      "avg_launch_ms", "launches", "hops_per_launch", "algorithmic_bytes_per_hop",
                                      "hbm_bytes_moved_per_hop_by_design", "device_copy_GBps", "algorithmic_bytes_per_frame",
                                      "frames_per_step", "roof_ms", "achieved_ms", "split_ms", "elements_per_launch") if k in r}
        # The driver's record keeps the SCALARS of `roofline` and `config` (strings cut at ~160 characters, nested objects
        # dropped): everything a reader of that record needs is a flat scalar here, the nested objects repeat it with detail.
        if r.get("traffic_source"):
            roof["traffic_source"] = short_source(r["traffic_source"])
        m = full.get("roofline_median")
        if m:      # BASELINE's second metric
            roof["median47_frac"] = m["frac"]
            roof["median47_avg_launch_us"] = 1e3 * m["sustained"]["avg_launch_ms"]
            roof["median47_launches"] = m["sustained"]["launches"]
            roof["median47_traffic"] = m.get("traffic")
            roof["median47_is"] = ("%s, %d x %d, 47 taps, plain zen_hip_mfilt_run (no promise), >= 1 s back to back, 8 B/element"
                                   % (K_MEDIAN_WHOLE, m["rows"], m["cols"]))
            for x in m.get("shapes", []):      # every shape BASELINE.md lists: >= 1 s of back-to-back launches, 8 B per element
                roof["median_frac_%dx%d_%s%d" % (x["rows"], x["cols"], x["direction"][0], x["taps"])] = round(x["frac"], 4)
            if m.get("shapes"):
                roof["median_frac_is"] = ("rows x cols _ t|f taps: >= 1 s of back-to-back zen_hip_mfilt_run per shape on a handle with "
                                          "zen_hip_mfilt_assume_nonneg (the engine's launches); plain wrapper: median_shapes_plain")
            roof["median47"] = {"sustained_seconds": m["sustained"]["seconds"], "burst_frac": m["burst"]["frac"], "cold_frac": m["cold"]["frac"],
                                "traffic_source": short_source(m.get("traffic_source")), "frac_of_device_copy": m.get("frac_of_device_copy"),
                                "long_masks_frac": {"%d taps / %d bins" % (x["taps"], x["cols"]): round(x["frac"], 4)
                                                    for x in m.get("long_masks", [])}}
            if m.get("shapes_plain"):
                roof["median_shapes_plain"] = {"%dx%d/%s%d" % (x["rows"], x["cols"], x["direction"][0], x["taps"]): round(x["frac"], 3)
                                               for x in m["shapes_plain"]}
            roof["device_copy_is"] = m.get("device_copy", {}).get("device_copy_GBps_is")
            roof["hipMemcpy_d2d_GBps"] = m.get("device_copy", {}).get("hipMemcpy_d2d_GBps")
        line["roofline"] = roof
    if "cpu_baseline" in full:
        c = full["cpu_baseline"]
        line["cpu_baseline"] = {k: c.get(k) for k in ("value", "unit", "cores", "kind", "sample", "host_cpu") if k in c}
        if "gpu_over_cpu" in full:
            line["gpu_over_cpu"] = full["gpu_over_cpu"]
    return line


def emit(full, detail):
    """Rank 0: the compact line on stdout; the full record to gpurun_out/bench_detail.json when that directory can be
    written (it is what profiles/rNN_*_bench_default.json is copied from), or instead of the compact line with --detail."""
    if detail:
        print(json.dumps(full))
        return
    line = compact_line(full)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_detail.json"), "w") as f:
            json.dump(full, f)
        line["detail"] = "gpurun_out/bench_detail.json (per-kernel tables of every leg; `--detail` prints it)"
    except OSError:
        pass
    print(json.dumps(line))


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
    line = {"metric": "dry run (launch / sharding plumbing only)", "dry": True, "value": None,
            "unit": "hops/s", "n_gpus": world, "ranks_reported": int(ranks), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1),
            "higher_is_better": True, "scaling": "strong" if args.workload == "offline_long" else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "none",
            "config": {"workload": args.workload, "units_all_ranks": int(tot_units),
                       "units_rank0": units, "parallelism": par, "backend": "gloo"}}
    if args.workload == "realtime_block" and world > 1 and not args.no_legs:
        # the sharded offline-batch leg of the N > 1 line: every rank owns --leg-clips clips of the global batch
        mine = zdist.shard_units(args.leg_clips * world, world, rank)
        grp.barrier()
        tot_clips, ranks2 = grp.sum([len(mine), 1])
        line["offline_batch_sharded"] = {"dry": True, "clips_total": int(tot_clips), "clips_rank0": len(mine),
                                         "ranks_reported": int(ranks2)}
        # the host-clip leg's aggregation: per-rank wall times gathered as a one-hot sum, the slowest rank, the rank count
        mine_ms = 1.0 + rank
        one_hot = [0.0] * world
        one_hot[rank] = mine_ms
        per_rank = grp.sum(one_hot)
        line["offline_host_sharded"] = {"dry": True, "per_rank_wall_ms": per_rank, "slowest_rank_wall_ms": grp.max(mine_ms),
                                        "ranks_reported": int(grp.sum([1])[0]),
                                        "numa_bound": zdist.gpu_numa_cpus(rank) is not None}
    if rank == 0:
        print(json.dumps(line))
    grp.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="realtime_block", choices=["realtime_block", "offline_batch", "offline_long", "offline_host"])
    ap.add_argument("--host-seconds", type=float, default=3600.0, help="offline_host: length of the clip")
    ap.add_argument("--host-variants", action="store_true", help="offline_host: also time range lengths / unregistered / pinned buffers")
    ap.add_argument("--host-shard-seconds", type=float, default=600.0, help="N > 1: length of every rank's host clip in the offline_host_sharded leg")
    ap.add_argument("--hops", type=int, default=25840, help="hops per step per stream (25840 = 10 min)")
    ap.add_argument("--streams", type=int, default=1, help="independent streams per GPU")
    ap.add_argument("--clips", type=int, default=64, help="offline_batch: clips per GPU")
    ap.add_argument("--clip-seconds", type=float, default=30.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-realtime", action="store_true")
    ap.add_argument("--no-legs", action="store_true",
                    help="realtime_block: only `value` and its roofline -- none of the legs outside the timed region "
                         "(roofline_median, all_outputs, s_noise, sse_block, offline_batch, offline_long)")
    ap.add_argument("--leg-steps", type=int, default=6, help="timed steps of each leg (2 warm-up steps before them)")
    ap.add_argument("--leg-clips", type=int, default=64, help="clips per GPU of the offline_batch leg")
    ap.add_argument("--outputs", default="P", choices=["P", "HPR"],
                    help="realtime_block: percussive only (the headline config) or all three outputs")
    ap.add_argument("--seed-kind", default="music", choices=["music", "noise"], help="realtime_block input: S-music or S-noise")
    ap.add_argument("--no-block-fused", action="store_true",
                    help="realtime_block: STFT / median / iSTFT kernels instead of the fused per-hop kernel")
    ap.add_argument("--fused-minb", type=int, default=0, help="tuning: occupancy the fused kernel is built for")
    ap.add_argument("--settle-ms", type=float, default=1000.0,
                    help="untimed steps run for this long before the W warm-up steps, so that the clocks have "
                         "settled when the timed region starts (the timed region is still exactly K steps)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL over xGMI; gloo for the CPU plumbing test)")
    ap.add_argument("--detail", action="store_true", help="print the full record (every leg's per-kernel tables) instead of the compact line")
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
    numa_cpus = zdist.bind_to_gpu_numa(local_rank) if world > 1 else None   # before torch / HIP: the ranks' host buffers stay local
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d from the launcher; using WORLD_SIZE" % (args.gpus, world),
              file=sys.stderr)
    if args.dry:
        return dry_main(args, zdist)

    legs = args.workload == "realtime_block" and not args.no_legs and not args.no_block_fused and args.outputs == "P" \
        and args.seed_kind == "music" and args.streams == 1
    cpu_all = None
    if world == 1 and not args.no_cpu_baseline and (args.workload == "offline_batch" or legs):
        cpu_all = cpu_baseline_offline_all_cores(4096, 256)      # forks: before anything loads or touches the GPU
    import torch
    if os.environ.get("ZEN_ALLOW_GPU_SHARING") and torch.cuda.device_count() > 0:
        # testing the N-rank path on a box with fewer GPUs than ranks (use --backend gloo: RCCL refuses two ranks
        # on one device); never a measurement configuration
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    grp = zdist.Group(args.backend, torch.device("cuda", local_rank) if args.backend == "nccl" else None,
                      force=bool(os.environ.get("ZEN_FORCE_PROCESS_GROUP")))

    import zen_amd
    zen_amd.init(local_rank)
    if args.no_block_fused:
        zen_amd.set_option("no_block_fused", 1)
    if args.fused_minb:
        zen_amd.set_option("block_fused_minb", args.fused_minb)

    def barrier():
        torch.cuda.synchronize()
        zen_amd.synchronize()
        grp.barrier()
        torch.cuda.synchronize()

    out = {"metric": "hops/sec (1024-hop HPR, 44.1 kHz mono)", "unit": "hops/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic"}

    if args.workload == "realtime_block":
        M, S = args.hops, args.streams
        n = M * HOP
        gen = s_music if args.seed_kind == "music" else s_noise
        x = np.stack([gen(n, seed=1000 * rank + s) for s in range(S)])
        all_out = args.outputs == "HPR"
        flags = (zen_amd.OUTPUT_PERCUSSIVE | zen_amd.OUTPUT_HARMONIC | zen_amd.OUTPUT_RESIDUAL) if all_out \
            else zen_amd.OUTPUT_PERCUSSIVE
        run = block_run(zen_amd, grp, x, flags, M, args.steps, args.warmup, args.settle_ms, barrier)
        dt, breakdown = run["dt"], run["breakdown"]
        chk, ranks = grp.sum([run["checksum"], 1.0])
        fused = breakdown["rt_fused"]["launches"] > 0
        if rank == 0:
            copy_bw = device_copy_bandwidth(zen_amd)
            value = world * S * M * args.steps / dt
            if fused:
                roof = fused_roofline(run, S, M, args.steps, 3 if all_out else 1, K_FUSED_HPR if all_out else K_FUSED_P, copy_bw)
            else:
                med_ms, med_launches, med_elems = run["median"]
                t_med = 1e-3 * med_ms / max(med_launches, 1)
                el_med = med_elems // max(med_launches, 1)
                ach_med = 8.0 * el_med / t_med / 1e9 if t_med > 0 else 0.0
                tr, src = traffic_record(K_MEDIAN_HALF, el_med)
                roof = {"bound": "hbm", "achieved": ach_med, "peak": 8000.0, "unit": "GB/s", "frac": ach_med / 8000.0,
                        "traffic": tr, "traffic_source": src, "kernel": K_MEDIAN_HALF, "elements_per_launch": el_med,
                        "avg_launch_ms": 1e3 * t_med, "share_of_step": (med_ms / 1e3) / dt if dt > 0 else None}
            out.update({
                "value": value, "ms_per_step": 1e3 * dt / args.steps, "ranks_reported": int(round(ranks)),
                "config": {
                    "workload": "HPRRealtime<GPU> semantics: hop 1024, nwin 2048, transform 4096, beta 2.0, "
                                "%s, hard mask, causal; S-%s 44.1 kHz mono stream resident "
                                "in HBM; block mode (zen_hip_hpr_process), %d hops/step/stream"
                                % ("OUTPUT_HARMONIC|PERCUSSIVE|RESIDUAL" if all_out else "OUTPUT_PERCUSSIVE", args.seed_kind, M),
                    "hops_per_step": M, "streams_per_gpu": S, "fs": FS, "hop": HOP,
                    "time_mask": 3, "freq_mask": 47, "parallelism": "replicas x%d" % world,
                    "api": "zen_hip_hpr_process: the block form of process_next_hop (an MI355X extension; the reference's "
                           "per-hop API is timed in `realtime`)",
                    "path": "fused per-hop kernel + overlap-add" if fused else "STFT / median / iSTFT kernels + overlap-add"},
                "x_realtime": value * HOP / FS,
                "checksum": chk,
                "kernel_ms_per_step": {k: v["ms"] / args.steps for k, v in breakdown.items() if v["launches"]},
                "roofline": roof})
            if legs and fused and world == 1:
                out["roofline_median"], out["three_kernel_path"] = median_rooflines(zen_amd, run, S, M, copy_bw)
                # BASELINE's second metric on every listed shape, one sustained protocol: the engine's launches (handles with the
                # non-negativity promise: |S| >= +0), and the plain wrapper beside it for a shorter run
                out["roofline_median"]["shapes"] = median_shapes(zen_amd, 1.0, nonneg=True)
                out["roofline_median"]["shapes_plain"] = median_shapes(zen_amd, 0.35, nonneg=False)
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_realtime(x[0])
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            if world == 1 and not args.no_realtime:
                out["realtime"] = realtime_leg(zen_amd, x[0])
        free_run(run)
        if legs and rank == 0 and world == 1 and fused:
            out["block_host"] = block_host_run(zen_amd, x[0], M)
        if legs:
            ls, lw = args.leg_steps, 2

            def sync_only():
                zen_amd.synchronize()

            solo = _Solo()
            if rank == 0 and world == 1:
                # -- H + P + R on the same stream (second case)
                flags3 = zen_amd.OUTPUT_PERCUSSIVE | zen_amd.OUTPUT_HARMONIC | zen_amd.OUTPUT_RESIDUAL
                r3 = block_run(zen_amd, solo, x, flags3, M, 20, 5, 100.0, sync_only)
                v3 = S * M * 20 / r3["dt"]
                out["all_outputs"] = {"value": v3, "unit": "hops/s", "ms_per_step": 1e3 * r3["dt"] / 20, "steps": 20,
                                      "x_realtime": v3 * HOP / FS, "outputs": "H+P+R", "checksum": r3["checksum"],
                                      "kernel_ms_per_step": {k: v["ms"] / 20 for k, v in r3["breakdown"].items() if v["launches"]},
                                      "roofline": fused_roofline(r3, S, M, 20, 3, K_FUSED_HPR, copy_bw)}
                free_run(r3)
                # -- the headline configuration on the S-noise seed
                xn = s_noise(n, seed=1)[None, :]
                rn = block_run(zen_amd, solo, xn, zen_amd.OUTPUT_PERCUSSIVE, M, 20, 5, 100.0, sync_only)
                vn = S * M * 20 / rn["dt"]
                out["s_noise"] = {"value": vn, "unit": "hops/s", "ms_per_step": 1e3 * rn["dt"] / 20, "steps": 20,
                                  "x_realtime": vn * HOP / FS, "checksum": rn["checksum"],
                                  "roofline": fused_roofline(rn, S, M, 20, 1, K_FUSED_P, copy_bw)}
                free_run(rn)
                del xn
                # -- BASELINE configs[4]: SSE path, hop 512 (nwin 1024, transform 2048, boxes 7 / 23), nocopybord
                hop5, M5 = 512, 2 * M
                x5 = x[:, :M5 * hop5]
                r5 = block_run(zen_amd, solo, x5, zen_amd.OUTPUT_PERCUSSIVE, M5, 10, 3, 100.0, sync_only, hop=hop5, sse=True,
                               copy_bord=False)
                v5 = M5 * 10 / r5["dt"]
                roof5, kern5 = sse_rooflines(r5["breakdown"], 10, M5, 4 * hop5, hop5)
                out["sse_block"] = {"value": v5, "unit": "hops/s", "ms_per_step": 1e3 * r5["dt"] / 10, "steps": 10,
                                    "x_realtime": v5 * hop5 / FS, "checksum": r5["checksum"], "roofline": roof5, "kernels": kern5,
                                    "config": {"workload": "HPRRealtime<GPU>(44100, 512, 2.0, P, nocopybord).use_sse_filter() "
                                                           "semantics, block mode, %d hops/step" % M5}}
                free_run(r5)
                # -- the offline workloads, short
                ob, clip0, (n1, n2) = offline_batch_run(zen_amd, zdist, solo, 0, 1, args.leg_clips, args.clip_seconds, ls, lw,
                                                       100.0, sync_only)
                if not args.no_cpu_baseline:
                    ob["cpu_baseline"] = cpu_baseline_offline(clip0, 4096, 256, n1 + n2)
                    ob["cpu_baseline_all_cores"] = cpu_all
                out["offline_batch"] = ob
                ol, ch0, (m1, m2) = offline_long_run(zen_amd, zdist, solo, 0, 1, ls, lw, 100.0, sync_only)
                if not args.no_cpu_baseline:
                    ol["cpu_baseline"] = cpu_baseline_offline(ch0, 4096, 256, m1 + m2, seconds=4.0, beta=2.5, soft=True)
                out["offline_long"] = ol
                # -- the reference's own offline entry point: process() on host vectors, copies included
                out["offline_host"] = offline_host_run(zen_amd, args.host_seconds, 4, not args.no_cpu_baseline)
            elif world > 1:
                # -- the path that shards: the offline batch, clips dealt to the ranks, no data-path collective
                ob, _, _ = offline_batch_run(zen_amd, zdist, grp, rank, world, args.leg_clips, args.clip_seconds, ls, lw, 100.0,
                                             barrier, rooflines=False)
                if rank == 0:
                    out["offline_batch_sharded"] = ob
                # -- and the path a user runs (`zen batch --gpus N`): every rank separates its own HOST clip through
                #    HPRIOffline::process at the same time -- N x 12 bytes per sample over the host links
                oh = offline_host_sharded_run(zen_amd, grp, rank, world, args.host_shard_seconds, barrier, numa_cpus)
                if rank == 0:
                    out["offline_host_sharded"] = oh
    elif args.workload == "offline_batch":
        res, clip0, (n1, n2) = offline_batch_run(zen_amd, zdist, grp, rank, world, args.clips, args.clip_seconds, args.steps,
                                                 args.warmup, args.settle_ms, barrier)
        if rank == 0:
            out.update(res)
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_offline(clip0, 4096, 256, n1 + n2)
                out["cpu_baseline_all_cores"] = cpu_all
    elif args.workload == "offline_long":
        res, ch0, (n1, n2) = offline_long_run(zen_amd, zdist, grp, rank, world, args.steps, args.warmup, args.settle_ms, barrier)
        if rank == 0:
            out.update(res)
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_offline(ch0, 4096, 256, n1 + n2, seconds=4.0, beta=2.5, soft=True)
    elif args.workload == "offline_host":
        if rank == 0:
            res = offline_host_run(zen_amd, args.host_seconds, max(args.leg_steps, 3), not args.no_cpu_baseline, args.host_variants)
            out.update(res)
            out["ms_per_step"] = res["wall_ms"]
            out["steps"], out["warmup"] = res["reps"], 1
    if rank == 0:
        emit(out, args.detail)
    grp.close()


if __name__ == "__main__":
    main()
