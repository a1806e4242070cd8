"""GPU parity tests added in round 4.  Every call goes through the C-ABI (libzen_hip.so via ctypes) and is compared
BIT-EXACTLY (tolerance 0) with the CPU oracle."""
import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu
ALL = o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE | o.OUTPUT_RESIDUAL
FS = 44100.0


@pytest.fixture(scope="module")
def z():
    import zen_amd
    zen_amd.init(0)
    return zen_amd


# ---------------------------------------------------------------------------- low sample rates: long frequency masks on short rows
@pytest.mark.parametrize("fs,hop", [(2000.0, 64), (2500.0, 64), (3000.0, 128), (2000.0, 128), (4000.0, 256)])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("flags", [ALL, o.OUTPUT_PERCUSSIVE])
def test_low_sample_rate_blocks_vs_oracle(z, fs, hop, causal, flags):
    """l_perc is 500 Hz of bins (hps.h:229): at fs 2000, hop 64 the frequency mask is 65 taps on 256-bin rows, and the
    row's last mf/2 = 32 bins (replicate border) no longer fit slot 15 of the transposed mask-bit layout
    (nfft/16 = 16 bins).  Blocks of >= 8 frames with hard masks must take the H / P rows path there
    (stft.h mask_bits_supported) and give the oracle's samples, like the per-hop path."""
    hh = o.HPR(fs, hop, 2.0, flags, o.TIME_CAUSAL if causal else o.TIME_ANTICAUSAL)
    n_hops = 2 * hh.stft_width + 40
    rng = np.random.default_rng(int(fs) + hop)
    t = np.arange(hop * n_hops) / fs
    x = (0.3 * np.sin(2 * np.pi * 0.11 * fs * t) + 0.3 * rng.uniform(-1, 1, t.size)).astype(np.float32)
    x[::hop * 5 + 3] += 0.9
    ref = hh.process_stream(x)
    keys = "PHR" if flags == ALL else "P"
    cz = z.TIME_CAUSAL if causal else z.TIME_ANTICAUSAL
    block = z.HPR(fs, hop, 2.0, flags, cz).process_stream_host(x)          # one block call: >= 8 frames
    g = z.HPR(fs, hop, 2.0, flags, cz)
    per_hop = [g.process_stream_host(x[i * hop:(i + 1) * hop]) for i in range(n_hops)]
    for k in keys:
        assert np.any(ref[k] != 0)
        assert np.array_equal(block[k], ref[k]), ("block", k)
        assert np.array_equal(np.concatenate([p[k] for p in per_hop]), ref[k]), ("per hop", k)


# ---------------------------------------------------------------------------- HPRIOffline::process on host vectors: the pipeline
def _clip(n, seed):
    from tests.test_gpu_parity import music
    return music(n, seed)


@pytest.mark.parametrize("hop_h,hop_p,n,rng_samples", [(4096, 256, 161571, 40960), (1024, 256, 70001, 8192),
                                                      (4096, 256, 4096 * 20 + 11, 4096), (512, 128, 30000, 30000 - 1)])
@pytest.mark.parametrize("mode", ["hard", "soft", "sse"])
@pytest.mark.parametrize("register", [True, False])
def test_host_process_pipeline_vs_oracle(z, hop_h, hop_p, n, rng_samples, mode, register):
    """zen_hip_hpri_process (HPRIOffline<GPU>::process, hps.cu:128-221) as a pipeline over time ranges -- forced onto
    short clips with "offline_range" so that the oracle can check every sample: registered (asynchronous copies) and
    unregistered (blocking copies, upload of range k+1 before the download of range k) caller buffers, hard / soft / SSE,
    a last range of a few samples, outputs left out."""
    x = _clip(n, 7 + n % 5)
    ro = o.HPRIOffline(FS, hop_h, hop_p, 2.0, 2.0)
    if mode == "soft":
        ro.use_soft_mask()
    if mode == "sse":
        ro.use_sse_filter()
    rh, rp, rr = ro.process(x)
    z.set_option("offline_range", rng_samples)
    z.set_option("offline_no_register", 0 if register else 1)
    try:
        g = z.HPRIOffline(FS, hop_h, hop_p, 2.0, 2.0)
        if mode == "soft":
            g.use_soft_mask()
        if mode == "sse":
            g.use_sse_filter()
        for rep in range(2):                      # the second call reuses the staging buffers, streams and events
            h, p, r = (np.full(n, np.nan, np.float32) for _ in range(3))
            g.process(x, out=(h, p, r))
            st = g.host_stats()
            assert st["n_ranges"] > 1 and st["range_samples"] % hop_h == 0
            assert bool(st["input_pinned"]) == register and bool(st["outputs_pinned"]) == register
            assert np.array_equal(h, rh) and np.array_equal(p, rp) and np.array_equal(r, rr)
        p2 = np.full(n, np.nan, np.float32)       # percussive alone (zen offline --only-percussive)
        g.process(x, out=(None, p2, None))
        assert np.array_equal(p2, rp)
    finally:
        z.set_option("offline_range", 0)
        z.set_option("offline_no_register", 0)
    assert np.all(rr == 0)


def test_host_process_pinned_caller_buffers_and_default_ranges(z):
    """The default range length (4 Mi samples for a clip of 9 Mi) with caller buffers that are already pinned
    (hipHostMalloc through zen_hip_host_alloc_mapped): no registration, asynchronous copies; equal to the whole clip in
    one range, and to plain numpy buffers registered for the call."""
    n = 9 * (1 << 20) + 123
    x = _clip(1 << 20, 3)
    x = np.tile(x, 10)[:n].copy()
    g = z.HPRIOffline(FS, 4096, 256, 2.0, 2.0)
    z.set_option("offline_range", 1 << 30)
    try:
        ref_h, ref_p, ref_r = g.process(x)                          # one range: up, both passes, down
        assert g.host_stats()["n_ranges"] == 1
    finally:
        z.set_option("offline_range", 0)
    bufs = [z.IOGPU(n) for _ in range(3)]                           # (host_out: plain pinned host memory)
    bufs[0].host_out[:] = x
    h, p = bufs[1].host_out, bufs[2].host_out
    h[:] = np.nan
    p[:] = np.nan
    g.process(bufs[0].host_out, out=(h, p, None))
    st = g.host_stats()
    assert st["n_ranges"] == 3 and st["input_pinned"] and st["outputs_pinned"]
    assert np.array_equal(h, ref_h) and np.array_equal(p, ref_p)
    hh, pp, rr = g.process(x)                                       # plain numpy arrays: registered for the call
    assert g.host_stats()["n_ranges"] == 3
    assert np.array_equal(hh, ref_h) and np.array_equal(pp, ref_p) and np.all(rr == 0) and np.all(ref_r == 0)


# ---------------------------------------------------------------------------- MedianFilterGPU: rows with and without negative samples
@pytest.mark.parametrize("variant", [0, 1])
def test_median47_rows_rekeyed_only_where_negative(z, variant):
    """median47_dpp_kernel (47 taps on 4096-bin rows, the shape of BASELINE's median metric) stages every row as raw bits
    and re-keys only rows in which some sample has its sign bit set: magnitude matrices run at the engine's speed through
    the plain zen_hip_mfilt_run, signed data stays exact.  Rows of every kind next to each other: all positive, one
    negative sample (in the middle, in the first and in the last column: the replicate border, mfilt.h:270-342), -0.0
    among positives, all negative, +/-inf; against the oracle (MedianFilterCPU semantics) bit for bit; and a handle with
    zen_hip_mfilt_assume_nonneg on the non-negative rows alone."""
    rows, cols = 64, 4096
    rng = np.random.default_rng(47)
    a = rng.random((rows, cols), dtype=np.float32)                      # >= +0
    a[1, 2000] = -0.5
    a[2, 0] = -1e-30
    a[3, cols - 1] = -3.0
    a[4, 17] = -0.0
    a[5] = -a[5]
    a[6] = rng.uniform(-1, 1, cols).astype(np.float32)
    a[7, 100:130] = np.inf
    a[8, 100:130] = -np.inf
    a[9, ::2] = 0.0
    a[10] = 0.0
    a[11] = -0.0
    for r in range(12, rows, 3):
        a[r] = rng.uniform(-1, 1, cols).astype(np.float32)
    z.set_option("median47_variant", variant)
    try:
        got = z.MedianFilterGPU(rows, cols, 47, z.FREQUENCY).filter_host(a)
        f = z.MedianFilterGPU(rows, cols, 47, z.FREQUENCY)
        f.assume_nonneg()
        pos = np.abs(a)                                                  # (abs clears the sign of -0.0 too)
        got_pos = f.filter_host(pos)
    finally:
        z.set_option("median47_variant", 0)
    ref = o.median_filter(a, 47, o.FREQUENCY)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(got_pos.view(np.uint32), o.median_filter(pos, 47, o.FREQUENCY).view(np.uint32))


def test_diagnostic_options_are_not_in_the_shipped_library(z):
    """Timing diagnostics whose outputs are not medians / not the reference's masks exist in -DZEN_HIP_DIAG builds only;
    the process-wide "mfilt_nonneg" is gone (per handle: zen_hip_mfilt_assume_nonneg)."""
    for name, val in (("median47_variant", 2), ("median47_variant", 3), ("rt_fused_diag", 1), ("mask_divide", 1),
                      ("mfilt_nonneg", 1)):
        with pytest.raises(z.ZenHipError):
            z.set_option(name, val)
    z.set_option("median47_variant", 1)
    z.set_option("median47_variant", 0)


# ---------------------------------------------------------------------------- the N-GPU plumbing on one GPU
def _run_py(code, env_extra=None, timeout=300):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    return subprocess.run([sys.executable, "-c", code], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout)


def test_rccl_process_group_world_of_one():
    """zen_amd.dist.Group with the `nccl` backend (= RCCL) at WORLD_SIZE 1: communicator bound to the device, all-reduce
    MAX and SUM of the scalars bench.py exchanges, barrier, destroy -- what every rank of the 8-GPU run does first and last
    (SURVEY 8(e)); a fresh interpreter, as the launcher starts them."""
    code = (
        "import torch\n"
        "from zen_amd import dist as zd\n"
        "torch.cuda.set_device(0)\n"
        "g = zd.Group('nccl', torch.device('cuda', 0), force=True)\n"
        "assert g.dist is not None and g.world == 1\n"
        "g.barrier()\n"
        "assert g.max(3.5) == 3.5\n"
        "assert g.sum([1.0, 2.5]) == [1.0, 2.5]\n"
        "import zen_amd\n"
        "zen_amd.init(0)\n"
        "g.barrier()\n"
        "g.close()\n"
        "print('RCCL_OK')\n")
    r = _run_py(code)
    assert r.returncode == 0 and b"RCCL_OK" in r.stdout, r.stderr.decode()[-2000:]


def test_bench_under_the_launcher_path_with_rccl():
    """`bench.py --gpus 1` started the way `--gpus N` starts its ranks (zen_amd.dist.spawn_ranks: fresh interpreter, RANK /
    WORLD_SIZE / MASTER_* in the environment) with the process group forced on: RCCL init, the barriers and all-reduces
    around the timed region, the sharded-units arithmetic and the one JSON line -- on the one GPU this box has."""
    import json
    code = (
        "import sys\n"
        "from zen_amd import dist as zd\n"
        "sys.exit(zd.spawn_ranks([sys.executable, 'bench.py', '--gpus', '1', '--steps', '2', '--warmup', '1', '--hops', '2048',\n"
        "                         '--no-legs', '--no-cpu-baseline', '--no-realtime', '--settle-ms', '0'], 1, timeout=240,\n"
        "                        env_extra={'ZEN_FORCE_PROCESS_GROUP': '1'}))\n")
    r = _run_py(code)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["ranks_reported"] == 1 and j["value"] > 0 and j["roofline"]["frac"] > 0


# ---------------------------------------------------------------------------- SSE path, blocks of frames: analysis + ONE fused launch
@pytest.mark.parametrize("hop", [128, 256, 512, 1024])
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("flags", [ALL, o.OUTPUT_PERCUSSIVE, o.OUTPUT_HARMONIC])
def test_sse_blocks_fused_synthesis_vs_oracle_and_four_launch_path(z, hop, causal, flags):
    """apply_sse_filter (hps.cu:582-652) for blocks of frames: sse_synth_kernel (time box, frequency box, Wiener masks
    and the inverse transforms of a frame in one workgroup; no H / P rows) against the oracle, against the four-launch
    path it replaces ("no_sse_block") and against per-hop calls; two streams, uneven blocks (history rows of the time
    box come from the previous call's ring rows), a switch to SSE in mid-stream."""
    cz, co = (z.TIME_CAUSAL, o.TIME_CAUSAL) if causal else (z.TIME_ANTICAUSAL, o.TIME_ANTICAUSAL)
    ho = o.HPR(FS, hop, 2.0, flags, co)
    ho.use_sse_filter()
    n_hops = 3 * ho.stft_width + 31
    x = np.stack([_clip(hop * n_hops, 5 + hop), _clip(hop * n_hops, 6 + hop)])
    refs = []
    for s in range(2):
        h = o.HPR(FS, hop, 2.0, flags, co)
        h.use_sse_filter()
        refs.append(h.process_stream(x[s]))
    keys = [k for k, f in (("P", o.OUTPUT_PERCUSSIVE), ("H", o.OUTPUT_HARMONIC)) if flags & f]    # no residual with SSE

    def run(opt, block):
        if opt:
            z.set_option(opt, 1)
        try:
            g = z.HPR(FS, hop, 2.0, flags, cz, False, 2)
            g.use_sse_filter()
            return g.process_stream_host(x, block=block)
        finally:
            if opt:
                z.set_option(opt, 0)

    for block in (None, 7, 2):
        got = run(None, block)
        old = run("no_sse_block", block)
        for k in keys:
            for s in range(2):
                assert np.array_equal(got[k][s], refs[s][k]), (block, k, s)
                assert np.any(refs[s][k] != 0)
            assert np.array_equal(got[k], old[k]), ("no_sse_block", block, k)
    # median path first, SSE from hop `first` on (hps.h:289: use_sse_filter may be called at any time)
    first = ho.stft_width + 9
    h = o.HPR(FS, hop, 2.0, flags, co)
    ra = h.process_stream(x[0][:first * hop])
    h.use_sse_filter()
    rb = h.process_stream(x[0][first * hop:])
    g = z.HPR(FS, hop, 2.0, flags, cz)
    ga = g.process_stream_host(x[0][:first * hop], block=first)
    g.use_sse_filter()
    gb = g.process_stream_host(x[0][first * hop:], block=11)
    for k in keys:
        assert np.array_equal(ga[k], ra[k]) and np.array_equal(gb[k], rb[k]), ("switch", k)


def test_sse_bench_block_windows_vs_oracle(z):
    """BASELINE configs[4] at the size bench.py times it: ONE zen_hip_hpr_process call of 51 680 hops (hop 512, SSE,
    nocopybord, percussive output) against the oracle on windows at its start, middle and end.  A causal SSE output hop
    depends on the stft_width - 1 frames before it (time box) and on the hop before it (overlap-add): an oracle started
    stft_width + 2 hops early from zero state is exact from the window's first hop on -- except that zero history rows
    are 1/0 = inf for the oracle's first frames, which is what a fresh stream sees too, so the windows away from the start
    compare only hops whose time box holds no pre-start row."""
    hop, M, K = 512, 51680, 20
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    x = b.s_music(M * hop, seed=0)
    g = z.HPR(FS, hop, 2.0, o.OUTPUT_PERCUSSIVE, z.TIME_CAUSAL, False, 1, M)
    g.use_sse_filter()
    got = g.process_stream_host(x)["P"]
    W = o.HPR(FS, hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL).stft_width
    for first in (0, M // 2, M - K):
        j = max(first - (W + 2), 0)
        ho = o.HPR(FS, hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL, False)
        ho.use_sse_filter()
        ref = ho.process_stream(x[j * hop:(first + K) * hop])["P"]
        skip = first - j
        assert np.array_equal(got[first * hop:(first + K) * hop], ref[skip * hop:], equal_nan=True), first
        assert np.any(np.nan_to_num(ref[skip * hop:]) != 0)


# ---------------------------------------------------------------------------- the per-hop API through a resident kernel
def _per_hop(z, rt, io, x, hop, n_hops, copy, pause_every=0, pause_s=0.0):
    import time
    out = np.zeros(hop * n_hops, np.float32)
    for i in range(n_hops):
        io.host_in[:] = x[i * hop:(i + 1) * hop]
        rt.process_next_hop(io.device_in)
        copy(io.device_out)
        out[i * hop:(i + 1) * hop] = io.host_out
        if pause_every and i % pause_every == pause_every - 1:
            time.sleep(pause_s)
    return out


@pytest.mark.timeout(300)
@pytest.mark.parametrize("hop", [128, 256, 512, 1024])
@pytest.mark.parametrize("flags,key", [(o.OUTPUT_PERCUSSIVE, "P"), (o.OUTPUT_HARMONIC, "H")])
def test_resident_kernel_sse_path_vs_oracle(z, hop, flags, key):
    """The same for the causal SSE path (use_sse_filter, hps.cu:582-652; BASELINE configs[4] hop by hop): the single-launch SSE
    kernel's body inside the resident kernel; the history rows of the time box are ring rows the workgroup wrote during
    the hops before.  Back to back, with idle exits, and switching to SSE in mid-stream with the kernel resident."""
    n_hops = 60
    x = _clip(hop * n_hops, 17 + hop)
    ho = o.HPR(FS, hop, 2.0, flags, o.TIME_CAUSAL)
    ho.use_sse_filter()
    ref = ho.process_stream(x)[key]
    io = z.IOGPU(hop)
    rt = z.HPRRealtime(FS, hop, 2.0, flags)
    rt.use_sse_filter()
    copy = rt.copy_percussive if key == "P" else rt.copy_harmonic
    eng = rt.p_impl
    eng.set_resident(200)
    got = _per_hop(z, rt, io, x, hop, n_hops, copy)
    assert np.array_equal(got, ref, equal_nan=True) and np.any(np.nan_to_num(ref) != 0)
    assert eng.resident_stats() == {"launches": 1, "hops_of_ended_launches": 0, "active": True}
    eng.reset_buffers()
    eng.set_resident(4)
    got = _per_hop(z, rt, io, x, hop, n_hops, copy, pause_every=7, pause_s=0.03)
    assert np.array_equal(got, ref, equal_nan=True)
    assert eng.resident_stats()["launches"] >= 1 + 7
    # median path resident, then use_sse_filter() in mid-stream (hps.h:289): the kernel goes home, the SSE one takes over
    rt2 = z.HPRRealtime(FS, hop, 2.0, flags)
    copy2 = rt2.copy_percussive if key == "P" else rt2.copy_harmonic
    rt2.p_impl.set_resident(100)
    h2 = o.HPR(FS, hop, 2.0, flags, o.TIME_CAUSAL)
    first = 25
    ra = h2.process_stream(x[:first * hop])[key]
    h2.use_sse_filter()
    rb = h2.process_stream(x[first * hop:])[key]
    ga = _per_hop(z, rt2, io, x, hop, first, copy2)
    rt2.use_sse_filter()
    gb = _per_hop(z, rt2, io, x[first * hop:], hop, n_hops - first, copy2)
    assert np.array_equal(ga, ra) and np.array_equal(gb, rb, equal_nan=True)
    assert rt2.p_impl.resident_stats()["launches"] == 2


@pytest.mark.timeout(300)
@pytest.mark.parametrize("hop", [128, 256, 512, 1024])
@pytest.mark.parametrize("flags,key", [(o.OUTPUT_PERCUSSIVE, "P"), (o.OUTPUT_HARMONIC, "H")])
def test_resident_kernel_per_hop_calls_vs_oracle(z, hop, flags, key):
    """zen_hip_hpr_set_resident: process_next_hop + copy_* (libzen/hps.cu:334-363, the loop of zen/fakert.h:221-247)
    served by ONE workgroup that stays on the device and takes each hop from a mailbox.  Same samples as the oracle hop for
    hop: back to back; with pauses longer than the idle time (the kernel leaves and is launched again, more than once);
    with a block call, a reset and per-launch hops in between; and two hops posted without a copy between them."""
    n_hops = 72
    x = _clip(hop * n_hops, 11 + hop)
    ref = o.HPR(FS, hop, 2.0, flags, o.TIME_CAUSAL).process_stream(x)[key]
    io = z.IOGPU(hop)
    rt = z.HPRRealtime(FS, hop, 2.0, flags)
    copy = rt.copy_percussive if key == "P" else rt.copy_harmonic
    eng = rt.p_impl
    eng.set_resident(200)
    got = _per_hop(z, rt, io, x, hop, n_hops, copy)
    st = eng.resident_stats()
    assert np.array_equal(got, ref) and np.any(ref != 0)
    assert st["launches"] == 1 and st["active"]
    # the kernel leaves after 5 ms without a hop: pauses of 40 ms every 9 hops
    eng.reset_buffers()
    eng.set_resident(5)
    got = _per_hop(z, rt, io, x, hop, n_hops, copy, pause_every=9, pause_s=0.04)
    assert np.array_equal(got, ref)
    assert eng.resident_stats()["launches"] >= 1 + 8      # (the first part's one; then one per pause but the last)
    # a block call and per-launch hops in the middle of the stream; then resident again
    eng.reset_buffers()
    eng.set_resident(100)
    a = _per_hop(z, rt, io, x, hop, 20, copy)
    blk = eng.process_stream_host(x[20 * hop:40 * hop])[key]                  # (sends the kernel home first)
    eng.set_resident(0)
    b = _per_hop(z, rt, io, x[40 * hop:], hop, 10, copy)
    eng.set_resident(100)
    c = _per_hop(z, rt, io, x[50 * hop:], hop, n_hops - 50, copy)
    assert np.array_equal(np.concatenate([a, blk, b, c]), ref)
    # two hops posted back to back, one copy: the second post waits for the first hop, the copy hands out the second
    eng.reset_buffers()
    io2 = z.IOGPU(hop)
    io.host_in[:] = x[:hop]
    io2.host_in[:] = x[hop:2 * hop]
    rt.process_next_hop(io.device_in)
    rt.process_next_hop(io2.device_in)
    copy(io.device_out)
    assert np.array_equal(io.host_out, ref[hop:2 * hop])
    del rt, eng                                                                # destroy with the kernel resident


# ---------------------------------------------------------------------------- time median + frequency median + mask bits in one launch
@pytest.mark.parametrize("fs,hop", [(44100.0, 256), (48000.0, 256), (44100.0, 512), (48000.0, 512)])
@pytest.mark.parametrize("flags", [ALL, o.OUTPUT_PERCUSSIVE])
def test_time_and_frequency_median_in_one_launch(z, fs, hop, flags):
    """Anticausal blocks with hard masks at the geometries of pass 2 (11 / 13 taps at 44.1 kHz, 13 / 11 at 48 kHz, hop 256;
    7 / 23 and 7 / 21 at hop 512): median_tf_herm_bits_kernel computes the harmonic estimate (time median, mfilt.h:310-314)
    and the percussive one (frequency median, :316-318) of the same rows and writes only the mask bits (hps.cu:501-505,
    :535-540).  Against the oracle, against the two-launch path ("no_median_tf"), two streams, block lengths that are not
    multiples of the kernel's twelve rows, state carried from call to call."""
    ho = o.HPR(fs, hop, 2.0, flags, o.TIME_ANTICAUSAL)
    assert (ho.l_harm | 1, ho.l_perc | 1) in ((11, 13), (13, 11), (7, 23), (7, 21))
    n_hops = 3 * ho.stft_width + 53
    x = np.stack([_clip(hop * n_hops, 3 + hop, ), _clip(hop * n_hops, 4 + hop)])
    refs = [o.HPR(fs, hop, 2.0, flags, o.TIME_ANTICAUSAL).process_stream(x[s]) for s in range(2)]
    keys = [k for k, f in (("P", o.OUTPUT_PERCUSSIVE), ("H", o.OUTPUT_HARMONIC), ("R", o.OUTPUT_RESIDUAL)) if flags & f]

    def run(opt, block):
        if opt:
            z.set_option(opt, 1)
        try:
            return z.HPR(fs, hop, 2.0, flags, z.TIME_ANTICAUSAL, True, 2).process_stream_host(x, block=block)
        finally:
            if opt:
                z.set_option(opt, 0)

    for block in (None, 29, 13, 8):
        got = run(None, block)
        old = run("no_median_tf", block)
        for k in keys:
            for s in range(2):
                assert np.array_equal(got[k][s], refs[s][k]), (block, k, s)
            assert np.array_equal(got[k], old[k]), ("no_median_tf", block, k)
            assert np.any(refs[0][k] != 0)


def test_host_process_refuses_overlapping_buffers(z):
    """HPRIOffline::process takes its clip by value and returns new vectors (hps.cu:128-131): through the C-ABI the four host
    buffers must be distinct -- finished ranges come down while later ranges still go up."""
    n = 50000
    x = _clip(n, 1)
    g = z.HPRIOffline(FS, 1024, 256, 2.0, 2.0)
    buf = np.zeros(2 * n, np.float32)
    buf[:n] = x
    with pytest.raises(z.ZenHipError):
        g.process(buf[:n], out=(buf[n // 2:n // 2 + n], None, None))
    with pytest.raises(z.ZenHipError):
        g.process(x, out=(buf[:n], buf[n - 1:2 * n - 1], None))
    h, p, r = g.process(buf[:n], out=(buf[n:], np.zeros(n, np.float32), None))   # adjacent, not overlapping
    rh, rp, _ = o.HPRIOffline(FS, 1024, 256, 2.0, 2.0).process(x)
    assert np.array_equal(h, rh) and np.array_equal(p, rp)


# ---------------------------------------------------------------------------- the offline passes synthesised in runs
def _runs_expected(n_hops, hop, chunk, groups, run):
    """hpr.hip run_pass_groups: every chunk of the pass at least 8 hops, and a kernel for the transform and the groups.
    None: nfft >= 2048 with the run length left to the engine -- it takes the long workgroups of istft_run_wide_kernel only
    where they fill the device (pick_wide_run: not for one short clip)"""
    log2n = int(np.log2(4 * hop))
    kernel = (8 <= log2n <= 10 and groups == 1) or 11 <= log2n <= 14
    if log2n > 10 and run == 0:
        return None
    return kernel and min(n_hops, chunk or n_hops) >= 8 and (chunk == 0 or n_hops % chunk == 0 or n_hops % chunk >= 8)


@pytest.mark.parametrize("hop_h,hop_p,n", [(4096, 256, 161571), (1024, 256, 70001), (2048, 128, 50000), (4096, 256, 4096 * 9 + 1),
                                           (512, 128, 30000), (4096, 256, 2100), (256, 64, 9000), (4096, 1024, 120000),
                                           (2048, 512, 77777)])
@pytest.mark.parametrize("chunk,run", [(0, 0), (64, 16), (40, 7), (8, 1), (24, 64), (3, 0), (1, 5)])
def test_offline_passes_synthesised_in_runs(z, hop_h, hop_p, n, chunk, run):
    """Both passes of HPRIOffline with hard masks (hps.cu:142-167: H, and P + R summed for pass 2; :185-205: P alone): a
    wavefront (nfft <= 1024) or a workgroup synthesises a run of consecutive frames, adds the overlapping halves in
    registers -- and the two outputs of a group -- and writes the finished hops where the driver wants them (istft.hip
    istft_run_kernel / istft_run_wide_kernel): no Y rows, no overlap-add launch.  Bit-identical to the oracle and to the
    launches it replaces ("no_istft_runs"), with the engines' chunks bounded ("offline_chunk_hops": the carry crosses
    chunk boundaries, a run starts from the frame before it) and runs of 1 .. 64 frames; clips shorter than a chunk or a few
    hops long; a pass falls back to the launches as a whole when one of its chunks is too short for the masks-as-bits road
    or no kernel covers its transform."""
    x = _clip(n, 11 + n % 7)
    rh, rp, rr = o.HPRIOffline(FS, hop_h, hop_p, 2.0, 2.0).process(x)
    z.set_option("offline_chunk_hops", chunk)
    z.set_option("istft_run", run)
    z.set_option("istft_run_wide", run)
    try:
        g = z.HPRIOffline(FS, hop_h, hop_p, 2.0, 2.0)
        for rep in range(2):
            h, p, r = g.process(x)
            assert np.array_equal(h, rh) and np.array_equal(p, rp) and np.all(r == 0), rep
        p_only = np.full(n, np.nan, np.float32)        # the harmonic output left out: pass 1 has one group (P + R)
        g.process(x, out=(None, p_only, None))
        assert np.array_equal(p_only, rp)
        # the new path did run where it can: no overlap-add launch in that pass
        n1, n2 = g.hop_counts(n)
        g.profile(True)
        g.process(x)
        prof = g.profile_get_all()
        g.profile(False)
        for ps, nh, hop, groups in (("pass1", n1, hop_h, 2), ("pass2", n2, hop_p, 1)):
            exp = _runs_expected(nh, hop, chunk, groups, run)
            assert exp is None or (prof[ps]["finalize"]["launches"] == 0) == exp, (ps, nh, chunk, prof[ps])
        z.set_option("no_istft_runs", 1)
        h2, p2, _ = z.HPRIOffline(FS, hop_h, hop_p, 2.0, 2.0).process(x)
        assert np.array_equal(h2, rh) and np.array_equal(p2, rp)
    finally:
        z.set_option("no_istft_runs", 0)
        z.set_option("offline_chunk_hops", 0)
        z.set_option("istft_run", 0)
        z.set_option("istft_run_wide", 0)


def test_offline_batch_of_clips_synthesised_in_runs(z):
    """The same for several clips per call (zen_hip_hpri_process_device with n_clips > 1: one stream per clip, runs never
    cross a clip) against the per-clip oracle."""
    n, clips = 40000, 3
    xs = np.stack([_clip(n, 31 + c) for c in range(clips)])
    z.set_option("istft_run_wide", 5)
    try:
        g = z.HPRIOffline(FS, 1024, 256, 2.0, 2.0, n_clips=clips)
        din, dh, dp = z.DeviceBuffer(clips * n), z.DeviceBuffer(clips * n), z.DeviceBuffer(clips * n)
        din.upload(xs.reshape(-1))
        g.profile(True)
        g.process_device(din.ptr, n, n, harm=dh.ptr, perc=dp.ptr, out_stride=n)
        z.synchronize()
        prof = g.profile_get_all()
        assert prof["pass1"]["finalize"]["launches"] == 0 and prof["pass2"]["finalize"]["launches"] == 0
        H, P = dh.download().reshape(clips, n), dp.download().reshape(clips, n)
        for c in range(clips):
            rh, rp, _ = o.HPRIOffline(FS, 1024, 256, 2.0, 2.0).process(xs[c])
            assert np.array_equal(H[c], rh) and np.array_equal(P[c], rp), c
        for b in (din, dh, dp):
            b.free()
    finally:
        z.set_option("istft_run_wide", 0)


def test_offline_batch_config4_runs_engage_and_match_the_launches(z):
    """BASELINE config 4's shape per GPU (64 clips x 30 s, 4096 / 256, hard masks): with the run length left to the engine
    BOTH passes are synthesised in runs (pass 1: the long workgroups fill the device, hpr.hip pick_wide_run), and the outputs
    are bit-identical to the synthesis + overlap-add launches ("no_istft_runs" = 1), which the full-size property test of
    round 3 pins to the oracle."""
    import bench as b
    n, clips = 1323000, 64
    x = np.stack([b.s_music(n, seed=c) for c in range(4)])
    xs = np.concatenate([x] * (clips // 4))                      # (four different clips, repeated)
    din, dh, dp = z.DeviceBuffer(clips * n), z.DeviceBuffer(clips * n), z.DeviceBuffer(clips * n)
    din.upload(xs.reshape(-1))
    outs = []
    for no_runs in (0, 1):
        z.set_option("no_istft_runs", no_runs)
        try:
            g = z.HPRIOffline(FS, 4096, 256, 2.0, 2.0, n_clips=clips)
            g.profile(True)
            g.process_device(din.ptr, n, n, harm=dh.ptr, perc=dp.ptr, out_stride=n)
            z.synchronize()
            prof = g.profile_get_all()
            for ps in ("pass1", "pass2"):
                assert (prof[ps]["finalize"]["launches"] == 0) == (no_runs == 0), (no_runs, ps, prof[ps])
            outs.append((dh.download(), dp.download()))
            del g
        finally:
            z.set_option("no_istft_runs", 0)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][1][:n], outs[0][1][4 * n:5 * n])   # clip 4 is clip 0 again
    for bfr in (din, dh, dp):
        bfr.free()


def test_ten_minute_clip_runs_across_the_engines_own_chunks(z):
    """One 10-minute mono clip with hard masks, resident in HBM: pass 2 has 103 000 hops, more than the 65 536 an engine takes
    per launch, so the carry of the synthesis in runs crosses a chunk boundary at the engine's own chunk size (the small
    cases above bound the chunks with an option).  Bit-identical to the synthesis + overlap-add launches, and to the oracle
    on windows at the start, around the chunk boundary and at the end (an oracle started warm-up frames early reaches the
    serial state: tests/test_gpu_round3.py)."""
    import bench as b
    n = 26460000
    x = b.s_music(n, seed=5)
    din, dh, dp = z.DeviceBuffer(n), z.DeviceBuffer(n), z.DeviceBuffer(n)
    din.upload(x)
    outs = []
    for no_runs in (0, 1):
        z.set_option("no_istft_runs", no_runs)
        try:
            g = z.HPRIOffline(FS, 4096, 256, 2.0, 2.0)
            g.profile(True)
            g.process_device(din.ptr, n, n, harm=dh.ptr, perc=dp.ptr, out_stride=n)
            z.synchronize()
            prof = g.profile_get_all()
            assert (prof["pass2"]["finalize"]["launches"] == 0) == (no_runs == 0), prof["pass2"]
            if no_runs == 0:
                assert prof["pass2"]["istft"]["launches"] >= 2          # two chunks
            outs.append((dh.download(), dp.download()))
            del g
        finally:
            z.set_option("no_istft_runs", 0)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    # oracle windows (the pattern of tests/test_gpu_round3.py): a slice that starts 12 pass-1 hops early, on a pass-1 hop
    # boundary, reaches the serial state before the window; an output sample depends on a few hops of input beyond it
    H, P = outs[0]
    hop_h, lead, m = 4096, 12 * 4096, int(2 * FS)
    rh, rp, _ = o.HPRIOffline(FS, 4096, 256, 2.0, 2.0).process(x[:int(4 * FS)])
    assert np.array_equal(H[:m], rh[:m]) and np.array_equal(P[:m], rp[:m])
    b0 = 65536 * 256 - (m // 2 // hop_h) * hop_h                 # the window straddles pass 2's chunk boundary
    rh, rp, _ = o.HPRIOffline(FS, 4096, 256, 2.0, 2.0).process(x[b0 - lead:b0 + m + 8 * hop_h])
    assert np.array_equal(H[b0:b0 + m], rh[lead:lead + m]) and np.array_equal(P[b0:b0 + m], rp[lead:lead + m])
    e0 = ((n - m) // hop_h) * hop_h                              # the clip's end, padding and all
    rh, rp, _ = o.HPRIOffline(FS, 4096, 256, 2.0, 2.0).process(x[e0 - lead:])
    assert np.array_equal(H[e0:], rh[lead:]) and np.array_equal(P[e0:], rp[lead:])
    for bfr in (din, dh, dp):
        bfr.free()
