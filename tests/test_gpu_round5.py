"""GPU parity tests added in round 5.  Every call goes through the C-ABI (libzen_hip.so via ctypes) and is compared
BIT-EXACTLY (tolerance 0) with the CPU oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu
ALL = o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE | o.OUTPUT_RESIDUAL
FS = 44100.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def z():
    import zen_amd
    zen_amd.init(0)
    return zen_amd


def _clip(n, seed):
    from tests.test_gpu_parity import music
    return music(n, seed)


# ---------------------------------------------------------------------------- memory checking (memguard)
def test_red_zones_are_on_and_clean(z):
    r = z.memcheck()
    assert r["redzone_bytes"] >= 4096, "tests/conftest.py switches the red zones on before the library's first allocation"
    assert r["corrupt_words"] == 0, r["first_message"]


_POKE_CHILD = """
import json, sys
import zen_amd
zen_amd.init(0)
where = sys.argv[1]
buf = zen_amd.DeviceBuffer(1000)
rep = [zen_amd.memcheck()]
zen_amd.debug_poke(buf.ptr, 4 * 1000 + 256 if where == "behind" else -4, 0x12345678)   # (behind: past the 256-byte alignment slack)
rep.append(zen_amd.memcheck())
rep.append(zen_amd.memcheck())                   # reported once: the zone is repaired
zen_amd.debug_poke(buf.ptr, 4 * 999, 0)          # the last word of the allocation itself: fine
rep.append(zen_amd.memcheck())
print("REPORTS " + json.dumps(rep))
"""


@pytest.mark.parametrize("where", ["behind", "in_front"])
def test_deliberately_broken_store_is_caught(z, where):
    """A store one word outside an allocation (zen_hip_debug_poke: a one-thread kernel) must show up in zen_hip_memcheck:
    as overwritten red-zone words in any build, and as a recorded out-of-bounds access in a -DZEN_HIP_BOUNDS build.  In a
    child process: a process that found a zone overwritten exits with status 86 (next test), and this one must not."""
    import json
    env = dict(os.environ, ZEN_HIP_REDZONE="4096", ZEN_HIP_BOUNDS_TRAP="0")
    r = subprocess.run([sys.executable, "-c", _POKE_CHILD, where], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=300)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("REPORTS ")]
    assert line, (r.returncode, r.stdout[-1000:], r.stderr[-2000:])
    before, after, again, last = json.loads(line[0][8:])
    assert before["redzone_bytes"] >= 4096 and before["corrupt_words"] == 0
    assert after["corrupt_words"] == 1, after
    assert "red zone" in after["first_message"] or after["bounds_violations"] > 0
    if after["bounds_build"]:
        assert after["bounds_violations"] == before["bounds_violations"] + 1
    assert again["corrupt_words"] == 1
    assert last["corrupt_words"] == 1 and last["bounds_violations"] == after["bounds_violations"]
    assert r.returncode == 86


_SHORT_DST_CHILD = """
import json
import numpy as np
import zen_amd
zen_amd.init(0)
rows, cols = 64, 4096
src = zen_amd.DeviceBuffer.from_host(np.random.default_rng(0).random((rows, cols), dtype=np.float32))
dst = zen_amd.DeviceBuffer(rows * cols - 1000)            # 1000 floats short: the last row's stores run 4000 bytes past the end
f = zen_amd.MedianFilterGPU(rows, cols, 47, zen_amd.FREQUENCY)
f.filter(src, dst)
zen_amd.synchronize()
print("REPORT " + json.dumps(zen_amd.memcheck()))
"""


def test_a_product_kernel_writing_past_its_destination_is_caught(z):
    """The same through a PRODUCT kernel of another translation unit (median47_dpp_kernel via zen_hip_mfilt_run) handed a
    destination that is 1000 floats too short -- a caller's bug, here on purpose: the red zone behind the buffer takes the
    stores (any build), and a -DZEN_HIP_BOUNDS build records every one of them with the source line of the store."""
    import json
    env = dict(os.environ, ZEN_HIP_REDZONE="4096", ZEN_HIP_BOUNDS_TRAP="0")
    r = subprocess.run([sys.executable, "-c", _SHORT_DST_CHILD], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=300)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("REPORT ")]
    assert line, (r.returncode, r.stdout[-1000:], r.stderr[-2000:])
    rep = json.loads(line[0][7:])
    assert rep["corrupt_words"] >= 900, rep                 # (1000 floats minus the 256-byte alignment slack behind the buffer)
    assert "back red zone" in rep["first_message"] or rep["bounds_violations"] > 0
    if rep["bounds_build"]:
        assert rep["bounds_violations"] > 0 and "median47" in rep["first_violation"]
    assert r.returncode == 86


def test_child_process_with_a_broken_store_exits_86(z):
    """Child processes of the tier (C++ host tests, CLI) run with the same red zones: one that overwrote a zone does not
    exit with status 0 whatever it thinks of itself (memguard's exit handler)."""
    code = ("import zen_amd\n"
            "zen_amd.init(0)\n"
            "b = zen_amd.DeviceBuffer(64)\n"
            "zen_amd.debug_poke(b.ptr, -8, 1)\n"
            "b.free()\n")
    env = dict(os.environ, ZEN_HIP_REDZONE="4096", ZEN_HIP_BOUNDS_TRAP="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=300)
    assert r.returncode == 86, (r.returncode, r.stderr[-2000:])
    assert "red zone" in r.stderr


def test_copy_into_interior_leaves_the_neighbouring_slots_alone(z):
    """copy_* into the middle slot of a mapped buffer of three (zen/fakert.h:221-247 hands out IOGPU::device_out; a host may
    well keep several hops in one allocation): the slot written is the oracle's hop, the slots on either side keep their
    fill pattern -- checked for every hop, with the percussive and the harmonic output."""
    hop, n_hops = 512, 10
    x = _clip(hop * n_hops, 3)
    ref = o.HPR(FS, hop, 2.0, ALL, o.TIME_CAUSAL).process_stream(x)
    rt = z.HPRRealtime(FS, hop, 2.0, ALL)
    io = z.IOGPU(3 * hop)
    for i in range(n_hops):
        io.host_in[:hop] = x[i * hop:(i + 1) * hop]
        rt.process_next_hop(io.device_in)
        for copy, key in ((rt.copy_percussive, "P"), (rt.copy_harmonic, "H"), (rt.copy_residual, "R")):
            io.host_out[:] = -7.0
            copy(io.device_out + 4 * hop)
            assert np.array_equal(io.host_out[hop:2 * hop], ref[key][i * hop:(i + 1) * hop]), (i, key)
            assert np.all(io.host_out[:hop] == -7.0) and np.all(io.host_out[2 * hop:] == -7.0), (i, key)


# ---------------------------------------------------------------------------- ADVICE round 4
@pytest.mark.timeout(300)
@pytest.mark.parametrize("hop", [256, 1024])
def test_one_hop_block_calls_with_the_resident_kernel_enabled(z, hop):
    """zen_hip_hpr_process with n_hops == 1 while set_resident is on: the block call queues its overlap-add on the engine's
    own stream, so it must not hand the hop to the resident kernel (which works asynchronously on another stream) -- a race
    that returned stale samples.  One-hop block calls, per-hop calls through the mailbox and longer blocks interleaved;
    every hop against the oracle."""
    n_hops = 48
    x = _clip(hop * n_hops, 23 + hop)
    ref = o.HPR(FS, hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL).process_stream(x)["P"]
    io = z.IOGPU(hop)
    rt = z.HPRRealtime(FS, hop, 2.0, z.OUTPUT_PERCUSSIVE)
    eng = rt.p_impl
    eng.set_resident(200)
    got = np.zeros_like(x)
    i = 0
    pattern = ["hop", "hop", "block1", "block1", "hop", "block3", "block1", "hop"]
    k = 0
    while i < n_hops:
        kind = pattern[k % len(pattern)]
        k += 1
        if kind == "hop":
            io.host_in[:] = x[i * hop:(i + 1) * hop]
            rt.process_next_hop(io.device_in)
            rt.copy_percussive(io.device_out)
            got[i * hop:(i + 1) * hop] = io.host_out
            i += 1
        else:
            m = min(int(kind[5:]), n_hops - i)
            got[i * hop:(i + m) * hop] = eng.process_stream_host(x[i * hop:(i + m) * hop])["P"]
            i += m
    assert np.array_equal(got, ref) and np.any(ref != 0)
    assert eng.resident_stats()["launches"] >= 3     # every block call sent the kernel home


@pytest.mark.timeout(600)
@pytest.mark.parametrize("release", [0, 1])
@pytest.mark.parametrize("mode,hop", [("median", 1024), ("median", 256), ("sse", 512), ("median", 2048)])
@pytest.mark.parametrize("resident", [False, True])
def test_publication_of_a_hop_host_poll_stress(z, release, mode, hop, resident):
    """The host polls the sequence word behind a finished hop and copies the hop from mapped memory (hpr.hip copy_output).
    Both publication forms -- system-scope release fence + release store (the default since round 6) and write-through sample
    stores + relaxed flag ("publish_release" 0, ZEN_HIP_PUBLISH_LIGHT=1) -- over several thousand hops, per launch and resident, EVERY sample
    against the oracle: a hop handed over before its samples arrived shows up as a mismatch."""
    n_hops = 3000 if hop <= 512 else (2000 if hop <= 1024 else 1000)      # (hop 2048: the cooperative kernel, rt_wide.hip)
    rng = np.random.default_rng(hop + release)
    x = rng.uniform(-1, 1, hop * n_hops).astype(np.float32)          # every hop different from the one before
    ho = o.HPR(FS, hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL)
    if mode == "sse":
        ho.use_sse_filter()
    ref = ho.process_stream(x)["P"]
    z.set_option("publish_release", release)
    try:
        io = z.IOGPU(hop)
        rt = z.HPRRealtime(FS, hop, 2.0, z.OUTPUT_PERCUSSIVE)
        if mode == "sse":
            rt.use_sse_filter()
        if resident:
            rt.p_impl.set_resident(100)
        got = np.zeros_like(x)
        for i in range(n_hops):
            io.host_in[:] = x[i * hop:(i + 1) * hop]
            rt.process_next_hop(io.device_in)
            rt.copy_percussive(io.device_out)
            got[i * hop:(i + 1) * hop] = io.host_out
        bad = np.flatnonzero(~((got == ref) | (np.isnan(got) & np.isnan(ref))))
        assert bad.size == 0, ("first mismatch in hop", int(bad[0]) // hop, "of", n_hops, bad.size, "samples differ")
    finally:
        z.set_option("publish_release", 1)      # (the default)


def test_pipeline_buffers_are_sized_before_the_first_range(z):
    """zen_hip_hpri_process as a pipeline: range 0 has no warm-up halo, so sized by it the buffers grew again at range 1 (a
    device-wide synchronisation in the middle of the pipeline).  Sized for the largest range up front, the FIRST call on a
    handle and the second one give the oracle's samples (and the second allocates nothing)."""
    n = 4096 * 40 + 123
    x = _clip(n, 9)
    rh, rp, rr = o.HPRIOffline(FS, 1024, 256, 2.0, 2.0).process(x)
    z.set_option("offline_range", 8192 * 4)
    try:
        h = z.HPRIOffline(FS, 1024, 256, 2.0, 2.0)
        for call in range(2):
            a0 = z.memcheck()["allocations"]
            gh, gp, gr = h.process(x)
            assert h.host_stats()["n_ranges"] >= 4
            assert np.array_equal(gh, rh) and np.array_equal(gp, rp) and np.array_equal(gr, rr), call
            if call == 1:
                assert z.memcheck()["allocations"] == a0
    finally:
        z.set_option("offline_range", 0)


# ---------------------------------------------------------------------------- resident kernel at the long hops
def _per_hop(rt, io, x, hop, n_hops, copy, pause_every=0, pause_s=0.0):
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


@pytest.mark.timeout(600)
@pytest.mark.parametrize("hop,fs", [(2048, 44100.0), (4096, 44100.0), (2048, 48000.0), (4096, 48000.0), (2048, 22050.0)])
@pytest.mark.parametrize("flags,key", [(o.OUTPUT_PERCUSSIVE, "P"), (o.OUTPUT_HARMONIC, "H")])
def test_resident_kernel_at_long_hops_vs_oracle(z, hop, fs, flags, key):
    """zen_hip_hpr_set_resident at hop 2048 / 4096 (nfft 8192 / 16384; the reference's published sweep covers them,
    docs/cpu_vs_gpu.png): the COOPERATIVE single-hop kernel (2 / 4 workgroups, five grid barriers per hop) stays on the device,
    workgroup 0 takes each hop from the mailbox and hands its decision to the others (rt_wide.hip rt_wide_resident_kernel).
    Same samples as the oracle hop for hop: back to back; with pauses longer than the idle time (all workgroups leave together
    and are launched again, more than once); with a block call, a reset and per-launch hops in between."""
    n_hops = 44
    x = _clip(hop * n_hops, 31 + hop)
    ref = o.HPR(fs, hop, 2.0, flags, o.TIME_CAUSAL).process_stream(x)[key]
    io = z.IOGPU(hop)
    rt = z.HPRRealtime(fs, hop, 2.0, flags)
    copy = rt.copy_percussive if key == "P" else rt.copy_harmonic
    eng = rt.p_impl
    eng.set_resident(200)
    got = _per_hop(rt, io, x, hop, n_hops, copy)
    st = eng.resident_stats()
    assert np.array_equal(got, ref) and np.any(ref != 0)
    assert st["launches"] == 1 and st["active"]
    # the kernel leaves after 5 ms without a hop: pauses of 40 ms every 9 hops
    eng.reset_buffers()
    eng.set_resident(5)
    got = _per_hop(rt, io, x, hop, n_hops, copy, pause_every=9, pause_s=0.04)
    assert np.array_equal(got, ref)
    assert eng.resident_stats()["launches"] >= 1 + 4
    # per-launch hops, a block call, resident again: the barrier word's account and the vote words survive every change of hands
    eng.reset_buffers()
    eng.set_resident(100)
    a = _per_hop(rt, io, x, hop, 11, copy)
    eng.set_resident(0)
    b = _per_hop(rt, io, x[11 * hop:], hop, 7, copy)                        # per launch (an odd count: the vote parity flips)
    blk = eng.process_stream_host(x[18 * hop:26 * hop])[key]
    eng.set_resident(100)
    c = _per_hop(rt, io, x[26 * hop:], hop, 9, copy)
    eng.set_resident(3)
    d = _per_hop(rt, io, x[35 * hop:], hop, n_hops - 35, copy, pause_every=2, pause_s=0.03)
    assert np.array_equal(np.concatenate([a, b, blk, c, d]), ref)
    del rt, eng                                                             # destroy with the kernels resident


# ---------------------------------------------------------------------------- exact short divisions (exact_div.h)
@pytest.mark.timeout(600)
def test_short_divisions_equal_ieee_division_for_every_float(z):
    """zen_amd/csrc/exact_div.h: the reciprocal and the division by a mask length the SSE kernels use (three instructions each
    instead of the compiler's ~15) against the compiler's IEEE division -- EVERY float in the range the kernels let them handle,
    every divisor 1..255: 4.2e9 reciprocals and 9.7e11 quotients, bit for bit (tools/check_div.hip; under a second on the GPU)."""
    exe = os.path.join(ROOT, "tools", "bin", "check_div")
    if not os.path.exists(exe):
        from zen_amd import build
        build.build_tools()
    r = subprocess.run([exe], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=500)
    assert r.returncode == 0 and "all identical" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]
    assert "recip_exact: 0 of 42" in r.stdout and "div_const_exact: 0 of 96" in r.stdout, r.stdout


# ---------------------------------------------------------------------------- analysis of real frames (rfft_dev.h)
@pytest.mark.parametrize("fs,hop", [(44100.0, 256), (44100.0, 512), (44100.0, 1024), (48000.0, 2048), (44100.0, 4096), (16000.0, 64),
                                    (8000.0, 32), (22050.0, 128)])
@pytest.mark.parametrize("mode", ["hard", "soft", "sse"])
def test_real_input_analysis_equals_the_complex_transform_and_the_oracle(z, fs, hop, mode):
    """Blocks of frames through the analysis kernel that transforms the REAL frame with the Hermitian half of the radix-2 DAG
    (stft_real_kernel, rfft_dev.h; hps.cu:456-465) against the round 1-4 kernel that runs the full complex transform
    ("no_rfft") and against the oracle: identical samples, every transform size 128 ... 16384, all three mask types, state
    carried over block boundaries (blocks of 1, 3, 17 and the rest: the chunk's first frame takes the saved tail, the last
    W - 1 frames of a call keep whole magnitude rows), two streams."""
    try:
        ho = o.HPR(fs, hop, 2.0, ALL if mode == "hard" else o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE, o.TIME_ANTICAUSAL)
    except Exception:
        pytest.skip("the reference refuses this geometry")
    flags = ALL if mode == "hard" else o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE
    n_hops = min(2 * ho.stft_width + 30, 90 if hop >= 2048 else 400)
    rng = np.random.default_rng(int(fs) + hop)
    x = np.stack([_clip(hop * n_hops, 5 + hop), (rng.uniform(-1, 1, hop * n_hops) * (rng.uniform(0, 1, hop * n_hops) < 0.6)).astype(np.float32)])

    def engine():
        g = z.HPR(fs, hop, 2.0, flags, z.TIME_ANTICAUSAL, True, 2, 0)
        if mode == "soft":
            g.use_soft_mask()
        if mode == "sse":
            g.use_sse_filter()
        return g

    def run(g):
        outs = {k: [] for k in "PHR"}
        pos = 0
        for blk in (1, 3, 17, n_hops - 21):
            r = g.process_stream_host(x[:, pos * hop:(pos + blk) * hop])
            for k in outs:
                outs[k].append(r[k])
            pos += blk
        return {k: np.concatenate(v, axis=1) for k, v in outs.items()}

    new = run(engine())
    z.set_option("no_rfft", 1)
    try:
        old = run(engine())
    finally:
        z.set_option("no_rfft", 0)
    keys = "PHR" if mode == "hard" else "PH"
    for s_ in range(2):
        oo = o.HPR(fs, hop, 2.0, flags, o.TIME_ANTICAUSAL)
        if mode == "soft":
            oo.use_soft_mask()
        if mode == "sse":
            oo.use_sse_filter()
        ref = oo.process_stream(x[s_])
        for k in keys:
            assert np.array_equal(new[k][s_], ref[k], equal_nan=True), (k, s_, "real-input analysis vs oracle")
            assert np.array_equal(old[k][s_], ref[k], equal_nan=True), (k, s_, "complex analysis vs oracle")


# ---------------------------------------------------------------------------- single hops of the SSE path over all four SIMDs
@pytest.mark.timeout(600)
@pytest.mark.parametrize("fs,hop,flags,streams", [(44100.0, 128, ALL, 2), (44100.0, 256, o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE, 1),
                                                  (44100.0, 512, o.OUTPUT_PERCUSSIVE, 1), (48000.0, 512, ALL, 3),
                                                  (44100.0, 1024, ALL, 1), (22050.0, 1024, o.OUTPUT_HARMONIC, 2),
                                                  (96000.0, 256, ALL, 1), (16000.0, 128, o.OUTPUT_PERCUSSIVE, 1)])
def test_sse_single_hops_latency_layout_vs_the_two_wavefront_kernel_and_the_oracle(z, fs, hop, flags, streams):
    """One hop of the causal SSE path per call (HPRRealtime::process_next_hop -> apply_sse_filter, hps.cu:429-486, :582-652)
    through rt_sse_lat.hip -- the frame's transform in 4 / 8 / 16 values per thread on 128 or 256 threads (lfft_dev.h: the
    same radix-2 DAG in more, shorter passes), estimates and masks in the registers of the thread that owns the bin, the
    replicate border as a clamped index -- and through rt_sse.hip's kernel (option "no_sse_lat"): both bit-identical to the
    oracle, per launch, mixed with block calls in both directions, and changing hands between the two kernels in mid-stream.
    The time boxes here are 5 to 69 frames long: history of fewer than eight rows, of several batches of eight, and the first
    frames of a stream whose history is clamped at row 0."""
    n_hops = 40
    x = np.stack([_clip(hop * n_hops, 70 + 3 * s + hop, ) for s in range(streams)])
    xs = (lambda a, b: x[:, a * hop:b * hop]) if streams > 1 else (lambda a, b: x[0, a * hop:b * hop])
    refs = []
    for s in range(streams):
        ho = o.HPR(fs, hop, 2.0, flags, o.TIME_CAUSAL)
        ho.use_sse_filter()
        refs.append(ho.process_stream(x[s]))
    keys = [k for k, f in (("P", o.OUTPUT_PERCUSSIVE), ("H", o.OUTPUT_HARMONIC)) if flags & f]

    def run(schedule):
        g = z.HPR(fs, hop, 2.0, flags, z.TIME_CAUSAL, True, streams)
        g.use_sse_filter()
        parts, pos = [], 0
        for n, block, old in schedule:
            z.set_option("no_sse_lat", old)
            try:
                parts.append(g.process_stream_host(xs(pos, pos + n), block=block))
            finally:
                z.set_option("no_sse_lat", 0)
            pos += n
        assert pos == n_hops
        return parts

    for name, schedule in (("new", [(n_hops, 1, 0)]), ("old", [(n_hops, 1, 1)]),
                           ("mixed", [(7, 1, 0), (5, 4, 0), (6, 1, 1), (3, 1, 0), (9, 3, 0), (10, 1, 0)])):
        parts = run(schedule)
        for s in range(streams):
            for k in keys:
                got = np.concatenate([(p[k][s] if streams > 1 else p[k]) for p in parts])
                assert np.array_equal(got, refs[s][k], equal_nan=True), (name, k, s)
                assert np.any(np.nan_to_num(refs[s][k]) != 0)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("hop", [128, 256, 512, 1024])
def test_sse_resident_kernel_in_both_layouts(z, hop):
    """The resident kernel of the SSE path (zen_hip_hpr_set_resident) with the latency layout's body -- twiddles and window
    loaded once per launch, the previous hop and the overlap-add carries kept in registers from hop to hop -- and with
    rt_sse.hip's ("no_sse_lat"): the same samples as the oracle, back to back, with idle exits (the registers' contents are
    then picked up from memory by the next launch), and with per-launch hops of the other layout in between."""
    n_hops = 48
    x = _clip(hop * n_hops, 23 + hop)
    ho = o.HPR(FS, hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL)
    ho.use_sse_filter()
    ref = ho.process_stream(x)["P"]
    for old in (0, 1):
        z.set_option("no_sse_lat", old)
        try:
            io = z.IOGPU(hop)
            rt = z.HPRRealtime(FS, hop, 2.0, o.OUTPUT_PERCUSSIVE)
            rt.use_sse_filter()
            eng = rt.p_impl
            eng.set_resident(200)
            got = _per_hop(rt, io, x, hop, n_hops, rt.copy_percussive)
            assert np.array_equal(got, ref, equal_nan=True), old
            assert eng.resident_stats()["launches"] == 1
            eng.reset_buffers()
            eng.set_resident(4)
            a = _per_hop(rt, io, x, hop, 20, rt.copy_percussive, pause_every=6, pause_s=0.03)
            eng.set_resident(0)
            z.set_option("no_sse_lat", 1 - old)                           # per launch, the other layout
            b = _per_hop(rt, io, x[20 * hop:], hop, 5, rt.copy_percussive)
            z.set_option("no_sse_lat", old)
            eng.set_resident(50)
            c = _per_hop(rt, io, x[25 * hop:], hop, n_hops - 25, rt.copy_percussive)
            assert np.array_equal(np.concatenate([a, b, c]), ref, equal_nan=True), old
            del rt, eng
        finally:
            z.set_option("no_sse_lat", 0)


# ---------------------------------------------------------------------------- single hops of the median path over all four SIMDs
@pytest.mark.timeout(900)
@pytest.mark.parametrize("fs,hop", [(44100.0, 128), (44100.0, 256), (48000.0, 256), (44100.0, 512), (48000.0, 512), (48000.0, 1024),
                                    (44100.0, 1024)])
@pytest.mark.parametrize("flags,soft,streams", [(o.OUTPUT_PERCUSSIVE, False, 1), (ALL, False, 2), (o.OUTPUT_HARMONIC | o.OUTPUT_RESIDUAL, True, 1),
                                                (o.OUTPUT_HARMONIC, False, 3)])
def test_median_single_hops_latency_layout_vs_the_block_kernels_build_and_the_oracle(z, fs, hop, flags, soft, streams):
    """One hop of the causal median path per call (HPRRealtime::process_next_hop, hps.cu:334-339, :429-486, :488-580) through
    rt_hop_lat.hip -- lfft_dev.h's transform on 128 to 512 threads, |S| in an LDS row image of its own, the 47-tap block scheme or
    the sorting-network medians in chunks of 4 / 8, masks by exact comparison -- and through rt_fused.hip's single-hop builds
    (option "no_hop_lat"): bit-identical to the oracle for every (transform size, frequency mask) pair the engine has a
    single-launch kernel for, one to three outputs, hard and soft masks, one to three streams; mixed with block calls in both
    directions and changing hands between the two kernels in mid-stream."""
    n_hops = 36
    x = np.stack([_clip(hop * n_hops, 90 + 5 * s + hop) for s in range(streams)])
    xs = (lambda a, b: x[:, a * hop:b * hop]) if streams > 1 else (lambda a, b: x[0, a * hop:b * hop])
    refs = []
    for s in range(streams):
        ho = o.HPR(fs, hop, 2.0, flags, o.TIME_CAUSAL)
        if soft:
            ho.use_soft_mask()
        refs.append(ho.process_stream(x[s]))
    keys = [k for k, f in (("P", o.OUTPUT_PERCUSSIVE), ("H", o.OUTPUT_HARMONIC), ("R", o.OUTPUT_RESIDUAL)) if flags & f]

    def run(schedule):
        g = z.HPR(fs, hop, 2.0, flags, z.TIME_CAUSAL, True, streams)
        if soft:
            g.use_soft_mask()
        parts, pos = [], 0
        for n, block, old in schedule:
            z.set_option("no_hop_lat", old)
            try:
                parts.append(g.process_stream_host(xs(pos, pos + n), block=block))
            finally:
                z.set_option("no_hop_lat", 0)
            pos += n
        assert pos == n_hops
        return parts

    for name, schedule in (("new", [(n_hops, 1, 0)]), ("old", [(n_hops, 1, 1)]),
                           ("mixed", [(6, 1, 0), (5, 4, 0), (5, 1, 1), (3, 1, 0), (8, 3, 0), (9, 1, 0)])):
        parts = run(schedule)
        for s in range(streams):
            for k in keys:
                got = np.concatenate([(p[k][s] if streams > 1 else p[k]) for p in parts])
                assert np.array_equal(got, refs[s][k]), (name, k, s)
            assert any(np.any(refs[s][k] != 0) for k in keys)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("fs,hop", [(44100.0, 128), (44100.0, 256), (48000.0, 512), (44100.0, 512), (44100.0, 1024), (48000.0, 1024)])
def test_median_resident_kernel_in_both_layouts(z, fs, hop):
    """The resident kernel of the median path (zen_hip_hpr_set_resident) with the latency layout's body (twiddles, window, the
    previous hop and the carries in registers from hop to hop) and with rt_fused.hip's ("no_hop_lat"): the oracle's samples back
    to back, with idle exits, with per-launch hops of the other layout and a block call in between."""
    n_hops = 50
    x = _clip(hop * n_hops, 29 + hop)
    ref = o.HPR(fs, hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL).process_stream(x)["P"]
    for old in (0, 1):
        z.set_option("no_hop_lat", old)
        try:
            io = z.IOGPU(hop)
            rt = z.HPRRealtime(fs, hop, 2.0, o.OUTPUT_PERCUSSIVE)
            eng = rt.p_impl
            eng.set_resident(200)
            got = _per_hop(rt, io, x, hop, n_hops, rt.copy_percussive)
            assert np.array_equal(got, ref), old
            assert eng.resident_stats()["launches"] == 1
            eng.reset_buffers()
            eng.set_resident(4)
            a = _per_hop(rt, io, x, hop, 17, rt.copy_percussive, pause_every=5, pause_s=0.03)
            eng.set_resident(0)
            z.set_option("no_hop_lat", 1 - old)                           # per launch, the other layout
            b = _per_hop(rt, io, x[17 * hop:], hop, 5, rt.copy_percussive)
            z.set_option("no_hop_lat", old)
            blk = eng.process_stream_host(x[22 * hop:29 * hop])["P"]
            eng.set_resident(50)
            c = _per_hop(rt, io, x[29 * hop:], hop, n_hops - 29, rt.copy_percussive)
            assert np.array_equal(np.concatenate([a, b, blk, c]), ref), old
            del rt, eng
        finally:
            z.set_option("no_hop_lat", 0)
