"""GPU parity tests added in round 3: the paths bench.py actually times, at the sizes it times them.

Every call goes through the C-ABI (libzen_hip.so via ctypes) and is compared BIT-EXACTLY (tolerance 0) with the
CPU oracle: on windows of the full-size workloads where the whole run is too long for the oracle, using the
locality of the algorithm (a causal output hop depends on three input hops; an offline output sample on a halo of
a few hops of either pass)."""
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


def bench_module():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


# ---------------------------------------------------------------------------- the headline block (BASELINE configs[1])
@pytest.mark.parametrize("flags,seed_kind", [(o.OUTPUT_PERCUSSIVE, "music"), (ALL, "music"), (o.OUTPUT_PERCUSSIVE, "noise")])
@pytest.mark.parametrize("path", ["fused", "three_kernel"])
def test_headline_block_windows_vs_oracle(z, flags, seed_kind, path):
    """The 25 840-hop block bench.py times (hop 1024, beta 2, causal, hard mask; P only and H+P+R; S-music and
    S-noise), ONE zen_hip_hpr_process call, against the oracle on windows at its start, middle and end.

    Causal output hop i = second half of frame i-1 + first half of frame i, frame i = input hops i-1, i (the causal
    time median is the identity): it depends on input hops i-2 .. i only.  An oracle that starts from
    zero state at hop j is therefore right from output hop j + 2 on (libzen/hps.cu:429-486)."""
    hop, M, K = 1024, 25840, 24
    b = bench_module()
    x = b.s_music(M * hop, seed=0) if seed_kind == "music" else b.s_noise(M * hop, seed=0)
    z.set_option("no_block_fused", 1 if path == "three_kernel" else 0)
    try:
        g = z.HPR(FS, hop, 2.0, flags, z.TIME_CAUSAL, True, 1, M)
        got = g.process_stream_host(x)
    finally:
        z.set_option("no_block_fused", 0)
    keys = "PHR" if flags == ALL else "P"
    for first in (0, M // 2, M - K):
        j = max(first - 2, 0)
        skip = first - j
        ref = o.HPR(FS, hop, 2.0, flags, o.TIME_CAUSAL).process_stream(x[j * hop:(first + K) * hop])
        for k in keys:
            assert np.array_equal(got[k][first * hop:(first + K) * hop], ref[k][skip * hop:]), (first, k)
            assert np.any(ref[k][skip * hop:] != 0)
    if flags != ALL:
        assert np.all(got["H"] == 0) and np.all(got["R"] == 0)        # hps.test.cu:321-343


# ---------------------------------------------------------------------------- BASELINE configs[2]: sharded soft-mask offline
@pytest.mark.parametrize("hop_h,hop_p,n", [(4096, 256, 400000), (1024, 256, 161571)])
def test_offline_time_sharded_soft_mask_equals_whole_clip_and_oracle(z, hop_h, hop_p, n):
    """bench.py --workload offline_long runs process_range + use_soft_mask(): the same code path at a size the
    oracle finishes, for world 2 / 3 / 7, against the whole-clip call AND the oracle."""
    from zen_amd import dist as zdist
    from tests.test_gpu_parity import music
    x = music(n, 78)
    whole = z.HPRIOffline(FS, hop_h, hop_p, 2.5, 2.5)
    whole.use_soft_mask()
    h_ref, p_ref, _ = whole.process(x)
    oo = o.HPRIOffline(FS, hop_h, hop_p, 2.5, 2.5)
    oo.use_soft_mask()
    rh, rp, _ = oo.process(x)
    assert np.array_equal(h_ref, rh) and np.array_equal(p_ref, rp)
    d_in = z.DeviceBuffer.from_host(x)
    for world in (2, 3, 7):
        eng = z.HPRIOffline(FS, hop_h, hop_p, 2.5, 2.5)
        eng.use_soft_mask()
        for rank, (b, e) in enumerate(zdist.time_shards(n, world, hop_h)):
            if b == e:
                continue
            dh, dp = z.DeviceBuffer(e - b), z.DeviceBuffer(e - b)
            eng.process_range(d_in.ptr, n, b, e, dh.ptr, dp.ptr)
            z.synchronize()
            assert np.array_equal(dh.download(), rh[b:e]), (world, rank, "harm")
            assert np.array_equal(dp.download(), rp[b:e]), (world, rank, "perc")


def test_full_size_offline_long_config3_properties(z):
    """BASELINE configs[2] at full size: one channel of the 10-minute clip (26 460 000 samples; the second channel
    is the same code on other data), HPR-I 4096/256, soft mask p = 2.  Too long for the oracle, so: (1) the ranges of a
    7-way time sharding, each computed from its own halo as a different GPU would, equal the whole-clip call bit
    for bit; (2) the first 4 s equal the oracle run on a 6 s prefix (an output sample depends on input up to a few
    hops of either pass beyond it); (3) a range in the middle equals the oracle run on its halo of input."""
    from zen_amd import dist as zdist
    b_ = bench_module()
    n, hop_h, hop_p = int(600 * FS), 4096, 256
    x = b_.s_music(n, seed=9000)
    d_in = z.DeviceBuffer.from_host(x)
    eng = z.HPRIOffline(FS, hop_h, hop_p, 2.5, 2.5)
    eng.use_soft_mask()
    dh, dp = z.DeviceBuffer(n), z.DeviceBuffer(n)
    eng.process_device(d_in.ptr, n, n, dh.ptr, dp.ptr, None, n)
    z.synchronize()
    H, P = dh.download(), dp.download()
    assert np.all(np.isfinite(H)) and np.all(np.isfinite(P)) and np.any(P != 0) and np.any(H != 0)
    del dh, dp
    sh = z.HPRIOffline(FS, hop_h, hop_p, 2.5, 2.5)
    sh.use_soft_mask()
    for rank, (b, e) in enumerate(zdist.time_shards(n, 7, hop_h)):
        rh_d, rp_d = z.DeviceBuffer(e - b), z.DeviceBuffer(e - b)
        sh.process_range(d_in.ptr, n, b, e, rh_d.ptr, rp_d.ptr)
        z.synchronize()
        assert np.array_equal(rh_d.download(), H[b:e]), (rank, "harm")
        assert np.array_equal(rp_d.download(), P[b:e]), (rank, "perc")
        del rh_d, rp_d
    oo = o.HPRIOffline(FS, hop_h, hop_p, 2.5, 2.5)
    oo.use_soft_mask()
    m = int(4 * FS)
    rh, rp, _ = oo.process(x[:int(6 * FS)])
    assert np.array_equal(H[:m], rh[:m]) and np.array_equal(P[:m], rp[:m])
    # a 2 s range in the middle from the oracle: zero state 12 pass-1 hops (and > 2W+2 pass-2 hops) before it is the
    # same warm-up the sharded engine uses (hpri.hip plan_range)
    b0 = (n // 2 // hop_h) * hop_h
    lead, m2 = 12 * hop_h, int(2 * FS)
    oo2 = o.HPRIOffline(FS, hop_h, hop_p, 2.5, 2.5)
    oo2.use_soft_mask()
    rh2, rp2, _ = oo2.process(x[b0 - lead:b0 + m2 + 8 * hop_h])
    assert np.array_equal(H[b0:b0 + m2], rh2[lead:lead + m2]) and np.array_equal(P[b0:b0 + m2], rp2[lead:lead + m2])


# ---------------------------------------------------------------------------- an output switched off in mid-stream
@pytest.mark.parametrize("switch", ["soft", "sse"])
@pytest.mark.parametrize("caus", [o.TIME_CAUSAL, o.TIME_ANTICAUSAL])
@pytest.mark.parametrize("hop", [256, 1024])
def test_dropped_output_drains_like_the_reference(z, switch, caus, hop):
    """use_soft_mask() / use_sse_filter() stop the residual (hps.cu:562, :582-652), but the reference keeps rotating
    its accumulator once per hop (hps.cu:435-449): copy_residual hands out the last finished hop until the next hop,
    the second half of the last frame after that hop (on every copy), zeros from the hop after.  Sequence:
    hops, copy, switch, copy, copy, hop, copy, copy, hop, copy -- through the per-hop API and the oracle."""
    from tests.test_gpu_parity import music
    n_pre = 16                                        # (anticausal hop 256: the first lag = 11 output hops are the zero state)
    x = music(hop * (n_pre + 3), 33)
    ref = o.HPR(FS, hop, 2.0, ALL, caus)
    g = z.HPR(FS, hop, 2.0, ALL, caus, True, 1, 1)
    io = z.IOGPU(hop)
    which = {"P": z.OUTPUT_PERCUSSIVE, "H": z.OUTPUT_HARMONIC, "R": z.OUTPUT_RESIDUAL}

    def hop_both(i):
        ref.process_next_hop(x[i * hop:(i + 1) * hop])
        io.host_in[:hop] = x[i * hop:(i + 1) * hop]
        g.process_next_hop(io.device_in)

    def check(tag):
        want = {"P": ref.percussive_out[:hop].copy(), "H": ref.harmonic_out[:hop].copy(), "R": ref.residual_out[:hop].copy()}
        for k in "RPH":
            g.copy_output(which[k], io.device_out)
            assert np.array_equal(np.asarray(io.host_out[:hop]), want[k]), (tag, k)
        return want

    for i in range(n_pre):
        hop_both(i)
    before = check("before the switch")
    assert np.any(before["R"] != 0)
    getattr(ref, "use_soft_mask" if switch == "soft" else "use_sse_filter")()
    getattr(g, "use_soft_mask" if switch == "soft" else "use_sse_filter")()
    assert np.array_equal(check("after the switch, no hop yet")["R"], before["R"])
    check("again")
    hop_both(n_pre)
    tail = check("one hop after the switch")["R"]
    assert np.any(tail != 0)
    assert np.array_equal(check("repeated copy")["R"], tail)
    hop_both(n_pre + 1)
    assert np.all(check("two hops after the switch")["R"] == 0)
    hop_both(n_pre + 2)
    check("three hops after")


def test_dropped_output_block_calls(z):
    """The same through block calls: the block after the switch starts with the parked tail, then zeros."""
    from tests.test_gpu_parity import music
    hop = 512
    x = music(hop * 30, 34)
    ref = o.HPR(FS, hop, 2.0, ALL, o.TIME_CAUSAL)
    a = ref.process_stream(x[:hop * 11])
    ref.use_soft_mask()
    b = ref.process_stream(x[hop * 11:])
    g = z.HPR(FS, hop, 2.0, ALL, z.TIME_CAUSAL)
    ga = g.process_stream_host(x[:hop * 11], block=4)
    g.use_soft_mask()
    gb = g.process_stream_host(x[hop * 11:], block=7)
    for k in "PHR":
        assert np.array_equal(ga[k], a[k]) and np.array_equal(gb[k], b[k]), k
    assert np.any(b["R"][:hop] != 0) and np.all(b["R"][hop:] == 0)


# ---------------------------------------------------------------------------- engines that grow with their calls
def test_default_engine_grows_with_block_calls(z):
    """max_hops_per_chunk = 0: the buffers start at one hop and grow to the block calls that come, mid-stream, with
    the stft_width-1 history rows and the overlap-add carry carried over (hpr.hip grow_buffers)."""
    from tests.test_gpu_parity import music, same
    for caus in (o.TIME_CAUSAL, o.TIME_ANTICAUSAL):
        hop = 256
        x = music(hop * 200, 35)
        ref = o.HPR(FS, hop, 2.0, ALL, caus).process_stream(x)
        g = z.HPR(FS, hop, 2.0, ALL, caus)                 # default: grows
        din = z.DeviceBuffer.from_host(x)
        outs = {k: z.DeviceBuffer(x.size) for k in "PHR"}
        off = 0
        for m in (1, 1, 5, 2, 40, 1, 3, 100, 47):           # growth at 5, 40, 100; single hops in between
            g.process(din.offset(off * hop), m, x.size, outs["H"].offset(off * hop), outs["P"].offset(off * hop),
                      outs["R"].offset(off * hop), x.size)
            off += m
        z.synchronize()
        assert off == 200
        assert same({k: v.download() for k, v in outs.items()}, ref)


# ---------------------------------------------------------------------------- the reference's bench shapes
@pytest.mark.parametrize("dim", [4096, 16384])
def test_median_bench_squares_full_size_samples_vs_oracle(z, dim):
    """libzen/mfilt.bench.cu:222-232: dim x dim squares, 11 taps, both directions.  The large squares against the
    oracle on samples: frequency direction is row-independent (whole sampled rows), time direction is
    column-independent (16-column strips over all rows); plus the iota data of the bench itself."""
    rng = np.random.default_rng(dim)
    d = rng.uniform(-1, 1, (dim, dim)).astype(np.float32)
    src, dst = z.DeviceBuffer.from_host(d), z.DeviceBuffer(d.size)
    mf = z.MedianFilterGPU(dim, dim, 11, z.FREQUENCY)
    mf.filter(src, dst)
    z.synchronize()
    Pm = dst.download().reshape(dim, dim)
    rows = np.unique(np.concatenate([[0, 1, dim - 1], rng.integers(0, dim, 8)]))
    assert np.array_equal(Pm[rows], o.median_filter(d[rows], 11, o.FREQUENCY))
    for direction, odir in ((z.TIME_ANTICAUSAL, o.TIME_ANTICAUSAL), (z.TIME_CAUSAL, o.TIME_CAUSAL)):
        mt = z.MedianFilterGPU(dim, dim, 11, direction)
        mt.filter(src, dst)
        z.synchronize()
        Hm = dst.download().reshape(dim, dim)
        for c0 in (0, dim - 16, int(rng.integers(0, dim // 16)) * 16):
            strip = np.ascontiguousarray(d[:, c0:c0 + 16])
            assert np.array_equal(Hm[:, c0:c0 + 16], o.median_filter(strip, 11, odir)), (direction, c0)
    # the bench's own data: iota (mfilt.bench.cu:17-32); a monotone ramp is its own median away from the borders
    it = np.arange(dim * dim, dtype=np.float32).reshape(dim, dim)
    src.upload(it)
    mf.filter(src, dst)
    z.synchronize()
    Pi = dst.download().reshape(dim, dim)
    assert np.array_equal(Pi[:, 5:-5], it[:, 5:-5])
    assert np.array_equal(Pi[[0, dim - 1]], o.median_filter(it[[0, dim - 1]], 11, o.FREQUENCY))


def test_fft_32768_bench_size(z):
    """libzen/fftw.bench.cu:231-252 sweeps 2^8 .. 2^15: the largest size, forward / inverse / round trip, bit-exact
    against the oracle's transform and within the reference's own tolerance (fftw.test.cu:16) of a float64 FFT."""
    n = 32768
    rng = np.random.default_rng(15)
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    f = z.FFTC2CWrapperGPU(n)
    f.fft_vec.upload(x)
    f.forward()
    z.synchronize()
    X = f.fft_vec.download()
    assert np.array_equal(X.view(np.float32), o.fft_c2c(x).view(np.float32))
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.abs(X.real - ref.real).max() <= 2e-4 * 2 and np.abs(X.imag - ref.imag).max() <= 2e-4 * 2   # |X| ~ sqrt(n)
    f.backward()                                    # unnormalised: nfft * x
    z.synchronize()
    Y = f.fft_vec.download()
    assert np.array_equal(Y.view(np.float32), o.fft_c2c(X, inverse=True).view(np.float32))
    assert np.abs(Y / n - x).max() <= 2e-4          # fftw.test.cu:16 tolerance
    # batched, through a scratch buffer that grows with the batch
    b = 5
    xb = (rng.uniform(-1, 1, (b, n)) + 1j * rng.uniform(-1, 1, (b, n))).astype(np.complex64)
    d = z.DeviceBuffer.from_host(xb)
    f.exec_batched(d.ptr, b)
    z.synchronize()
    got = d.download().reshape(b, n)
    for i in range(b):
        assert np.array_equal(got[i].view(np.float32), o.fft_c2c(xb[i]).view(np.float32))
    with pytest.raises(z.ZenHipError):
        z.FFTC2CWrapperGPU(65536)


# ---------------------------------------------------------------------------- rt_wide.hip under contention
@pytest.mark.parametrize("hop", [2048, 4096])
def test_long_hop_single_launch_under_contention(z, hop):
    """The cooperative single-hop kernel (nfft 8192 / 16384: the frame spread over 2 / 4 workgroups that meet at
    hand-rolled grid barriers) while ANOTHER stream keeps every CU busy with the headline block kernel: the
    cooperating workgroups must all become resident (no barrier may give up -- copy_* would report it) and the
    hops must equal the oracle's."""
    from tests.test_gpu_parity import music
    n_hops = 160
    x = music(hop * n_hops, 41)
    ref = o.HPR(FS, hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL).process_stream(x)["P"]
    # the contender: blocks of 25 840 hop-1024 frames on its own stream, queued deep enough to last the whole test
    import ctypes as C
    lib = z.load()
    st = C.c_void_p()
    assert lib.zen_hip_stream_create(C.byref(st)) == 0
    M = 25840
    load = z.HPR(FS, 1024, 2.0, z.OUTPUT_PERCUSSIVE, z.TIME_CAUSAL, True, 1, M)
    load.set_stream(st)
    lin, lout = z.DeviceBuffer(M * 1024), z.DeviceBuffer(M * 1024)
    lin.zero()
    rt = z.HPRRealtime(FS, hop, 2.0, z.OUTPUT_PERCUSSIVE)
    io = z.IOGPU(hop)
    out = np.zeros_like(x)
    try:
        for i in range(n_hops):
            if i % 8 == 0:                                  # ~0.6 ms of device-filling work per launch, 8 per 8 hops
                for _ in range(8):
                    load.process(lin.ptr, M, M * 1024, None, lout.ptr, None, M * 1024)
            io.host_in[:hop] = x[i * hop:(i + 1) * hop]
            rt.process_next_hop(io.device_in)
            rt.copy_percussive(io.device_out)               # raises if a grid barrier timed out
            out[i * hop:(i + 1) * hop] = io.host_out[:hop]
    finally:
        lib.zen_hip_synchronize(st)
        load.set_stream(None)
        lib.zen_hip_stream_destroy(st)
    assert np.array_equal(out, ref)


# ---------------------------------------------------------------------------- blocks of frames: hard masks as two bits per bin
@pytest.mark.parametrize("hop,n_hops,soft", [(2048, 1100, False), (4096, 600, False), (4096, 600, True), (2048, 1100, True),
                                             (256, 700, False), (512, 500, False), (1024, 300, False)])
@pytest.mark.parametrize("flags", [ALL, o.OUTPUT_PERCUSSIVE, o.OUTPUT_HARMONIC | o.OUTPUT_RESIDUAL])
def test_blocks_of_frames_mask_bits(z, hop, n_hops, soft, flags):
    """Blocks of >= 8 anticausal frames with hard masks: the two comparisons of every bin are made once -- by the
    frequency-direction median kernel itself (median_big.hip / median_net.hip BITS builds) or by mask_bits_kernel -- and
    the synthesis kernels load two bits per bin (istft.hip MODE 3; hps.cu:501-505, :535-540, :562-567).  Same samples as
    the builds that compare H and P inside the synthesis ("no_mask_bits") and as mask_bits_kernel ("no_median_bits");
    as the oracle on a prefix and -- through the state the engine carries from call to call -- on a second, short call."""
    from tests.test_gpu_parity import music
    if soft and flags == (o.OUTPUT_HARMONIC | o.OUTPUT_RESIDUAL):
        pytest.skip("soft masks: all outputs and the percussive one alone")
    x = music(hop * n_hops, 60 + hop)
    keys = [k for k, f in (("P", o.OUTPUT_PERCUSSIVE), ("H", o.OUTPUT_HARMONIC), ("R", o.OUTPUT_RESIDUAL)) if flags & f]
    if soft:
        keys = [k for k in keys if k != "R"]

    def run(opt):
        if opt:
            z.set_option(opt, 1)
        try:
            g = z.HPR(FS, hop, 2.0, flags, z.TIME_ANTICAUSAL)
            if soft:
                g.use_soft_mask()
            a = g.process_stream_host(x[:hop * (n_hops - 30)])      # one block: mask bits (hard masks)
            b = g.process_stream_host(x[hop * (n_hops - 30):hop * (n_hops - 5)])
            c = g.process_stream_host(x[hop * (n_hops - 5):])       # 5 hops: fewer than 8 frames, no mask bits
        finally:
            if opt:
                z.set_option(opt, 0)
        return {k: np.concatenate([a[k], b[k], c[k]]) for k in keys}

    got = run(None)
    if soft:                     # (soft masks: the median kernels leave the mask values themselves; "no_mask_bits": H and P)
        other = run("no_mask_bits")
        for k in keys:
            assert np.array_equal(got[k], other[k]), ("no_mask_bits", k)
    else:
        for opt in ("no_mask_bits", "no_median_bits", "no_median_tf"):
            other = run(opt)
            for k in keys:
                assert np.array_equal(got[k], other[k]), (opt, k)
    m = 36
    ho = o.HPR(FS, hop, 2.0, flags, o.TIME_ANTICAUSAL)
    if soft:
        ho.use_soft_mask()
    ref = ho.process_stream(x[:hop * m])
    for k in keys:
        assert np.array_equal(got[k][:hop * m], ref[k]), k
    # the tail of the stream against an oracle that starts a few hops earlier (state = the last stft_width + 1 hops)
    j0 = n_hops - 60
    ho2 = o.HPR(FS, hop, 2.0, flags, o.TIME_ANTICAUSAL)
    if soft:
        ho2.use_soft_mask()
    ref2 = ho2.process_stream(x[hop * j0:])
    for k in keys:
        assert np.array_equal(got[k][hop * (j0 + 24):], ref2[k][hop * 24:]), k


# ---------------------------------------------------------------------------- the fused block kernel that finishes hops itself
@pytest.mark.parametrize("flags", [o.OUTPUT_PERCUSSIVE, ALL, o.OUTPUT_HARMONIC | o.OUTPUT_RESIDUAL])
@pytest.mark.parametrize("streams,blocks", [(1, [25840, 25840, 7, 1, 300, 2, 1031, 129, 128, 127]), (3, [50, 700, 1, 9, 1025, 260]),
                                            (8, [3000, 17, 513])])
def test_fused_block_kernel_finishes_hops_itself(z, streams, blocks, flags):
    """Block calls of the headline configuration (hop 1024, P only, hard mask): the workgroup of a hop adds up the hop of
    the workgroup 128 items before it in its XCD's run from the L2 they share and writes it to the caller's buffer; what
    it cannot finish (first hops, ends of runs, rows not yet published) a fix-up launch does (rt_fused.hip).  Whatever the
    block sizes, stream counts (runs that cross stream boundaries) and call sequence: the samples of the same calls with
    the plain overlap-add launch ("no_direct_out"), which the other tests pin to the oracle -- checked here on a prefix
    too -- and, 40 times over, of itself (the hand-off is timing dependent: a wrong ordering would show as a flicker)."""
    hop = 1024
    n_hops = sum(blocks)
    rng = np.random.default_rng(streams)
    x = rng.uniform(-1, 1, (streams, hop * n_hops)).astype(np.float32)

    def run(no_direct):
        z.set_option("no_direct_out", no_direct)
        try:
            g = z.HPR(FS, hop, 2.0, flags, z.TIME_CAUSAL, True, streams, max(blocks))
            din = z.DeviceBuffer.from_host(x)
            outs = {k: z.DeviceBuffer(x.size) for k in "PHR"}
            off = 0
            for m in blocks:
                g.process(din.offset(off * hop), m, x.shape[1], outs["H"].offset(off * hop), outs["P"].offset(off * hop),
                          outs["R"].offset(off * hop), x.shape[1])
                off += m
            z.synchronize()
            return np.stack([outs[k].download().reshape(streams, -1) for k in "PHR"])
        finally:
            z.set_option("no_direct_out", 0)

    plain = run(1)
    ref = o.HPR(FS, hop, 2.0, flags, o.TIME_CAUSAL).process_stream(x[0][:hop * 30])
    for i, k in enumerate("PHR"):
        assert np.array_equal(plain[i][0][:hop * 30], ref[k]), k
    for rep in range(30 if (streams == 1 and flags == o.OUTPUT_PERCUSSIVE) else 6):
        assert np.array_equal(run(0), plain), rep


# ---------------------------------------------------------------------------- `zen offline --sse`: the two-pass path with box filters
@pytest.mark.parametrize("hop_h,hop_p,n", [(4096, 256, 150000), (1024, 256, 50000), (2048, 512, 90001)])
def test_offline_sse_filter_whole_and_sharded(z, hop_h, hop_p, n):
    """HPRIOffline::use_sse_filter (hps.cu:95-100; zen/offline.h --sse): both passes on the SSE (box-filter) path -- no
    residual, so pass 2's input is P + the reference's all-zero residual accumulator (hps.cu:153-160) -- whole clip and
    time-sharded for three ranks, against the oracle."""
    from zen_amd import dist as zdist
    from tests.test_gpu_parity import music
    x = music(n, 91)
    oo = o.HPRIOffline(FS, hop_h, hop_p, 2.0, 2.0)
    oo.use_sse_filter()
    rh, rp, rr = oo.process(x)
    g = z.HPRIOffline(FS, hop_h, hop_p, 2.0, 2.0)
    g.use_sse_filter()
    h, p, r = g.process(x)
    assert np.array_equal(h, rh) and np.array_equal(p, rp) and np.array_equal(r, rr)
    assert np.any(p != 0) and np.any(h != 0)
    d_in = z.DeviceBuffer.from_host(x)
    eng = z.HPRIOffline(FS, hop_h, hop_p, 2.0, 2.0)
    eng.use_sse_filter()
    for rank, (b, e) in enumerate(zdist.time_shards(n, 3, hop_h)):
        if b == e:
            continue
        dh, dp = z.DeviceBuffer(e - b), z.DeviceBuffer(e - b)
        eng.process_range(d_in.ptr, n, b, e, dh.ptr, dp.ptr)
        z.synchronize()
        assert np.array_equal(dh.download(), rh[b:e]) and np.array_equal(dp.download(), rp[b:e]), rank
