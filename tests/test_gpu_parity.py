"""GPU parity tests: every call goes through the C-ABI (libzen_hip.so via ctypes) and is compared
BIT-EXACTLY with the CPU oracle on the same seeded inputs (float waveforms included: the engine
reproduces the oracle's float32 dataflow operation for operation, so the north-star tolerance of
1e-5 relative RMS is met with margin 0)."""
import os

import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu
ALL = o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE | o.OUTPUT_RESIDUAL


@pytest.fixture(scope="module")
def z():
    import zen_amd
    zen_amd.init(0)
    return zen_amd


@pytest.fixture(params=["net", "net_plain47", "general", "dpp47_direct", "net47"])
def zk(z, request):
    """All median kernels: the defaults (sorting networks; median47_dpp_kernel for 47 taps on 4096-bin rows),
    the same with direct stores (dpp47_direct), the generic network kernel for that shape too (net47: with
    its own DPP exchange of sorted blocks, net_plain47: without), and the general wave kernel."""
    z.set_option("median_general", 1 if request.param == "general" else 0)
    z.set_option("no_median47_dpp", 1 if request.param in ("net47", "net_plain47") else 0)
    z.set_option("no_median47_neighbour", 1 if request.param == "net_plain47" else 0)
    z.set_option("median47_variant", 1 if request.param == "dpp47_direct" else 0)
    yield z
    z.set_option("median_general", 0)
    z.set_option("no_median47_dpp", 0)
    z.set_option("no_median47_neighbour", 0)
    z.set_option("median47_variant", 0)


def noise(n, seed=0):
    return np.random.default_rng(seed).uniform(-1, 1, n).astype(np.float32)


def music(n, seed=0, fs=44100.0):
    """S-music of BASELINE.md: four sines + decaying clicks every 0.25 s + a little noise."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    x = sum(0.2 * np.sin(2 * np.pi * f * t) for f in (220.0, 440.0, 660.0, 1320.0))
    step = int(0.25 * fs)
    env = np.exp(-np.arange(int(0.005 * fs)) / (0.001 * fs))
    for s in range(0, n, step):
        m = min(env.size, n - s)
        x[s:s + m] += 0.9 * env[:m] * rng.uniform(-1, 1, m)
    return (x + 0.01 * rng.uniform(-1, 1, n)).astype(np.float32)


# ---------------------------------------------------------------------------- FFT
@pytest.mark.parametrize("n", [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_fft_bit_exact(z, n):
    rng = np.random.default_rng(n)
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    f = z.FFTC2CWrapperGPU(n)
    f.fft_vec.upload(x)
    f.forward()
    z.synchronize()
    fwd = f.fft_vec.download()
    ref = o.fft_c2c(x)
    assert np.array_equal(fwd.view(np.float32), ref.view(np.float32))
    f.backward()                       # unnormalised: nfft * x  (fftw.h:40-43, IPP_FFT_NODIV_BY_ANY)
    z.synchronize()
    rt = f.fft_vec.download()
    assert np.array_equal(rt.view(np.float32), o.fft_c2c(ref, inverse=True).view(np.float32))
    assert np.abs(rt / n - x).max() <= 2e-4    # fftw.test.cu:16 tolerance


def test_fft_batched_and_reference_tolerance(z):
    n, b = 1024, 37
    rng = np.random.default_rng(1)
    x = (rng.uniform(-1, 1, (b, n)) + 1j * rng.uniform(-1, 1, (b, n))).astype(np.complex64)
    f = z.FFTC2CWrapperGPU(n)
    d = z.DeviceBuffer.from_host(x)
    f.exec_batched(d.ptr, b)
    z.synchronize()
    got = d.download().reshape(b, n)
    ref = np.fft.fft(x.astype(np.complex128), axis=1)
    assert np.abs(got.real - ref.real).max() <= 2e-4 and np.abs(got.imag - ref.imag).max() <= 2e-4
    for i in (0, 17, 36):
        assert np.array_equal(got[i].view(np.float32), o.fft_c2c(x[i]).view(np.float32))


def test_fft_rejects_bad_sizes(z):
    with pytest.raises(z.ZenHipError):
        z.FFTC2CWrapperGPU(48)
    with pytest.raises(z.ZenHipError):
        z.FFTC2CWrapperGPU(65536)                     # 32768, the top of fftw.bench.cu's sweep, is served (test_gpu_round3.py)


# ---------------------------------------------------------------------------- median filter
def stripes(x, y):
    d = np.zeros((x, y), np.float32)
    d[x // 2, :] = 5
    d[:, y // 2] = 8
    return d


@pytest.mark.parametrize("x,y,f", [(9, 9, 3), (10, 20, 5), (1024, 128, 5), (1024, 17, 5), (1024, 1024, 21)])
def test_median_reference_stripe_vectors(zk, x, y, f):
    """libzen/mfilt.test.cu fixtures; replicate border => the 'everywhere' expectations (:701-886)."""
    d = stripes(x, y)
    exp_t = np.zeros((x, y), np.float32)
    exp_t[:, y // 2] = 8
    exp_f = np.zeros((x, y), np.float32)
    exp_f[x // 2, :] = 5
    for direction in (zk.TIME_CAUSAL, zk.TIME_ANTICAUSAL):
        assert np.array_equal(zk.MedianFilterGPU(x, y, f, direction).filter_host(d), exp_t)
    assert np.array_equal(zk.MedianFilterGPU(x, y, f, zk.FREQUENCY).filter_host(d), exp_f)
    assert np.array_equal(zk.MedianFilterGPU(x, y, f, zk.FREQUENCY, True).filter_host(d), exp_f)  # copy_bord


@pytest.mark.parametrize("x,y,f", [(9, 9, 3), (10, 20, 5), (1024, 128, 5)])
def test_box_reference_column_vectors(z, x, y, f):
    """The enabled cases of libzen/box.test.cu (:124-199): reciprocal, time-direction box, reciprocal of a
    matrix whose middle column is 8 -> 8*(f+1) on the column, 0 (from +inf) elsewhere."""
    from test_oracle_golden import box_column_case
    rec, exp = box_column_case(x, y, f)
    res = z.BoxFilterGPU(x, y, f, z.TIME_CAUSAL).filter_host(rec)
    with np.errstate(divide="ignore"):
        back = (np.float32(f + 1.0) / res).astype(np.float32)
    assert np.array_equal(back, exp)
    assert np.array_equal(res, o.box_filter(rec, f, o.TIME_CAUSAL))


def test_median_filter_too_big_throws(z):
    for direction in (z.FREQUENCY, z.TIME_CAUSAL, z.TIME_ANTICAUSAL):   # mfilt.test.cu:525-534
        with pytest.raises(z.ZgException):
            z.MedianFilterGPU(9, 9, 171, direction)
        with pytest.raises(z.ZgException):
            z.BoxFilterGPU(9, 9, 171, direction)


@pytest.mark.parametrize("shape,flen", [
    ((6, 4096), 47), ((22, 1024), 13), ((22, 1024), 11), ((12, 2048), 23), ((12, 2048), 7), ((2, 8192), 93),
    ((2, 16384), 187), ((24, 1024), 11), ((7, 33), 3), ((300, 257), 21), ((64, 64), 11), ((5, 5), 5),
    ((1, 9), 1), ((3, 1000), 255), ((260, 70), 255), ((129, 300), 129), ((200, 130), 65),
    ((70, 4097), 47), ((33, 5000), 63), ((100, 37), 9), ((77, 1030), 31), ((64, 8192), 15), ((300, 12), 3),
    ((41, 2050), 25), ((1000, 260), 33), ((90, 1026), 61), ((513, 1028), 5),
    ((37, 4096), 47), ((19, 8192), 47), ((50, 12), 47), ((23, 100), 47), ((9, 12288), 47)])
def test_median_random_bit_exact(zk, shape, flen):
    rng = np.random.default_rng(flen + shape[0])
    d = rng.uniform(0, 10, shape).astype(np.float32)
    d[rng.integers(0, shape[0], 7), rng.integers(0, shape[1], 7)] = 0.0
    if flen <= shape[1]:
        got = zk.MedianFilterGPU(shape[0], shape[1], flen, zk.FREQUENCY).filter_host(d)
        assert np.array_equal(got, o.median_filter(d, flen, o.FREQUENCY))
    if flen <= shape[0]:
        got = zk.MedianFilterGPU(shape[0], shape[1], flen, zk.TIME_ANTICAUSAL).filter_host(d)
        assert np.array_equal(got, o.median_filter(d, flen, o.TIME_ANTICAUSAL))


@pytest.mark.parametrize("shape,flen", [
    ((3, 16384), 187), ((3, 16384), 171), ((5, 8192), 93), ((5, 8192), 85), ((2, 4100), 187), ((4, 5000), 93),
    ((3, 188), 187), ((3, 88), 85), ((2, 180), 171), ((2, 172), 171), ((7, 12288), 187), ((2, 16386), 187), ((2, 4097), 93)])
def test_median_long_frequency_masks(z, shape, flen):
    """The block-merge kernel (median_big.hip: 85/93/171/187 taps) against the oracle and the general wave
    kernel: segment seams (cols > 4096), rows shorter than the mask, signed values, ragged widths that take
    the general kernel instead."""
    rng = np.random.default_rng(flen * 7 + shape[1])
    for signed in (False, True):
        d = rng.standard_normal(shape).astype(np.float32) if signed else rng.random(shape, dtype=np.float32)
        d[0, : min(40, shape[1])] = 0.25     # ties
        exp = o.median_filter(d, flen, o.FREQUENCY)
        z.set_option("median_general", 0)
        assert np.array_equal(z.MedianFilterGPU(shape[0], shape[1], flen, z.FREQUENCY).filter_host(d), exp)
        z.set_option("median_general", 1)
        try:
            assert np.array_equal(z.MedianFilterGPU(shape[0], shape[1], flen, z.FREQUENCY).filter_host(d), exp)
        finally:
            z.set_option("median_general", 0)


def test_median_signed_values_and_even_length(zk):
    rng = np.random.default_rng(9)
    d = rng.normal(0, 3, (40, 200)).astype(np.float32)
    assert np.array_equal(zk.MedianFilterGPU(40, 200, 10, zk.FREQUENCY).filter_host(d),
                          o.median_filter(d, 10, o.FREQUENCY))          # 10 -> 11 (mfilt.h:89)
    assert np.array_equal(zk.MedianFilterGPU(40, 200, 40, zk.TIME_CAUSAL).filter_host(d),
                          o.median_filter(d, 40, o.TIME_CAUSAL))        # 40 -> 41 on 40 rows


def test_median_bench_squares_iota(zk):
    """libzen/mfilt.bench.cu:7-8,17-32: dim x dim, filter 11, iota data, both directions."""
    for dim in (32, 256, 1024):
        d = np.arange(dim * dim, dtype=np.float32).reshape(dim, dim)
        for direction in (zk.FREQUENCY, zk.TIME_ANTICAUSAL):
            assert np.array_equal(zk.MedianFilterGPU(dim, dim, 11, direction).filter_host(d),
                                  o.median_filter(d, 11, direction))


# ---------------------------------------------------------------------------- box filter
@pytest.mark.parametrize("shape,flen", [
    ((12, 2048), 23), ((12, 2048), 7), ((22, 1024), 13), ((2, 4096), 187), ((300, 70), 21), ((513, 1030), 5),
    ((260, 33), 255), ((200, 200), 179), ((700, 130), 65), ((5, 5000), 93), ((40, 1), 7), ((1, 40), 9), ((64, 64), 1)])
def test_box_bit_exact(z, shape, flen):
    """Tiled kernels (row segments of 1024 / 64-column tiles with halo rows), ragged widths, tile seams,
    masks up to 255 taps and the direct kernel behind them, against the oracle's ascending-order sums."""
    rng = np.random.default_rng(flen)
    d = rng.uniform(0, 10, shape).astype(np.float32)
    if flen <= shape[1]:
        assert np.array_equal(z.BoxFilterGPU(shape[0], shape[1], flen, z.FREQUENCY).filter_host(d),
                              o.box_filter(d, flen, o.FREQUENCY))
    if flen <= shape[0]:
        for direction in (o.TIME_CAUSAL, o.TIME_ANTICAUSAL):
            assert np.array_equal(z.BoxFilterGPU(shape[0], shape[1], flen, direction).filter_host(d),
                                  o.box_filter(d, flen, direction))


# ---------------------------------------------------------------------------- HPR engine
def run_oracle(fs, hop, beta, flags, caus, x, sse=False, soft=False):
    h = o.HPR(fs, hop, beta, flags, caus)
    if sse:
        h.use_sse_filter()
    if soft:
        h.use_soft_mask()
    return h, h.process_stream(x)


def same(a, b):
    return all(np.array_equal(a[k], b[k]) for k in "PHR")


@pytest.mark.parametrize("fs,hop,n_hops", [(44100.0, 256, 60), (44100.0, 512, 40), (44100.0, 1024, 30),
                                           (44100.0, 2048, 12), (44100.0, 4096, 8), (48000.0, 256, 60),
                                           (48000.0, 64, 130)])
@pytest.mark.parametrize("caus", [o.TIME_CAUSAL, o.TIME_ANTICAUSAL])
def test_hpr_params_and_stream_bit_exact(z, fs, hop, n_hops, caus):
    x = music(hop * n_hops, seed=hop, fs=fs) if hop >= 512 else noise(hop * n_hops, seed=hop)
    ho, ref = run_oracle(fs, hop, 2.0, ALL, caus, x)
    g = z.HPR(fs, hop, 2.0, ALL, caus)
    assert (g.nwin, g.nfft, g.stft_width, g.l_harm, g.l_perc, g.lag) == \
           (ho.nwin, ho.nfft, ho.stft_width, ho.l_harm, ho.l_perc, ho.lag)
    assert g.cola_factor == ho.cola_factor
    got = g.process_stream_host(x)
    assert same(got, ref)
    # hard masks really are exercised: the three outputs are distinct and non-trivial
    assert np.any(ref["P"] != 0) and np.any(ref["H"] != 0)


def test_hpr_blocking_is_invisible(z):
    """n_hops=1 calls (the reference's process_next_hop), odd block sizes and one big block agree."""
    fs, hop, n_hops = 44100.0, 256, 75
    x = noise(hop * n_hops, 3)
    _, ref = run_oracle(fs, hop, 2.5, ALL, o.TIME_ANTICAUSAL, x)
    for block, chunk in ((1, 1), (7, 4), (n_hops, 0), (n_hops, 16)):
        g = z.HPR(fs, hop, 2.5, ALL, z.TIME_ANTICAUSAL, True, 1, chunk)
        assert same(g.process_stream_host(x, block=block), ref), (block, chunk)


def test_hpr_realtime_api_mapped_memory(z):
    """HPRRealtime<GPU> as zen/fakert.h:221-247 drives it: mapped host_in -> process -> copy -> host_out."""
    fs, hop, n_hops = 44100.0, 1024, 25
    x = music(hop * n_hops, 5)
    _, ref = run_oracle(fs, hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL, x)
    rt = z.HPRRealtime(fs, hop, 2.0, z.OUTPUT_PERCUSSIVE)
    io = z.IOGPU(hop)
    out = np.zeros_like(x)
    for i in range(n_hops):
        io.host_in[:] = x[i * hop:(i + 1) * hop]
        rt.process_next_hop(io.device_in)
        rt.copy_percussive(io.device_out)          # synchronises: host_out is readable
        out[i * hop:(i + 1) * hop] = io.host_out
    assert np.array_equal(out, ref["P"])
    # perc-only leaves H and R accumulators zero (hps.test.cu:321-343)
    rt.copy_harmonic(io.device_out)
    assert np.all(io.host_out == 0)
    rt.copy_residual(io.device_out)
    assert np.all(io.host_out == 0)


def test_hpr_realtime_copy_into_interior_and_alternating_destinations(z):
    """copy_* polls the engine's staging buffer and copies on the host when the destination is mapped host memory:
    the host alias must be right for pointers INSIDE an allocation and for a destination that changes from hop to
    hop; a device destination takes the asynchronous device copy instead."""
    fs, hop, n_hops = 44100.0, 512, 12
    x = music(hop * n_hops, 6)
    _, ref = run_oracle(fs, hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL, x)
    rt = z.HPRRealtime(fs, hop, 2.0, z.OUTPUT_PERCUSSIVE)
    io = z.IOGPU(3 * hop)                            # three hop-sized slots in one mapped buffer
    dev = z.DeviceBuffer(hop)
    out = np.zeros_like(x)
    io.host_out[:] = -7.0
    for i in range(n_hops):
        io.host_in[:hop] = x[i * hop:(i + 1) * hop]
        rt.process_next_hop(io.device_in)
        if i % 4 == 3:                               # a plain device destination
            rt.copy_percussive(dev.ptr)
            out[i * hop:(i + 1) * hop] = dev.download()
            continue
        slot = i % 3
        rt.copy_percussive(io.device_out + 4 * hop * slot)
        out[i * hop:(i + 1) * hop] = io.host_out[slot * hop:(slot + 1) * hop]
    assert np.array_equal(out, ref["P"])


def test_hpr_reset_gives_identical_rerun(z):
    fs, hop = 48000.0, 256                          # hps.test.cu:345-372
    x = noise(hop * 30, 2)
    g = z.HPR(fs, hop, 2.0, z.OUTPUT_PERCUSSIVE, z.TIME_CAUSAL)
    a = g.process_stream_host(x)["P"]
    g.reset_buffers()
    b = g.process_stream_host(x)["P"]
    assert np.array_equal(a, b)


@pytest.mark.parametrize("caus", [o.TIME_CAUSAL, o.TIME_ANTICAUSAL])
@pytest.mark.parametrize("hop", [256, 512])
def test_hpr_soft_mask(z, hop, caus):
    x = music(hop * 40, 7)
    _, ref = run_oracle(44100.0, hop, 2.5, ALL, caus, x, soft=True)
    g = z.HPR(44100.0, hop, 2.5, ALL, caus)
    g.use_soft_mask()
    got = g.process_stream_host(x)
    assert same(got, ref) and np.all(got["R"] == 0)


@pytest.mark.parametrize("caus", [o.TIME_CAUSAL, o.TIME_ANTICAUSAL])
@pytest.mark.parametrize("hop", [512, 256])
def test_hpr_sse_filter(z, hop, caus):
    """config 5: SSE (box) path, nocopybord (a no-op under CPU semantics)."""
    x = music(hop * 40, 8)
    _, ref = run_oracle(44100.0, hop, 2.0, ALL, caus, x, sse=True)
    g = z.HPR(44100.0, hop, 2.0, ALL, caus, False)
    g.use_sse_filter()
    got = g.process_stream_host(x, block=9)
    assert same(got, ref)


@pytest.mark.parametrize("hop,first,block", [(512, 16, 16), (1024, 12, 5), (256, 30, 30), (512, 7, 1)])
def test_hpr_switch_to_sse_after_block_calls(z, hop, first, block):
    """use_sse_filter() may be called at any time (hps.h:289).  The causal median path runs blocks of hops
    through the fused kernel, which keeps no rings; the hop after the switch filters the magnitudes of the
    stft_width-1 frames before it (causal time box filter), so those rows must be there all the same."""
    n_hops = first + 14
    x = music(hop * n_hops, 21)
    h = o.HPR(44100.0, hop, 2.0, ALL, o.TIME_CAUSAL)
    ref_a = h.process_stream(x[:first * hop])
    h.use_sse_filter()
    ref_b = h.process_stream(x[first * hop:])
    g = z.HPR(44100.0, hop, 2.0, ALL, z.TIME_CAUSAL)
    got_a = g.process_stream_host(x[:first * hop], block=block)
    g.use_sse_filter()
    got_b = g.process_stream_host(x[first * hop:], block=3)
    assert same(got_a, ref_a)
    assert same(got_b, ref_b)


@pytest.mark.parametrize("fs,hop,flags,streams", [(44100.0, 512, ALL, 1), (44100.0, 1024, o.OUTPUT_PERCUSSIVE, 1),
                                                  (48000.0, 256, o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE, 1),
                                                  (44100.0, 128, ALL, 2), (22050.0, 512, o.OUTPUT_HARMONIC, 1)])
def test_hpr_sse_hop_by_hop_single_launch(z, fs, hop, flags, streams):
    """config 5 through the realtime API: every hop of the causal SSE path is ONE launch (rt_sse.hip), mixed with
    block calls (four launches): same samples as the oracle either way, history handed over through the
    magnitude ring in both directions."""
    n_hops = 30
    x = np.stack([music(hop * n_hops, 40 + s, fs) for s in range(streams)])
    g = z.HPR(fs, hop, 2.0, flags, z.TIME_CAUSAL, True, streams)
    g.use_sse_filter()
    parts = [g.process_stream_host(x[:, :hop * 9] if streams > 1 else x[0, :hop * 9], block=1),     # single launches
             g.process_stream_host(x[:, hop * 9:hop * 16] if streams > 1 else x[0, hop * 9:hop * 16], block=4),
             g.process_stream_host(x[:, hop * 16:] if streams > 1 else x[0, hop * 16:], block=1)]
    for s in range(streams):
        _, ref = run_oracle(fs, hop, 2.0, flags, o.TIME_CAUSAL, x[s], sse=True)
        for k in "PHR":
            got = np.concatenate([(p[k][s] if streams > 1 else p[k]) for p in parts])
            assert np.array_equal(got, ref[k]), (k, s)
    z.set_option("no_rt_fused", 1)                    # the four-launch path hop by hop, for the same stream
    try:
        g2 = z.HPR(fs, hop, 2.0, flags, z.TIME_CAUSAL, True, streams)
        g2.use_sse_filter()
        alt = g2.process_stream_host(x if streams > 1 else x[0], block=1)
    finally:
        z.set_option("no_rt_fused", 0)
    for s in range(streams):
        _, ref = run_oracle(fs, hop, 2.0, flags, o.TIME_CAUSAL, x[s], sse=True)
        assert all(np.array_equal(alt[k][s] if streams > 1 else alt[k], ref[k]) for k in "PHR")


@pytest.mark.parametrize("fs,hop,flags,streams,soft", [
    (44100.0, 2048, o.OUTPUT_PERCUSSIVE, 1, False), (44100.0, 4096, o.OUTPUT_PERCUSSIVE, 1, False),
    (48000.0, 2048, ALL, 1, False), (48000.0, 4096, ALL, 1, False), (44100.0, 4096, o.OUTPUT_HARMONIC | o.OUTPUT_RESIDUAL, 1, True),
    (44100.0, 2048, ALL, 3, False), (44100.0, 4096, o.OUTPUT_PERCUSSIVE | o.OUTPUT_HARMONIC, 2, True),
    (22050.0, 2048, ALL, 1, False), (24000.0, 2048, o.OUTPUT_PERCUSSIVE, 1, False), (32000.0, 2048, ALL, 1, True),
    (88200.0, 4096, ALL, 1, False), (96000.0, 4096, o.OUTPUT_PERCUSSIVE, 2, False), (64000.0, 4096, ALL, 1, False)])
def test_hpr_long_hops_hop_by_hop_single_launch(z, fs, hop, flags, streams, soft):
    """hop 2048 / 4096 through the realtime API: every hop of the causal median path is ONE launch whose frame is
    spread over cooperating workgroups, the transforms cut in two steps (rt_wide.hip).  Same samples as the
    oracle, mixed with block calls (general engine) in both directions, and equal to the four-launch path."""
    n_hops = 14
    x = np.stack([music(hop * n_hops, 60 + s, fs) for s in range(streams)])
    xs = (lambda a, b: x[:, a * hop:b * hop]) if streams > 1 else (lambda a, b: x[0, a * hop:b * hop])

    def engine():
        g = z.HPR(fs, hop, 2.0, flags, z.TIME_CAUSAL, True, streams)
        if soft:
            g.use_soft_mask()
        return g
    g = engine()
    parts = [g.process_stream_host(xs(0, 5), block=1), g.process_stream_host(xs(5, 9), block=4),
             g.process_stream_host(xs(9, n_hops), block=1)]
    refs = [run_oracle(fs, hop, 2.0, flags, o.TIME_CAUSAL, x[s], soft=soft)[1] for s in range(streams)]
    for s in range(streams):
        for k in "PHR":
            got = np.concatenate([(p[k][s] if streams > 1 else p[k]) for p in parts])
            assert np.array_equal(got, refs[s][k]), (k, s)
    z.set_option("no_rt_fused", 1)                    # the four-launch path hop by hop
    try:
        alt = engine().process_stream_host(x if streams > 1 else x[0], block=1)
    finally:
        z.set_option("no_rt_fused", 0)
    for s in range(streams):
        assert all(np.array_equal(alt[k][s] if streams > 1 else alt[k], refs[s][k]) for k in "PHR")


@pytest.mark.parametrize("hop,n_hops", [(2048, 4000), (4096, 2500)])
def test_hpr_long_hops_single_launch_stress(z, hop, n_hops):
    """Thousands of cooperative single-hop launches in a row (five grid barriers each, released at workgroup scope
    when the workgroups share an XCD) against the same stream in one block call of the general engine: a
    visibility race between the workgroups would show up as a differing sample."""
    fs = 44100.0
    x = music(hop * n_hops, 7, fs)
    ref = z.HPR(fs, hop, 2.0, o.OUTPUT_PERCUSSIVE, z.TIME_CAUSAL, True, 1, 256).process_stream_host(x, block=250)
    got = z.HPR(fs, hop, 2.0, o.OUTPUT_PERCUSSIVE, z.TIME_CAUSAL).process_stream_host(x, block=1)
    assert np.array_equal(got["P"], ref["P"])
    assert np.any(got["P"] != 0)


def test_hpr_long_hops_many_streams_take_the_general_engine(z):
    """More than 8 streams: single hops at hop 2048 go through the four-launch path (the cooperative kernel keeps its
    workgroups on one XCD, which is wrong for many streams); same samples either way."""
    fs, hop, n_hops, streams = 44100.0, 2048, 6, 10
    x = np.stack([music(hop * n_hops, 80 + s, fs) for s in range(streams)])
    got = z.HPR(fs, hop, 2.0, o.OUTPUT_PERCUSSIVE, z.TIME_CAUSAL, True, streams).process_stream_host(x, block=1)
    for s in (0, 4, 9):
        _, ref = run_oracle(fs, hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL, x[s])
        assert np.array_equal(got["P"][s], ref["P"]), s


def test_unscaled_double_sqrt_is_sqrt(z, tmp_path):
    """|S| = (float)sqrt((double)re^2 + (double)im^2): the kernels take that square root with the compiler's own
    correctly rounded sequence minus its range scaling (fft_dev.h cabs_exact).  tools/check_sqrt.hip compares the two
    on the device for 3e10 pairs of floats of every kind the path can produce."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "check_sqrt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-I", os.path.join(root, "zen_amd", "csrc"),
                    os.path.join(root, "tools", "check_sqrt.hip"), "-o", exe], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.PIPE, timeout=600)
    r = subprocess.run([exe], stdout=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stdout.decode()
    assert '"mismatches": 0' in r.stdout.decode()


def test_hpr_multi_stream_matches_single(z):
    fs, hop, n_hops, S = 44100.0, 256, 40, 5
    x = np.stack([noise(hop * n_hops, 100 + s) for s in range(S)])
    g = z.HPR(fs, hop, 2.0, ALL, z.TIME_ANTICAUSAL, True, S)
    got = g.process_stream_host(x, block=13)
    for s in range(S):
        _, ref = run_oracle(fs, hop, 2.0, ALL, o.TIME_ANTICAUSAL, x[s])
        assert all(np.array_equal(got[k][s], ref[k]) for k in "PHR")


def test_hpr_rejects_bad_args(z):
    with pytest.raises(z.ZenHipError):
        z.HPR(44100.0, 300, 2.0, ALL, z.TIME_CAUSAL)        # nfft not a power of two
    with pytest.raises(z.ZenHipError):
        z.HPR(44100.0, 8192, 2.0, ALL, z.TIME_CAUSAL)       # nfft 32768 unsupported
    with pytest.raises(z.ZenHipError):
        z.HPR(44100.0, 256, 2.0, ALL, z.FREQUENCY)


# ---------------------------------------------------------------------------- HPRIOffline
@pytest.mark.parametrize("extra", [0, 11])
def test_offline_reference_public_test(z, extra):
    """libzen/hps_gpu_public.test.cu / hps_cpu_public.test.cu:63-101: 20 x 4096 (+11) samples @48 kHz."""
    x = np.concatenate([noise(20 * 4096, 10), np.zeros(extra, np.float32)])
    off = z.HPRIOffline(48000.0, 4096, 256, 2.0, 2.0)
    h, p, r = off.process(x)
    assert p.size == x.size
    assert not np.any(p[:20 * 4096] == x[:20 * 4096])
    rh, rp, rr = o.HPRIOffline(48000.0, 4096, 256, 2.0, 2.0).process(x)
    assert np.array_equal(p, rp) and np.array_equal(h, rh) and np.array_equal(r, rr)
    # a second call on the same object starts from fresh state
    h2, p2, _ = off.process(x)
    assert np.array_equal(p2, rp) and np.array_equal(h2, rh)


def test_offline_config1_shape(z):
    """BASELINE config 1: --hps 4096 2.5 256 2.5 on a 161 571-sample clip (README.md:100), S-music."""
    x = music(161571, 1)
    h, p, r = z.HPRIOffline(44100.0, 4096, 256, 2.5, 2.5).process(x)
    rh, rp, rr = o.HPRIOffline(44100.0, 4096, 256, 2.5, 2.5).process(x)
    assert np.array_equal(p, rp) and np.array_equal(h, rh) and np.all(r == 0)


def test_offline_soft_mask_and_other_hops(z):
    x = music(50000, 2)
    a = z.HPRIOffline(44100.0, 1024, 256, 2.5, 2.5)
    b = o.HPRIOffline(44100.0, 1024, 256, 2.5, 2.5)
    a.use_soft_mask(), b.use_soft_mask()
    h, p, _ = a.process(x)
    rh, rp, _ = b.process(x)
    assert np.array_equal(p, rp) and np.array_equal(h, rh)
    with pytest.raises(z.ZgException):
        z.HPRIOffline(44100.0, 4096, 300)                  # hps.cu:33-36


def test_offline_batch_of_clips(z):
    C, n = 4, 30000
    x = np.stack([music(n, 20 + c) for c in range(C)])
    off = z.HPRIOffline(44100.0, 2048, 256, 2.0, 2.0, False, C)
    din = z.DeviceBuffer.from_host(x)
    dh, dp = z.DeviceBuffer(C * n), z.DeviceBuffer(C * n)
    off.process_device(din.ptr, n, n, dh.ptr, dp.ptr, None, n)
    z.synchronize()
    H, P = dh.download().reshape(C, n), dp.download().reshape(C, n)
    for c in range(C):
        rh, rp, _ = o.HPRIOffline(44100.0, 2048, 256, 2.0, 2.0).process(x[c])
        assert np.array_equal(P[c], rp) and np.array_equal(H[c], rh)


# ---------------------------------------------------------------------------- full-size properties
def test_full_size_median_properties(z):
    """BASELINE path shape 25 840 x 4096 (3 / 47): properties that need no oracle run at this size."""
    rows, cols = 25840, 4096
    rng = np.random.default_rng(0)
    d = rng.uniform(0, 1, (rows, cols)).astype(np.float32)
    src, dst = z.DeviceBuffer.from_host(d), z.DeviceBuffer(d.size)
    mf = z.MedianFilterGPU(rows, cols, 47, z.FREQUENCY)
    mf.filter(src, dst)
    z.synchronize()
    P = dst.download().reshape(rows, cols)
    # (1) every output is one of its window's inputs, (2) exactly `mid` window elements are <= / >= it,
    # checked on a random sample of positions; (3) a sampled set of whole rows matches the oracle
    for r in rng.integers(0, rows, 6):
        assert np.array_equal(P[r:r + 1], o.median_filter(d[r:r + 1], 47, o.FREQUENCY))
    # (4) monotone transform commutes with the median: med(2x+1) == 2 med(x)+1 exactly for these floats
    src.upload(d * 2 + 1)
    mf.filter(src, dst)
    z.synchronize()
    assert np.array_equal(dst.download().reshape(rows, cols), P * 2 + 1)
    # (5) idempotent on constant rows, order-preserving: min <= P <= max per row
    assert np.all(P.min(axis=1) >= d.min(axis=1)) and np.all(P.max(axis=1) <= d.max(axis=1))
    mt = z.MedianFilterGPU(rows, cols, 3, z.TIME_ANTICAUSAL)
    src.upload(d)
    mt.filter(src, dst)
    z.synchronize()
    Hm = dst.download().reshape(rows, cols)
    ref = np.median(np.stack([np.vstack([d[:1], d[:-1]]), d, np.vstack([d[1:], d[-1:]])]), axis=0)
    assert np.array_equal(Hm, ref)


def test_full_size_stream_linearity_of_blocks(z):
    """10 s of config 2 (hop 1024, P only, causal): one block == per-hop calls, checksum of checksums."""
    fs, hop, n_hops = 44100.0, 1024, 430
    x = music(hop * n_hops, 11)
    g1 = z.HPR(fs, hop, 2.0, z.OUTPUT_PERCUSSIVE, z.TIME_CAUSAL)
    a = g1.process_stream_host(x)["P"]
    g2 = z.HPR(fs, hop, 2.0, z.OUTPUT_PERCUSSIVE, z.TIME_CAUSAL, True, 1, 1)
    b = g2.process_stream_host(x, block=1)["P"]
    assert np.array_equal(a, b)
    _, ref = run_oracle(fs, hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL, x[:hop * 40])
    assert np.array_equal(a[:hop * 40], ref["P"])


# ---------------------------------------------------------------------------- edge cases
@pytest.mark.parametrize("n", [0, 1, 7, 255, 256, 257, 4095, 4096, 4097, 10000])
def test_offline_ragged_and_tiny_clips(z, n):
    """Clip lengths around the hop sizes, down to a single sample and the empty clip (hps.cu:109-126 padding
    arithmetic)."""
    x = noise(n, 40 + n)
    h, p, r = z.HPRIOffline(44100.0, 1024, 256, 2.0, 2.0).process(x)
    rh, rp, rr = o.HPRIOffline(44100.0, 1024, 256, 2.0, 2.0).process(x)
    assert h.size == n and np.array_equal(p, rp) and np.array_equal(h, rh) and np.all(r == 0)


def test_offline_silence_and_constant(z):
    for x in (np.zeros(9000, np.float32), np.full(9000, 0.25, np.float32)):
        h, p, _ = z.HPRIOffline(44100.0, 2048, 256, 2.0, 2.0).process(x)
        rh, rp, _ = o.HPRIOffline(44100.0, 2048, 256, 2.0, 2.0).process(x)
        assert np.array_equal(p, rp) and np.array_equal(h, rh)
        assert np.all(np.isfinite(p)) and np.all(np.isfinite(h))


def test_hpr_largest_supported_hop_48k(z):
    """hop 4096 @ 48 kHz: nfft 16384, l_perc 171 -> mask 171, the largest transform."""
    fs, hop = 48000.0, 4096
    x = music(hop * 10, 12, fs)
    for caus in (o.TIME_CAUSAL, o.TIME_ANTICAUSAL):
        ho, ref = run_oracle(fs, hop, 2.0, ALL, caus, x)
        g = z.HPR(fs, hop, 2.0, ALL, caus)
        assert g.freq_len == 171 and g.nfft == 16384
        assert same(g.process_stream_host(x, block=3), ref)


def test_hpr_smallest_hops(z):
    """hop 32 (libzen/hps.bench.cu:62 lower end): nfft 128, time mask 93+ taps, frequency mask 1."""
    for fs, hop in ((48000.0, 32), (44100.0, 16), (44100.0, 8)):
        try:
            ho = o.HPR(fs, hop, 2.0, ALL, o.TIME_ANTICAUSAL)
        except o.OracleError:
            with pytest.raises(z.ZenHipError):
                z.HPR(fs, hop, 2.0, ALL, z.TIME_ANTICAUSAL)
            continue
        n_hops = 3 * ho.stft_width
        x = noise(hop * n_hops, hop)
        for caus in (o.TIME_CAUSAL, o.TIME_ANTICAUSAL):
            _, ref = run_oracle(fs, hop, 2.0, ALL, caus, x)
            try:
                g = z.HPR(fs, hop, 2.0, ALL, caus)
            except z.ZenHipError as e:        # mask > 255 taps is outside what the kernels cover
                assert e.code == 5
                continue
            assert same(g.process_stream_host(x, block=50), ref)


def test_hpr_multi_stream_chunked_soft_sse(z):
    fs, hop, n_hops, S = 44100.0, 512, 30, 3
    x = np.stack([music(hop * n_hops, 200 + s) for s in range(S)])
    for mode in ("soft", "sse"):
        g = z.HPR(fs, hop, 2.0, ALL, z.TIME_ANTICAUSAL, True, S, 4)      # 4-hop chunks
        getattr(g, "use_soft_mask" if mode == "soft" else "use_sse_filter")()
        got = g.process_stream_host(x, block=11)
        for s in range(S):
            _, ref = run_oracle(fs, hop, 2.0, ALL, o.TIME_ANTICAUSAL, x[s], sse=(mode == "sse"), soft=(mode == "soft"))
            assert all(np.array_equal(got[k][s], ref[k]) for k in "PHR"), (mode, s)


def test_copy_output_before_any_hop_is_zero(z):
    g = z.HPR(44100.0, 256, 2.0, ALL, z.TIME_CAUSAL)
    d = z.DeviceBuffer(256)
    d.upload(np.ones(256, np.float32))
    g.copy_output(z.OUTPUT_PERCUSSIVE, d.ptr)
    assert np.all(d.download() == 0)


def test_median_negative_zero_and_infinities(zk):
    d = np.array([[0.0, -0.0, np.inf, 1.0, -np.inf, 2.0, -1.0, 3.0, 0.5, -0.5, 7.0, 8.0]], np.float32)
    d = np.repeat(d, 5, axis=0)
    for flen in (3, 5, 7, 9):
        got = zk.MedianFilterGPU(5, 12, flen, zk.FREQUENCY).filter_host(d)
        assert np.array_equal(got, o.median_filter(d, flen, o.FREQUENCY))


@pytest.mark.parametrize("fs,hop", [(44100.0, 128), (48000.0, 256), (44100.0, 256), (48000.0, 512), (44100.0, 512),
                                    (48000.0, 1024), (44100.0, 1024)])
@pytest.mark.parametrize("soft", [False, True])
def test_realtime_fused_single_launch_hop(z, fs, hop, soft):
    """Per-hop calls (one fused launch, rt_fused.hip) == the three-kernel path == the oracle, all outputs."""
    n_hops = 40
    x = music(hop * n_hops, 300 + hop, fs)
    _, ref = run_oracle(fs, hop, 2.0, ALL, o.TIME_CAUSAL, x, soft=soft)
    outs = []
    for no_fused in (0, 1):
        z.set_option("no_rt_fused", no_fused)
        try:
            g = z.HPR(fs, hop, 2.0, ALL, z.TIME_CAUSAL, True, 1, 8)
            if soft:
                g.use_soft_mask()
            outs.append(g.process_stream_host(x, block=1))
        finally:
            z.set_option("no_rt_fused", 0)
    assert same(outs[0], ref) and same(outs[1], ref)
    # mixing single hops and blocks on one engine
    g = z.HPR(fs, hop, 2.0, ALL, z.TIME_CAUSAL, True, 1, 8)
    if soft:
        g.use_soft_mask()
    din = z.DeviceBuffer.from_host(x)
    dout = {k: z.DeviceBuffer(x.size) for k in "PHR"}
    off = 0
    for m in (1, 1, 5, 1, 8, 3, 1):
        g.process(din.offset(off * hop), m, m * hop, dout["H"].offset(off * hop), dout["P"].offset(off * hop),
                  dout["R"].offset(off * hop), m * hop)
        off += m
    z.synchronize()
    for k in "PHR":
        assert np.array_equal(dout[k].download()[:off * hop], ref[k][:off * hop])


@pytest.mark.parametrize("fs,hop,n_hops", [(44100.0, 256, 60), (44100.0, 1024, 24), (44100.0, 4096, 8), (48000.0, 64, 130)])
@pytest.mark.parametrize("flags", [7, 3, 6, 5])
def test_hard_mask_outputs_in_one_workgroup(z, fs, hop, n_hops, flags):
    """Hard masks, several outputs (H|P|R subsets, incl. the residual without one of its masks): the synthesis
    kernel that keeps the two binary masks of a frame in a register and loops over the outputs equals the
    one-workgroup-per-output kernel and the oracle."""
    x = music(hop * n_hops, seed=hop + flags, fs=fs)
    _, ref = run_oracle(fs, hop, 2.0, flags, o.TIME_ANTICAUSAL, x)
    for off in (0, 1):
        z.set_option("no_istft_multi", off)
        try:
            got = z.HPR(fs, hop, 2.0, flags, z.TIME_ANTICAUSAL).process_stream_host(x)
        finally:
            z.set_option("no_istft_multi", 0)
        assert same(got, ref), off


@pytest.mark.parametrize("scale", [1e-10, 1e-20, 1e-30, 1e-38, 1e15, 1e30])
@pytest.mark.parametrize("soft", [False, True])
def test_hpr_extreme_amplitudes(z, scale, soft):
    """Denormal products in the soft mask, |S|^2 beyond float range (the hypot is evaluated in double),
    ratios against Eps: same bits as the CPU path on the fused kernel (causal) and on the general engine
    (anticausal).  The device code must neither flush denormals nor contract multiply-adds."""
    hop, n_hops = 256, 40
    x = (np.random.default_rng(5).uniform(-1, 1, hop * n_hops) * scale).astype(np.float32)
    for caus in (o.TIME_CAUSAL, o.TIME_ANTICAUSAL):
        _, ref = run_oracle(44100.0, hop, 2.0, ALL, caus, x, soft=soft)
        g = z.HPR(44100.0, hop, 2.0, ALL, caus)
        if soft:
            g.use_soft_mask()
        got = g.process_stream_host(x)
        for k in "PHR":
            assert np.array_equal(got[k], ref[k], equal_nan=True), (caus, k)


@pytest.mark.parametrize("fs,hop", [(44100.0, 128), (48000.0, 256), (44100.0, 512), (48000.0, 1024), (44100.0, 1024)])
@pytest.mark.parametrize("minb", [1, 2, 3])
def test_block_fused_matches_three_kernel_path(z, fs, hop, minb):
    """Blocks of causal hops run in the fused one-workgroup-per-hop kernel (rt_fused.hip); same samples as
    the oracle and as the STFT / median / iSTFT kernels, for several streams, with engine chunks shorter
    than the call, and when the two paths alternate on one engine (the fused block call leaves the
    spectrum rings untouched, which a causal stream never reads back)."""
    n_hops, S = 37, 3
    x = np.stack([music(hop * n_hops, 500 + hop + s, fs) for s in range(S)])
    refs = [run_oracle(fs, hop, 2.0, ALL, o.TIME_CAUSAL, x[s])[1] for s in range(S)]
    ref = {k: np.stack([r[k] for r in refs]) for k in "PHR"}
    z.set_option("block_fused_minb", minb)
    try:
        for chunk in (0, 8):
            g = z.HPR(fs, hop, 2.0, ALL, z.TIME_CAUSAL, True, S, chunk)
            assert same(g.process_stream_host(x), ref), chunk
            assert same(z.HPR(fs, hop, 2.0, ALL, z.TIME_CAUSAL, True, S, chunk).process_stream_host(x, block=5), ref)
        g = z.HPR(fs, hop, 2.0, ALL, z.TIME_CAUSAL, True, S, 16)
        din = z.DeviceBuffer.from_host(x)
        dout = {k: z.DeviceBuffer(x.size) for k in "PHR"}
        off, n = 0, x.shape[1]
        for i, m in enumerate((6, 1, 9, 4, 1, 16)):
            z.set_option("no_block_fused", i & 1)
            g.process(din.offset(off * hop), m, n, dout["H"].offset(off * hop), dout["P"].offset(off * hop),
                      dout["R"].offset(off * hop), n)
            off += m
        z.synchronize()
        for k in "PHR":
            assert np.array_equal(dout[k].download().reshape(S, n)[:, :off * hop], ref[k][:, :off * hop]), k
    finally:
        z.set_option("no_block_fused", 0)
        z.set_option("block_fused_minb", 3)


@pytest.mark.parametrize("hop_h,hop_p,n", [(4096, 256, 400000), (1024, 256, 161571), (2048, 512, 90000)])
def test_offline_time_sharded_equals_whole_clip(z, hop_h, hop_p, n):
    """one clip cut into time ranges, every range computed independently (as different GPUs
    would) with a warm-up halo, must reproduce the single-pass result bit for bit."""
    from zen_amd import dist as zdist
    x = music(n, 77)
    whole = z.HPRIOffline(44100.0, hop_h, hop_p, 2.0, 2.0)
    h_ref, p_ref, _ = whole.process(x)
    d_in = z.DeviceBuffer.from_host(x)
    for world in (2, 3, 7):
        eng = z.HPRIOffline(44100.0, hop_h, hop_p, 2.0, 2.0)
        for rank, (b, e) in enumerate(zdist.time_shards(n, world, hop_h)):
            if b == e:
                continue
            ib, ie = eng.range_halo(n, b, e)
            assert 0 <= ib <= b and e <= max(ie, e) and ie <= n
            dh, dp = z.DeviceBuffer(e - b), z.DeviceBuffer(e - b)
            eng.process_range(d_in.ptr, n, b, e, dh.ptr, dp.ptr)
            z.synchronize()
            assert np.array_equal(dh.download(), h_ref[b:e]), (world, rank, "harm")
            assert np.array_equal(dp.download(), p_ref[b:e]), (world, rank, "perc")
    # the halo is local: a middle shard of a long clip reads far less than the clip
    ib, ie = whole.range_halo(n, n // 2, n // 2 + hop_h)
    assert ie - ib < n // 2


def test_full_size_offline_batch_config4_properties(z):
    """BASELINE configs[3] per-GPU shape: 64 x 30 s clips, HPR-I 4096/256.  Too big for the oracle, so:
    (1) batching is invisible -- clips 0, 31, 63 of the batch equal single-clip runs; (2) the first second
    of clip 0 equals the oracle on the prefix (outputs at sample n depend on input up to n + halo only);
    (3) identical clips give identical outputs."""
    C, n = 64, 1323000
    rng = np.random.default_rng(5)
    base = rng.uniform(-1, 1, (4, n)).astype(np.float32)
    x = base[np.arange(C) % 4]                       # 64 clips, 4 distinct
    off = z.HPRIOffline(44100.0, 4096, 256, 2.0, 2.0, False, C)
    din = z.DeviceBuffer.from_host(x)
    dh, dp = z.DeviceBuffer(C * n), z.DeviceBuffer(C * n)
    off.process_device(din.ptr, n, n, dh.ptr, dp.ptr, None, n)
    z.synchronize()
    H, P = dh.download().reshape(C, n), dp.download().reshape(C, n)
    single = z.HPRIOffline(44100.0, 4096, 256, 2.0, 2.0)
    for c in (0, 31, 63):
        h1, p1, _ = single.process(x[c])
        assert np.array_equal(H[c], h1) and np.array_equal(P[c], p1)
    assert np.array_equal(P[1], P[5]) and np.array_equal(H[2], H[62])
    m = 44100
    halo = 4 * 4096 + 30 * 256
    rh, rp, _ = o.HPRIOffline(44100.0, 4096, 256, 2.0, 2.0).process(x[0][:m + halo])
    assert np.array_equal(P[0][:m], rp[:m]) and np.array_equal(H[0][:m], rh[:m])


def test_random_configurations_differential(z):
    """A short run of tools/fuzz_parity.py: random sample rates, hops, output flags, causality, mask types,
    stream counts, blockings and chunk sizes against the oracle (a 150 s run covers ~340 distinct
    configurations with no mismatch; this one a few dozen)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(root, "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    n_ok, n_bad, _, n_refused, n_seen = fuzz.run(12.0, 7)
    assert n_bad == 0 and n_refused == 0 and n_ok >= 10 and n_seen >= 10   # (refused: the oracle accepted, the engine did not)
