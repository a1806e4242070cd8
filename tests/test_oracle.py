"""Oracle self-consistency + the reference's HPR property tests (libzen/hps.test.cu:160-372,
libzen/hps_cpu_public.test.cu:63-101) run against the CPU restatement."""
import os

import numpy as np
import pytest
from scipy.ndimage import median_filter as sp_median
from scipy.ndimage import uniform_filter1d

from oracle import oracle as o

ALL = o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE | o.OUTPUT_RESIDUAL


def noise(n, seed=0):
    # hps.test.cu:24-36 : uniform(-1, 1) "realistic normalized floats"
    return np.random.default_rng(seed).uniform(-1, 1, n).astype(np.float32)


# ---------------------------------------------------------------- filters vs scipy / brute force
@pytest.mark.parametrize("shape,flen", [((7, 33), 3), ((22, 1024), 13), ((22, 1024), 11), ((6, 4096), 47),
                                        ((2, 1024), 187), ((40, 50), 21), ((5, 5), 5), ((1, 9), 1)])
def test_median_matches_scipy_and_bruteforce(shape, flen):
    rng = np.random.default_rng(flen)
    d = rng.uniform(0, 10, shape).astype(np.float32)
    d[rng.integers(0, shape[0], 5), rng.integers(0, shape[1], 5)] = 0   # ties
    if flen <= shape[1]:
        got = o.median_filter(d, flen, o.FREQUENCY)
        assert np.array_equal(got, sp_median(d, size=(1, flen), mode="nearest"))
        assert np.array_equal(got, o.median_filter_bruteforce(d, flen, o.FREQUENCY))
    if flen <= shape[0]:
        for direction in (o.TIME_CAUSAL, o.TIME_ANTICAUSAL):    # identical on CPU, mfilt.h:311-314
            got = o.median_filter(d, flen, direction)
            assert np.array_equal(got, sp_median(d, size=(flen, 1), mode="nearest"))
            assert np.array_equal(got, o.median_filter_bruteforce(d, flen, direction))


def test_box_matches_float64_mean():
    rng = np.random.default_rng(1)
    d = rng.uniform(0, 10, (12, 2048)).astype(np.float32)
    got = o.box_filter(d, 23, o.FREQUENCY)
    ref = uniform_filter1d(d.astype(np.float64), 23, axis=1, mode="nearest")
    assert np.allclose(got, ref, rtol=2e-6)
    got = o.box_filter(d, 7, o.TIME_ANTICAUSAL)
    ref = uniform_filter1d(d.astype(np.float64), 7, axis=0, mode="nearest")
    assert np.allclose(got, ref, rtol=2e-6)


def test_cabs_is_libm_hypotf():
    rng = np.random.default_rng(5)
    z = (rng.normal(size=200000) * 10 ** rng.uniform(-6, 4, 200000)
         + 1j * rng.normal(size=200000) * 10 ** rng.uniform(-6, 4, 200000)).astype(np.complex64)
    assert np.array_equal(o.cabs(z), np.hypot(z.real, z.imag))   # numpy float32 hypot -> libm hypotf


def test_twiddle_table_symmetry():
    for n in (8, 64, 1024, 16384):
        tw = o.twiddles(n)
        assert tw.size == n // 2
        q = n // 4
        assert tw[0] == 1 and tw[q] == -1j
        assert np.array_equal(tw[q:].real, tw[:q].imag) and np.array_equal(tw[q:].imag, -tw[:q].real)
        assert np.abs(tw - np.exp(-2j * np.pi * np.arange(n // 2) / n)).max() < 6e-8


# ---------------------------------------------------------------- parameter derivation
@pytest.mark.parametrize("fs,hop,exp", [
    (44100, 256, (512, 1024, 11, 22, 12)), (44100, 512, (1024, 2048, 6, 12, 23)),
    (44100, 1024, (2048, 4096, 3, 6, 46)), (44100, 2048, (4096, 8192, 1, 2, 93)),
    (44100, 4096, (8192, 16384, 1, 2, 186)), (48000, 256, (512, 1024, 12, 24, 11)),
    (48000, 4096, (8192, 16384, 1, 2, 171))])
def test_derived_sizes(fs, hop, exp):
    h = o.HPR(float(fs), hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_ANTICAUSAL)
    assert (h.nwin, h.nfft, h.l_harm, h.stft_width, h.l_perc) == exp
    assert h.lag == h.l_harm
    assert o.HPR(float(fs), hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL).lag == 1   # hps.h:265-268
    assert abs(h.cola_factor - 4.0) < 1e-4                                        # nfft / sum(w^2)


def test_window_is_periodic_sqrt_hann():
    w = o.window_sqrt_hann(512)
    ref = np.sqrt(0.5 * (1 - np.cos(2 * np.pi * np.arange(512) / 512)))
    assert np.abs(w - ref).max() < 3e-4 and w[0] < 3e-4   # float PI / cosf; w[0] is sqrt of a tiny residue
    assert abs((w[:256] ** 2 + w[256:] ** 2) - 1).max() < 1e-6   # COLA at hop = nwin/2


# ---------------------------------------------------------------- hps.test.cu properties
@pytest.fixture(scope="module")
def variants():
    hop, n_hops = 256, 100                     # hps.test.cu:136-157
    data = noise(hop * n_hops)
    out = {}
    for caus in (o.TIME_CAUSAL, o.TIME_ANTICAUSAL):
        for cb in (True, False):
            h = o.HPR(48000.0, hop, 2.0, ALL, caus, cb)
            out[(caus, cb)] = h.process_stream(data)
    return data, out


def test_processing_modifies_input(variants):
    data, out = variants                        # hps.test.cu:160-226
    for v in out.values():
        for k in "PHR":
            assert not np.any(v[k] == data)


def test_copybord_is_ignored_on_cpu_and_causal_differs(variants):
    _, out = variants                           # hps.test.cu:228-283
    for caus in (o.TIME_CAUSAL, o.TIME_ANTICAUSAL):
        for k in "PHR":
            assert np.array_equal(out[(caus, True)][k], out[(caus, False)][k])
    c, a = out[(o.TIME_CAUSAL, True)]["P"], out[(o.TIME_ANTICAUSAL, True)]["P"]
    assert not np.any(c == a)


def test_perc_only_leaves_h_and_r_zero():
    hop = 256                                   # hps.test.cu:285-343
    data = noise(hop * 40, 1)
    for caus in (o.TIME_CAUSAL, o.TIME_ANTICAUSAL):
        h = o.HPR(48000.0, hop, 2.0, o.OUTPUT_PERCUSSIVE, caus)
        res = h.process_stream(data)
        assert np.all(res["H"] == 0) and np.all(res["R"] == 0) and np.any(res["P"] != 0)


def test_reset_gives_identical_rerun():
    hop = 256                                   # hps.test.cu:345-372
    data = noise(hop * 30, 2)
    h = o.HPR(48000.0, hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL)
    a = h.process_stream(data)["P"]
    h.reset_buffers()
    b = h.process_stream(data)["P"]
    assert np.array_equal(a, b)
    h.warmup()                                  # hps.cu:410-427 ends in reset_buffers
    assert np.array_equal(h.process_stream(data)["P"], a)


def test_q1_causal_time_median_is_identity_at_consumed_row():
    hop = 256
    h = o.HPR(48000.0, hop, 2.0, ALL, o.TIME_CAUSAL)
    for i, x in enumerate(noise(hop * 30, 3).reshape(-1, hop)):
        h.process_next_hop(x)
        assert np.array_equal(h.matrix("harmonic_matrix")[-1], h.matrix("s_mag")[-1])


def test_q4_sliding_rows_equal_whole_clip_filter():
    hop, n = 256, 80
    h = o.HPR(48000.0, hop, 2.0, ALL, o.TIME_ANTICAUSAL)
    mags, hs, ps = [], [], []
    r = h.stft_width - h.lag
    for x in noise(hop * n, 4).reshape(-1, hop):
        h.process_next_hop(x)
        mags.append(h.matrix("s_mag")[-1])
        hs.append(h.matrix("harmonic_matrix")[r])
        ps.append(h.matrix("percussive_matrix")[r])
    mags = np.array(mags)
    W = h.stft_width
    whole = np.vstack([np.zeros((W, h.nfft), np.float32), mags])       # zero history before the stream
    Hw = o.median_filter(whole, h.l_harm, o.TIME_ANTICAUSAL)[W:]
    Pw = o.median_filter(whole, h.l_perc, o.FREQUENCY)[W:]
    lag = h.lag
    # row r at hop i holds frame i - (lag - 1)
    assert np.array_equal(np.array(hs)[lag - 1:], Hw[:n - lag + 1])
    assert np.array_equal(np.array(ps)[lag - 1:], Pw[:n - lag + 1])


def test_q7_percussive_matrix_not_mirror_symmetric_at_edges():
    hop = 256
    h = o.HPR(44100.0, hop, 2.0, ALL, o.TIME_CAUSAL)
    for x in noise(hop * 30, 5).reshape(-1, hop):
        h.process_next_hop(x)
    P = h.matrix("percussive_matrix")          # all rows (the CPU filters the whole matrix)
    N, mid = h.nfft, (h.l_perc + 1 - h.l_perc % 2) // 2
    k = np.arange(1, N // 2)
    asym = k[(P[:, k] != P[:, N - k]).any(axis=0)]
    assert asym.size > 0 and asym.max() <= mid


def test_hard_masks_are_binary_and_residual_is_complement():
    hop = 512
    h = o.HPR(44100.0, hop, 2.0, ALL, o.TIME_ANTICAUSAL)
    for x in noise(hop * 20, 6).reshape(-1, hop):
        h.process_next_hop(x)
    r = h.stft_width - h.lag
    pm, hm, rm = (h.matrix(n)[r] for n in ("percussive_mask", "harmonic_mask", "residual_mask"))
    assert set(np.unique(pm)) <= {0.0, 1.0} and set(np.unique(hm)) <= {0.0, 1.0}
    assert np.array_equal(rm, 1 - (hm + pm))


def test_soft_mask_truncates_beta_and_skips_residual():
    hop = 256
    data = noise(hop * 30, 7)
    a = o.HPR(44100.0, hop, 2.5, ALL, o.TIME_ANTICAUSAL)
    b = o.HPR(44100.0, hop, 2.0, ALL, o.TIME_ANTICAUSAL)
    a.use_soft_mask(), b.use_soft_mask()
    ra, rb = a.process_stream(data), b.process_stream(data)
    assert np.array_equal(ra["P"], rb["P"]) and np.array_equal(ra["H"], rb["H"])   # (int)2.5 == 2
    assert np.all(ra["R"] == 0)                                                    # hps.cu:562


def test_sse_path_runs_and_has_no_residual():
    hop = 512
    h = o.HPR(44100.0, hop, 2.0, ALL, o.TIME_CAUSAL, False)
    h.use_sse_filter()
    res = h.process_stream(noise(hop * 30, 8))
    assert np.all(np.isfinite(res["P"])) and np.any(res["P"] != 0) and np.any(res["H"] != 0)
    assert np.all(res["R"] == 0)


def test_output_scale_q6():
    hop = 256
    x = noise(hop * 200, 9)
    h = o.HPR(44100.0, hop, 2.0, ALL, o.TIME_CAUSAL)
    res = h.process_stream(x)
    total = (res["P"] + res["H"] + res["R"])[hop * 4:]
    ratio = np.sqrt(np.mean(total.astype(np.float64) ** 2)) / np.sqrt(np.mean(x.astype(np.float64) ** 2))
    assert 0.8 * 4 * h.nfft < ratio < 1.5 * 4 * h.nfft     # no 1/nfft, COLA ~ 4, |sin|+|cos| in [1, sqrt 2]


# ---------------------------------------------------------------- HPRIOffline
def test_chunk_padder():
    assert o.chunk_padder(161571, 4096, 1) == (41, 41 * 4096)     # hps.cu:109-126
    assert o.chunk_padder(161571, 256, 11) == (643, 643 * 256)
    assert o.chunk_padder(20 * 4096, 4096, 1) == (21, 21 * 4096)


def test_offline_hops_must_divide():
    with pytest.raises(o.OracleError) as e:       # hps.cu:33-36
        o.HPRIOffline(48000.0, 4096, 300)
    assert e.value.code == o.E_HOPS_NOT_DIVISIBLE


@pytest.mark.parametrize("extra", [0, 11])
def test_offline_basic_and_with_padding(extra):
    # hps_cpu_public.test.cu:63-101 : 20 x 4096 samples (+11), fs 48000, hops 4096/256, beta 2/2
    data = np.concatenate([noise(20 * 4096, 10), np.zeros(extra, np.float32)])
    off = o.HPRIOffline(48000.0, 4096, 256, 2.0, 2.0)
    h, p, r = off.process(data)
    assert p.size == data.size and h.size == data.size
    assert not np.any(p[:20 * 4096] == data[:20 * 4096])
    assert np.all(r == 0)                                            # Q8
    pc = off.process_cpu(data)
    assert np.array_equal(pc[0], p) and np.array_equal(pc[1], p) and np.array_equal(pc[2], p)
    # running twice on the same object state-resets (a fresh object gives the same answer)
    h2, p2, _ = o.HPRIOffline(48000.0, 4096, 256, 2.0, 2.0).process(data)
    assert np.array_equal(p, p2) and np.array_equal(h, h2)


def test_offline_is_two_streaming_passes():
    """HPRIOffline == pass 1 (H+P+R, anticausal) -> P+R shifted by lag*hop -> pass 2 (P only)."""
    fs, hh, hp, n = 44100.0, 1024, 256, 30000
    x = noise(n, 11)
    H, P, _ = o.HPRIOffline(fs, hh, hp, 2.0, 2.0).process(x)
    h1 = o.HPR(fs, hh, 2.0, ALL, o.TIME_ANTICAUSAL)
    n1, pad1 = o.chunk_padder(n, hh, h1.lag)
    xp = np.zeros(pad1, np.float32)
    xp[:n] = x
    r1 = h1.process_stream(xp)
    inter = r1["P"] + r1["R"]
    sh = h1.lag * hh
    inter[:pad1 - sh] = inter[sh:].copy()         # tail keeps stale values (Q9)
    harm = r1["H"][sh:]
    assert np.array_equal(H, harm[:n])
    h2 = o.HPR(fs, hp, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_ANTICAUSAL)
    n2, pad2 = o.chunk_padder(n, hp, h2.lag)
    assert pad2 <= pad1
    r2 = h2.process_stream(inter[:pad2])
    assert np.array_equal(P, r2["P"][h2.lag * hp:][:n])


# ---------------------------------------------------------------- independent float64 model of the path
def _model_hpr_causal(x, fs, hop, beta, soft):
    """The causal HPR path written from its definition in float64 with numpy/scipy only (no oracle code):
    sqrt-Hann analysis window over [previous hop | hop] (hps.h:260-274), zero-padded unnormalised C2C FFT
    (fftw.h:35-43), H = |S| (causal time median = identity), P = frequency median with
    replicate border (mfilt.h:316), masks (hps.h:100-129), unnormalised inverse FFT, * COLA, overlap-add of
    the first nwin samples without a synthesis window (hps.cu:435-449, :526-528).  Returns (P, H) streams."""
    nwin, nfft = 2 * hop, 4 * hop
    l_perc = int(np.floor(500.0 / (fs / nfft) + 0.5))
    wf = l_perc if l_perc % 2 else l_perc + 1
    n = np.arange(nwin)
    w = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * n / nwin))
    cola = nfft / np.sum(w * w)
    eps = np.finfo(np.float32).eps
    n_hops = x.size // hop
    outs = {"P": np.zeros(x.size), "H": np.zeros(x.size)}
    carry = {"P": np.zeros(hop), "H": np.zeros(hop)}
    prev = np.zeros(hop)
    for t in range(n_hops):
        cur = x[t * hop:(t + 1) * hop].astype(np.float64)
        frame = np.concatenate([prev, cur]) * w
        prev = cur
        S = np.fft.fft(frame, nfft)
        H = np.abs(S)
        P = sp_median(H, size=wf, mode="nearest")
        if soft:
            p = int(beta)
            mp, mh = P ** p / (P ** p + H ** p + eps), H ** p / (H ** p + P ** p + eps)
        else:
            mp, mh = (P / (H + eps) >= beta).astype(float), (H / (P + eps) >= beta - eps).astype(float)
        for k, m in (("P", mp), ("H", mh)):
            y = np.real(np.fft.ifft(S * m))[:nwin] * nfft * cola
            outs[k][t * hop:(t + 1) * hop] = carry[k] + y[:hop]
            carry[k] = y[hop:]
    return outs["P"], outs["H"]


@pytest.mark.parametrize("hop", [256, 1024])
def test_oracle_agrees_with_independent_float64_model(hop):
    """The waveforms have no golden vectors in the reference, so the oracle is also
    checked against a model of the path that shares no code with it.  Soft masks are continuous: agreement
    to 2e-6 of the signal's RMS (float32 round-off).  Hard masks flip where the float32 ratio lands on the
    other side of beta: measured 1e-6 .. 4e-5 of the RMS, bounded here by 2e-4 with > 95 % of the samples
    within 1e-4."""
    fs, n_hops = 44100.0, 60
    t = np.arange(hop * n_hops) / fs
    x = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1320 * t)
         + 0.3 * np.random.default_rng(2).uniform(-1, 1, t.size) * (np.arange(t.size) % 4096 < 200)).astype(np.float32)
    for soft in (True, False):
        h = o.HPR(fs, hop, 2.0, o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL)
        if soft:
            h.use_soft_mask()
        got = h.process_stream(x)
        mp, mh = _model_hpr_causal(x, fs, hop, 2.0, soft)
        for g, m in ((got["P"], mp), (got["H"], mh)):
            scale = np.sqrt(np.mean(m ** 2)) + 1e-30
            err = (g.astype(np.float64) - m) / scale
            if soft:
                assert np.sqrt(np.mean(err ** 2)) < 2e-6
            else:
                assert np.sqrt(np.mean(err ** 2)) < 2e-4 and np.mean(np.abs(err) < 1e-4) > 0.95


def test_hard_mask_flip_count_against_another_fft():
    """a hard mask thresholds a ratio of medians of magnitudes, so ANY transform other than the restatement's
    radix-2 DAG (IPP's, numpy's) moves |S| by an ulp now and then and flips bins.  Counted here: the oracle's percussive and
    harmonic masks (float32 DAG, causal hop 1024 -- BASELINE configs[1]) against the same masks from a float64 FFT rounded
    to float32, bin by bin, on a music-like stream.  The count is what INTEGRATION.md section 5a quotes: a few flips per
    10^6 bins, each flip a whole bin of one frame -- which is why the waveforms of two valid implementations agree to
    1e-6..4e-5 of the RMS with hard masks and not to round-off, and why "bit-exact" is promised against the restatement
    only.  The bound asserted is loose (< 200 per 10^6); the measured figure is printed."""
    fs, hop, n_hops, beta = 44100.0, 1024, 2500, 2.0          # 10^7 bins per mask
    nwin, nfft = 2 * hop, 4 * hop
    t = np.arange(hop * n_hops) / fs
    rng = np.random.default_rng(11)
    x = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1320 * t) + 0.05 * rng.uniform(-1, 1, t.size)
         + 0.4 * rng.uniform(-1, 1, t.size) * (np.arange(t.size) % 9000 < 300)).astype(np.float32)
    l_perc = int(np.floor(500.0 / (fs / nfft) + 0.5))
    wf = l_perc if l_perc % 2 else l_perc + 1
    w = o.window_sqrt_hann(nwin).astype(np.float64)
    eps = np.float32(np.finfo(np.float32).eps)
    h = o.HPR(fs, hop, beta, o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL)
    prev = np.zeros(hop, np.float32)
    flips_p = flips_h = bins = 0
    for i in range(n_hops):
        cur = x[i * hop:(i + 1) * hop]
        h.process_next_hop(cur)
        row = h.stft_width - h.lag
        mp_o = h.matrix("percussive_mask")[row]
        mh_o = h.matrix("harmonic_mask")[row]
        frame = (np.concatenate([prev, cur]).astype(np.float32) * w.astype(np.float32)).astype(np.float64)   # the same float32 frame
        prev = cur
        mag = np.abs(np.fft.fft(frame, nfft)).astype(np.float32)          # another FFT, rounded to float32 like any float library's
        P = sp_median(mag, size=wf, mode="nearest").astype(np.float32)
        H = mag                                                            # causal time median = identity
        mp = ((P / (H + eps)).astype(np.float32) >= np.float32(beta)).astype(np.float32)
        mh = ((H / (P + eps)).astype(np.float32) >= np.float32(beta) - eps).astype(np.float32)
        flips_p += int(np.sum(mp != mp_o))
        flips_h += int(np.sum(mh != mh_o))
        bins += nfft
    per_million = 1e6 * (flips_p + flips_h) / (2 * bins)
    print("hard-mask flips vs a float64 FFT: percussive %d, harmonic %d of %d bins each = %.1f per 10^6 bins"
          % (flips_p, flips_h, bins, per_million))
    assert per_million < 200.0
    assert mp_o.shape == (nfft,)


def _model_hpr_stream(x, fs, hop, beta, soft, causal, want=("P", "H", "R")):
    """HPR<B>::process_next_hop (hps.cu:429-580) hop by hop in float64, numpy/scipy only, no oracle code, for
    either causality: sliding matrix of W = 2*l_harm frames (new frame appended as the last row), time median
    (centred, replicate border over the W rows) and frequency median of the whole matrix, masks and synthesis of
    row W - lag (lag = 1 causal, l_harm anticausal).  Returns the streams copy_* would hand out."""
    nwin, nfft = 2 * hop, 4 * hop
    l_harm = int(np.floor(0.2 / ((nfft - hop) / fs) + 0.5))
    l_perc = int(np.floor(500.0 / (fs / nfft) + 0.5))
    W, lag = 2 * l_harm, (1 if causal else l_harm)
    wt = l_harm if l_harm % 2 else l_harm + 1
    wf = l_perc if l_perc % 2 else l_perc + 1
    n = np.arange(nwin)
    w = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * n / nwin))
    cola = nfft / np.sum(w * w)
    eps = np.finfo(np.float32).eps
    n_hops = x.size // hop
    S = np.zeros((W, nfft), np.complex128)
    outs = {k: np.zeros(n_hops * hop) for k in "PHR"}
    carry = {k: np.zeros(hop) for k in "PHR"}
    prev = np.zeros(hop)
    r = W - lag
    for t in range(n_hops):
        cur = x[t * hop:(t + 1) * hop].astype(np.float64)
        S = np.vstack([S[1:], np.fft.fft(np.concatenate([prev, cur]) * w, nfft)[None, :]])
        prev = cur
        mag = np.abs(S)
        H = sp_median(mag, size=(wt, 1), mode="nearest")[r]
        P = sp_median(mag[r], size=wf, mode="nearest")
        if soft:
            p = int(beta)
            mp, mh = P ** p / (P ** p + H ** p + eps), H ** p / (H ** p + P ** p + eps)
            masks = {"P": mp, "H": mh}
        else:
            mp, mh = (P / (H + eps) >= beta).astype(float), (H / (P + eps) >= beta - eps).astype(float)
            masks = {"P": mp, "H": mh, "R": 1.0 - (mh + mp)}
        for k in want:
            y = np.zeros(nwin) if k not in masks else np.real(np.fft.ifft(S[r] * masks[k]))[:nwin] * nfft * cola
            outs[k][t * hop:(t + 1) * hop] = carry[k] + y[:hop]
            carry[k] = y[hop:]
    return outs, l_harm


def _model_hpri_offline(x, fs, hop_h, hop_p, beta, soft):
    """HPRIOffline<GPU>::process (hps.cu:128-221) on top of _model_hpr_stream: pad to ceil(N/hop)+lag hops, pass
    1 with all outputs, intermediate = P + R shifted left by lag_h*hop_h IN PLACE (the tail keeps its old
    contents, hps.cu:171-176), pass 2 (percussive only) reads it up to its own padded length, both
    results shifted by their lag.  Returns (harm, perc)."""
    N = x.size

    def padded(hop, lag):
        return (int(np.ceil(np.float32(N) / np.float32(hop))) + lag) * hop

    l_h = int(np.floor(0.2 / ((4 * hop_h - hop_h) / fs) + 0.5))
    l_p = int(np.floor(0.2 / ((4 * hop_p - hop_p) / fs) + 0.5))
    n1, n2 = padded(hop_h, l_h), padded(hop_p, l_p)
    o1, _ = _model_hpr_stream(np.concatenate([x, np.zeros(n1 - N)]), fs, hop_h, beta, soft, False)
    inter = o1["P"] + o1["R"]
    harm = o1["H"].copy()
    sh1 = l_h * hop_h
    inter[:n1 - sh1] = inter[sh1:].copy()
    harm[:n1 - sh1] = harm[sh1:].copy()
    assert n2 <= n1
    o2, _ = _model_hpr_stream(inter[:n2], fs, hop_p, beta, soft, False, want=("P",))
    perc = o2["P"]
    sh2 = l_p * hop_p
    perc[:n2 - sh2] = perc[sh2:].copy()
    return harm[:N], perc[:N]


@pytest.mark.parametrize("hop,soft", [(256, True), (256, False), (512, True)])
def test_oracle_anticausal_stream_agrees_with_float64_model(hop, soft):
    """The anticausal streaming path (what the offline passes run): time median over the sliding matrix,
    consumed row W - l_harm, all three outputs.  Same tolerances as the causal check."""
    fs, n_hops = 44100.0, 70
    t = np.arange(hop * n_hops) / fs
    x = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1320 * t)
         + 0.3 * np.random.default_rng(3).uniform(-1, 1, t.size) * (np.arange(t.size) % 4096 < 200)).astype(np.float32)
    h = o.HPR(fs, hop, 2.0, o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE | o.OUTPUT_RESIDUAL, o.TIME_ANTICAUSAL)
    if soft:
        h.use_soft_mask()
    got = h.process_stream(x)
    model, _ = _model_hpr_stream(x, fs, hop, 2.0, soft, False)
    for k in ("P", "H") + (() if soft else ("R",)):
        scale = np.sqrt(np.mean(model[k] ** 2)) + 1e-30
        err = (got[k].astype(np.float64) - model[k]) / scale
        if soft:
            assert np.sqrt(np.mean(err ** 2)) < 2e-6, k
        else:
            assert np.sqrt(np.mean(err ** 2)) < 1e-3 and np.mean(np.abs(err) < 1e-4) > 0.9, k


@pytest.mark.parametrize("soft", [True, False])
def test_oracle_two_pass_offline_agrees_with_float64_model(soft):
    """HPRIOffline end to end (both anticausal passes, the in-place shifts, the stale tail of Q9) against the
    float64 model.  With soft masks the two differ by float32 round-off only; with hard masks single mask
    decisions flip and the first pass feeds the second, so the bound is looser."""
    fs, hop_h, hop_p = 44100.0, 1024, 256
    n = 1024 * 28 + 11
    t = np.arange(n) / fs
    x = (0.3 * np.sin(2 * np.pi * 330 * t) + 0.2 * np.sin(2 * np.pi * 990 * t)
         + 0.4 * np.random.default_rng(4).uniform(-1, 1, n) * (np.arange(n) % 5000 < 150)).astype(np.float32)
    eng = o.HPRIOffline(fs, hop_h, hop_p, 2.0, 2.0)
    if soft:
        eng.use_soft_mask()
    harm, perc, _ = eng.process(x)
    mh, mp = _model_hpri_offline(x, fs, hop_h, hop_p, 2.0, soft)
    for g, m, name in ((harm, mh, "harm"), (perc, mp, "perc")):
        scale = np.sqrt(np.mean(m ** 2)) + 1e-30
        err = (g.astype(np.float64) - m) / scale
        if soft:
            assert np.sqrt(np.mean(err ** 2)) < 5e-6, name
        else:
            assert np.sqrt(np.mean(err ** 2)) < 5e-3 and np.mean(np.abs(err) < 1e-3) > 0.9, name


def test_repeated_multiplication_stands_in_for_powf():
    """The reference raises with powf (soft_mask_functor hps.h:116-129: powf(x, int(beta)); SSE
    complex_abs_squared_functor hps.h:91-98: powf(abs, 2)); oracle and kernels multiply repeatedly.  Measured
    here on the magnitudes of a real clip: x*x and x*x*x differ from glibc's powf(x, 2) / powf(x, 3) by at most
    one ulp (powf is not correctly rounded; x*x is), and the soft masks built from either agree to 2e-7
    absolute -- far inside the 1e-5 RMS the north star allows on the waveforms."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    rng = np.random.default_rng(6)
    mags = np.abs(np.fft.fft(rng.uniform(-1, 1, 4096) * np.hanning(4096))).astype(np.float32)
    mags = np.concatenate([mags, rng.uniform(0, 5000, 4000).astype(np.float32), np.float32([0, 1e-20, 1e18])])
    for p in (2, 3):
        lib = np.float32([libm.powf(float(v), float(p)) for v in mags])
        rep = mags.copy()
        for _ in range(p - 1):
            rep = (rep * mags).astype(np.float32)
        ulp = np.spacing(np.maximum(np.abs(lib), np.float32(1e-38)))
        finite = np.isfinite(lib)
        assert np.all(np.abs(lib[finite].astype(np.float64) - rep[finite]) <= ulp[finite])
        if p == 2:                                 # in practice glibc's powf(x, 2) IS the rounded product almost always
            assert np.mean(lib[finite] == rep[finite]) > 0.99
        eps = np.finfo(np.float32).eps
        other = np.roll(mags, 7)
        lib_o = np.float32([libm.powf(float(v), float(p)) for v in other])
        rep_o = other.copy()
        for _ in range(p - 1):
            rep_o = (rep_o * other).astype(np.float32)
        with np.errstate(invalid="ignore", over="ignore"):
            m_lib = lib / (lib + lib_o + eps)
            m_rep = rep / (rep + rep_o + eps)
        ok = np.isfinite(m_lib) & np.isfinite(m_rep)
        assert np.max(np.abs(m_lib[ok].astype(np.float64) - m_rep[ok])) <= 2e-7


def test_waveforms_with_literal_powf_stay_within_the_north_star_tolerance(tmp_path):
    """The oracle rebuilt with the reference's literal powf calls (-DZO_LITERAL_POWF: soft_mask_functor
    hps.h:116-129, complex_abs_squared_functor hps.h:91-98) against the shipped one (repeated multiplication) on
    the soft-mask and SSE configurations of the golden fixtures: relative RMS difference of the separated
    waveforms <= 1e-5 (the north star's tolerance; measured: 0 .. 3e-8)."""
    import ctypes as C
    import subprocess
    so = str(tmp_path / "libzen_oracle_powf.so")
    src = os.path.join(os.path.dirname(o.__file__), "zen_oracle.c")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-fPIC", "-shared", "-DZO_LITERAL_POWF",
                           "-o", so, src, "-lm"])
    L = C.CDLL(so)
    fp = C.POINTER(C.c_float)
    L.zo_hpr_create.restype = C.c_void_p
    L.zo_hpr_create.argtypes = [C.c_float, C.c_size_t, C.c_float, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_int)]
    for name in ("zo_hpr_use_sse_filter", "zo_hpr_use_soft_mask", "zo_hpr_destroy"):
        getattr(L, name).argtypes = [C.c_void_p]
        getattr(L, name).restype = None
    L.zo_hpr_process_next_hop.argtypes = [C.c_void_p, fp]
    L.zo_hpr_process_next_hop.restype = None
    for name in ("zo_hpr_percussive_out", "zo_hpr_harmonic_out"):
        getattr(L, name).argtypes = [C.c_void_p]
        getattr(L, name).restype = fp

    def literal(fs, hop, beta, caus, x, soft, sse):
        err = C.c_int(0)
        h = L.zo_hpr_create(fs, hop, beta, o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE, caus, 1, C.byref(err))
        assert h
        if soft:
            L.zo_hpr_use_soft_mask(h)
        if sse:
            L.zo_hpr_use_sse_filter(h)
        outs = {"P": [], "H": []}
        for i in range(x.size // hop):
            hopbuf = np.ascontiguousarray(x[i * hop:(i + 1) * hop], np.float32)
            L.zo_hpr_process_next_hop(h, hopbuf.ctypes.data_as(fp))
            outs["P"].append(np.ctypeslib.as_array(L.zo_hpr_percussive_out(h), (hop,)).copy())
            outs["H"].append(np.ctypeslib.as_array(L.zo_hpr_harmonic_out(h), (hop,)).copy())
        L.zo_hpr_destroy(h)
        return {k: np.concatenate(v) for k, v in outs.items()}

    rng = np.random.default_rng(9)
    for fs, hop, beta, caus, soft, sse in ((44100.0, 1024, 2.0, o.TIME_CAUSAL, True, False),
                                           (44100.0, 256, 3.0, o.TIME_ANTICAUSAL, True, False),
                                           (44100.0, 512, 2.0, o.TIME_CAUSAL, False, True),
                                           (44100.0, 512, 2.0, o.TIME_CAUSAL, True, True)):
        n = hop * 40
        t = np.arange(n) / fs
        x = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.3 * rng.uniform(-1, 1, n) * (np.arange(n) % 3000 < 120)).astype(np.float32)
        h = o.HPR(fs, hop, beta, o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE, caus)
        if soft:
            h.use_soft_mask()
        if sse:
            h.use_sse_filter()
        ref = h.process_stream(x)
        lit = literal(fs, hop, beta, caus, x, soft, sse)
        for k in "PH":
            a, b = ref[k].astype(np.float64), lit[k].astype(np.float64)
            rms = np.sqrt(np.mean(b ** 2)) + 1e-30
            assert np.sqrt(np.mean((a - b) ** 2)) / rms <= 1e-5, (hop, soft, sse, k)


def test_hard_mask_without_divide_is_exact():
    """zen_amd/csrc/masks.h hard_mask_exact: fl(x / d) >= beta (hard_mask_functor, hps.h:100-113) decided as
    (double)x > thr * (double)d, thr the rounding boundary below beta (times 1 - 2^-50 where the boundary itself
    counts: even beta).  The same IEEE operations in numpy
    (float32 division is correctly rounded, the float64 product of a 25-bit and a 24-bit number is exact),
    on pairs placed a few ulps around the boundary, on random pairs, and on the special values."""
    rng = np.random.default_rng(5)

    def threshold(beta):
        b = np.float32(beta)
        u = b.view(np.uint32)
        pred = np.uint32(u - 1).view(np.float32)
        m = (np.float64(pred) + np.float64(b)) * 0.5
        return m * (1.0 - 2.0 ** -50) if (int(u) & 1) == 0 else m     # masks.h hard_mask_threshold

    def exact(x, d, beta):
        thr = threshold(beta)
        with np.errstate(invalid="ignore", over="ignore"):
            t = thr * d.astype(np.float64)      # (+inf for d = inf: nothing exceeds it under the strict comparison)
            return x.astype(np.float64) > t

    for beta in (2.0, np.float32(2.0) - np.float32(1.1920929e-07), 2.5, 1.0, 3.0, 0.1, 1.7, 1e-30, 3e38):
        beta = np.float32(beta)
        d = np.concatenate([rng.uniform(1e-7, 1e4, 200000), np.exp(rng.uniform(-80, 80, 200000))]).astype(np.float32)
        with np.errstate(over="ignore", under="ignore"):
            base = (beta * d).astype(np.float32)
            # the boundary sits within a couple of ulps of beta * d: walk +-4 ulps around it
            xs = [np.nextafter(base, np.float32(np.inf)) for _ in range(1)]
            x = base.copy()
            cases = [base]
            up, dn = base.copy(), base.copy()
            for _ in range(4):
                up = np.nextafter(up, np.float32(np.inf))
                dn = np.nextafter(dn, np.float32(-np.inf))
                cases += [up.copy(), dn.copy()]
            cases.append(rng.uniform(0, 1e4, d.size).astype(np.float32))
            for x in cases:
                ref = (x / d) >= beta
                assert np.array_equal(ref, exact(x, d, beta)), float(beta)
    # special values: the mask must be false for NaN anywhere and for d = inf, true for x = inf over finite d
    beta = np.float32(2.0)
    x = np.array([np.inf, np.inf, 1.0, np.nan, 1.0, 0.0, 3.0e38], np.float32)
    d = np.array([np.inf, 1.0, np.inf, 1.0, np.nan, 1.0, 1.0e-3], np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        ref = (x / d) >= beta
    assert np.array_equal(ref, exact(x, d, beta)) and list(ref) == [False, True, False, False, False, False, True]
