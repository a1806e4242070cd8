"""GPU parity tests added in round 6.  Every call goes through the C-ABI (libzen_hip.so via ctypes) and is compared
BIT-EXACTLY (tolerance 0) with the CPU oracle.

Masks beyond 255 taps: the reference accepts any filter_len <= the filtered dimension (libzen/mfilt.h:296-305,
box.h:243-252), and l_perc = roundf(500 / (fs / nfft)) (hps.h:229) is 256 -> 257 taps wherever fs / hop = 7.8125
(32 kHz at the CLI's default hop_h 4096, zen/offline.h:19-32; 16 kHz at hop 2048; 8 kHz at hop 1024)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu
ALL = o.OUTPUT_HARMONIC | o.OUTPUT_PERCUSSIVE | o.OUTPUT_RESIDUAL
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZEN = os.path.join(ROOT, "zen_amd", "bin", "zen")


@pytest.fixture(scope="module")
def z():
    import zen_amd
    zen_amd.init(0)
    return zen_amd


def _matrix(rows, cols, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "ties":          # few distinct values: every rank search meets runs of equal keys.  (No -0.0 next to +0.0: they
        return rng.integers(-3, 4, (rows, cols)).astype(np.float32)   # compare equal, which of the two bit patterns a median
                                                                      # returns is the sorting algorithm's business, IPP's unknown)
    if kind == "ramp":          # monotone lines: every slide shifts the whole window or nothing
        return (np.arange(rows * cols, dtype=np.float32).reshape(rows, cols) * (1 if seed & 1 else -1)).astype(np.float32)
    return rng.uniform(-1, 1, (rows, cols)).astype(np.float32)


# (rows, cols, filter_len): register windows of 5..32 registers (257..2047 taps), the window in LDS (beyond), even
# lengths (made odd, mfilt.h:305), a length equal to the dimension, and one window that does not fit the LDS
LONG_FREQ = [(5, 700, 257), (4, 700, 256), (6, 4096, 257), (3, 16384, 256), (70, 2048, 257), (3, 1500, 511), (3, 1300, 600), (3, 2500, 1025), (2, 2600, 1536), (2, 3000, 2047),
             (2, 3000, 2049), (2, 2100, 2100), (3, 5000, 4999), (1, 9000, 8191), (2, 20000, 16383)]
LONG_TIME = [(700, 70, 257), (300, 130, 299), (300, 64, 300), (1300, 33, 1025), (2200, 5, 2049), (2500, 3, 2500)]


@pytest.mark.parametrize("rows,cols,flen", LONG_FREQ)
@pytest.mark.parametrize("kind", ["uniform", "ties"])
def test_median_frequency_masks_beyond_255_taps(z, rows, cols, flen, kind):
    a = _matrix(rows, cols, flen, kind)
    got = z.MedianFilterGPU(rows, cols, flen, z.FREQUENCY).filter_host(a)
    ref = o.median_filter(a, flen, o.FREQUENCY)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("rows,cols,flen", LONG_TIME)
@pytest.mark.parametrize("kind", ["uniform", "ties"])
@pytest.mark.parametrize("direction", ["causal", "anticausal"])
def test_median_time_masks_beyond_255_taps(z, rows, cols, flen, kind, direction):
    a = _matrix(rows, cols, flen + 1, kind)
    d = z.TIME_CAUSAL if direction == "causal" else z.TIME_ANTICAUSAL    # (the same on the CPU backend, mfilt.h:311-314)
    got = z.MedianFilterGPU(rows, cols, flen, d).filter_host(a)
    ref = o.median_filter(a, flen, o.TIME_CAUSAL if direction == "causal" else o.TIME_ANTICAUSAL)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_median_monotone_lines_long_masks(z):
    for rows, cols, flen, d, od in ((3, 3000, 2301, "FREQUENCY", o.FREQUENCY), (2400, 4, 2201, "TIME_ANTICAUSAL", o.TIME_ANTICAUSAL),
                                    (3, 900, 401, "FREQUENCY", o.FREQUENCY)):
        for seed in (0, 1):
            a = _matrix(rows, cols, seed, "ramp")
            got = z.MedianFilterGPU(rows, cols, flen, getattr(z, d)).filter_host(a)
            assert np.array_equal(got, o.median_filter(a, flen, od))


def test_median_window_beyond_the_lds_goes_through_device_memory(z):
    """39 999 taps: 160 KB of keys, more than a workgroup's LDS -- the window lives in a scratch buffer in device memory."""
    rows, cols, flen = 2, 40100, 39999
    a = _matrix(rows, cols, 7, "uniform")
    a[1] = _matrix(1, cols, 8, "ties")[0]
    got = z.MedianFilterGPU(rows, cols, flen, z.FREQUENCY).filter_host(a)
    assert np.array_equal(got.view(np.uint32), o.median_filter(a, flen, o.FREQUENCY).view(np.uint32))


def test_median_too_big_still_throws(z):
    with pytest.raises(z.ZgException):
        z.MedianFilterGPU(9, 300, 301, z.FREQUENCY)           # mfilt.h:296-303: compared before the length is made odd
    with pytest.raises(z.ZgException):
        z.MedianFilterGPU(299, 9, 300, z.TIME_ANTICAUSAL)
    z.MedianFilterGPU(9, 300, 300, z.FREQUENCY)                # equal to the dimension: accepted (301 taps)


@pytest.mark.parametrize("rows,cols,flen,direction", [(5, 2000, 257, "f"), (3, 3000, 1024, "f"), (2, 9000, 8191, "f"),
                                                      (1, 45000, 44001, "f"), (400, 70, 257, "t"), (1200, 9, 1025, "t")])
def test_box_masks_beyond_255_taps(z, rows, cols, flen, direction):
    a = _matrix(rows, cols, flen, "uniform")
    d, od = (z.FREQUENCY, o.FREQUENCY) if direction == "f" else (z.TIME_ANTICAUSAL, o.TIME_ANTICAUSAL)
    got = z.BoxFilterGPU(rows, cols, flen, d).filter_host(a)
    assert np.array_equal(got.view(np.uint32), o.box_filter(a, flen, od).view(np.uint32))


# ---------------------------------------------------------------------------- the engine
def _stream(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, n).astype(np.float32)
    x *= (rng.uniform(0, 1, n) < 0.8)
    return x


# fs / hop = 7.8125: l_harm 1, l_perc 256 -> 257 taps; 31 kHz at hop 4096: 264 -> 265 taps
LONG_ENGINES = [(32000.0, 4096, 10), (16000.0, 2048, 10), (8000.0, 1024, 12), (4000.0, 512, 12), (2000.0, 256, 14),
                (31000.0, 4096, 8)]


@pytest.mark.parametrize("fs,hop,n_hops", LONG_ENGINES)
@pytest.mark.parametrize("mode", ["hard", "soft", "sse"])
@pytest.mark.parametrize("causal", [True, False])
def test_engine_frequency_masks_of_257_taps(z, fs, hop, n_hops, mode, causal):
    oc, zc = (o.TIME_CAUSAL, z.TIME_CAUSAL) if causal else (o.TIME_ANTICAUSAL, z.TIME_ANTICAUSAL)
    ref_e = o.HPR(fs, hop, 2.0, ALL, oc)
    assert (ref_e.l_perc | 1) > 255
    x = _stream(hop * n_hops, hop + int(fs))
    g = z.HPR(fs, hop, 2.0, ALL, zc, True, 1, 0)
    if mode == "soft":
        ref_e.use_soft_mask()
        g.use_soft_mask()
    if mode == "sse":
        ref_e.use_sse_filter()
        g.use_sse_filter()
    ref = ref_e.process_stream(x)
    for block in (1, n_hops):              # hop by hop (the reference's API), and one block call
        g.reset_buffers()
        got = g.process_stream_host(x, block=block)
        for k in "PHR":
            assert np.array_equal(got[k], ref[k], equal_nan=True), (k, block)


@pytest.mark.parametrize("fs,hop,causal,n_hops", [(96000.0, 32, True, 60), (96000.0, 32, False, 430), (128000.0, 32, True, 40),
                                                  (128000.0, 32, False, 300)])
@pytest.mark.parametrize("mode", ["hard", "sse"])
def test_engine_sliding_matrices_beyond_255_rows(z, fs, hop, causal, n_hops, mode):
    """hps.bench.cu:62-64 runs hop 32 at 48 kHz; at 96 kHz the sliding matrix has 400 rows (201 time taps), at 128 kHz 534 rows
    and a 267-tap anticausal time mask -- the longest an engine can have: beyond fs / hop = 4000 l_perc rounds to 0."""
    oc, zc = (o.TIME_CAUSAL, z.TIME_CAUSAL) if causal else (o.TIME_ANTICAUSAL, z.TIME_ANTICAUSAL)
    flags = o.OUTPUT_PERCUSSIVE | o.OUTPUT_HARMONIC
    ref_e = o.HPR(fs, hop, 2.0, flags, oc)
    assert ref_e.stft_width > 255
    g = z.HPR(fs, hop, 2.0, flags, zc, True, 1, 0)
    if mode == "sse":
        ref_e.use_sse_filter()
        g.use_sse_filter()
    x = _stream(hop * n_hops, 5)
    ref = ref_e.process_stream(x)
    got = g.process_stream_host(x, block=n_hops if not causal else 7)
    for k in "PH":
        assert np.array_equal(got[k], ref[k], equal_nan=True), k
    if not causal:
        assert np.abs(ref["P"]).max() > 0      # the stream is longer than the lag: the comparison is not of zeros


@pytest.mark.parametrize("fs,hop_h,hop_p", [(32000.0, 4096, 256), (16000.0, 2048, 128), (8000.0, 1024, 128)])
@pytest.mark.parametrize("mode", ["hard", "soft", "sse"])
def test_offline_two_pass_with_257_tap_pass(z, fs, hop_h, hop_p, mode):
    n = 9 * hop_h + 1234
    x = _stream(n, int(fs) + hop_p)
    ro = o.HPRIOffline(fs, hop_h, hop_p, 2.0, 2.0)
    g = z.HPRIOffline(fs, hop_h, hop_p, 2.0, 2.0)
    if mode == "soft":
        ro.use_soft_mask()
        g.use_soft_mask()
    if mode == "sse":
        ro.use_sse_filter()
        g.use_sse_filter()
    ref = ro.process(x)
    got = g.process(x)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b, equal_nan=True)


def test_cli_offline_defaults_on_a_32k_file(tmp_path):
    """`zen offline -i <32 kHz wav> --hps`: the CLI's default hops 4096 / 256 (zen/offline.h:19-32) give pass 1 a 257-tap
    frequency mask."""
    from tests.test_cpp_host import pcm16, read_wav_pcm16, write_wav_pcm16
    fs, n = 32000, 70001
    rng = np.random.default_rng(3)
    t = np.arange(n) / fs
    s = 0.3 * np.sin(2 * np.pi * 330 * t) + 0.1 * rng.uniform(-1, 1, n)
    s[::8000] += 0.5
    pcm = np.round(s * 20000).astype(np.int16)
    wav = str(tmp_path / "in32k.wav")
    write_wav_pcm16(wav, pcm, fs, channels=1)
    mono = pcm.astype(np.float32) / np.float32(32767.0)
    r = subprocess.run([ZEN, "offline", "-i", wav, "--hps", "-o", str(tmp_path / "off")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "harmonic params: 4096,2" in r.stdout
    h, p, _ = o.HPRIOffline(float(fs), 4096, 256, 2.0, 2.0).process(mono)
    for name, ref in (("off_harm.wav", h), ("off_perc.wav", p)):
        rfs, got = read_wav_pcm16(str(tmp_path / name))
        peak = max(-ref.min(), ref.max())
        assert rfs == fs and np.array_equal(got.astype(np.int64), pcm16(ref / np.float32(peak))), name


# ---------------------------------------------------------------------------- zen_hip_hpr_process_host
@pytest.mark.parametrize("hop,flags,mode", [(1024, o.OUTPUT_PERCUSSIVE, "hard"), (1024, ALL, "hard"), (256, ALL, "soft"),
                                            (512, o.OUTPUT_PERCUSSIVE | o.OUTPUT_HARMONIC, "sse"), (2048, o.OUTPUT_HARMONIC, "hard")])
@pytest.mark.parametrize("pinned", [True, False])
def test_block_of_hops_from_host_buffers(z, hop, flags, mode, pinned):
    """zen_hip_hpr_process_host: the timed region of zen/fakert.h:221-247 (host hop in, process, host hop out) for a block:
    pieces go up / are processed / come down on three streams.  Pinned and pageable buffers, piece lengths that do and do
    not divide the block, two calls continuing one stream: the samples of the per-hop oracle."""
    fs, n_hops = 44100.0, 57
    x = _stream(hop * n_hops, hop + flags)
    ref_e = o.HPR(fs, hop, 2.0, flags, o.TIME_CAUSAL)
    g = z.HPR(fs, hop, 2.0, flags, z.TIME_CAUSAL, True, 1, 0)
    if mode == "soft":
        ref_e.use_soft_mask()
        g.use_soft_mask()
    if mode == "sse":
        ref_e.use_sse_filter()
        g.use_sse_filter()
    ref = ref_e.process_stream(x)
    keep = []

    def buf(n):
        if pinned:
            keep.append(z.PinnedHost(n))
            return keep[-1].array
        return np.empty(n, np.float32)
    xin = buf(x.size)
    xin[:] = x
    want = {"P": bool(flags & o.OUTPUT_PERCUSSIVE), "H": bool(flags & o.OUTPUT_HARMONIC),
            "R": bool(flags & o.OUTPUT_RESIDUAL) and mode == "hard"}
    try:
        for piece in (0, 5, 19, 64):
            z.set_option("host_block_hops", piece)
            g.reset_buffers()
            outs = {k: (buf(x.size) if w else None) for k, w in want.items()}
            for a in outs.values():
                if a is not None:
                    a[:] = np.nan
            cut = 23 * hop                           # two calls: the second continues the stream of the first
            g.process_host(xin[:cut], harm=None if outs["H"] is None else outs["H"][:cut],
                           perc=None if outs["P"] is None else outs["P"][:cut], resid=None if outs["R"] is None else outs["R"][:cut])
            g.process_host(xin[cut:], harm=None if outs["H"] is None else outs["H"][cut:],
                           perc=None if outs["P"] is None else outs["P"][cut:], resid=None if outs["R"] is None else outs["R"][cut:])
            for k, a in outs.items():
                if a is not None:
                    assert np.array_equal(a, ref[k], equal_nan=True), (k, piece)
    finally:
        z.set_option("host_block_hops", 0)
        for b in keep:
            b.free()


def test_block_from_host_refuses_overlap_and_several_streams(z):
    g = z.HPR(44100.0, 256, 2.0, o.OUTPUT_PERCUSSIVE, z.TIME_CAUSAL, True, 1, 0)
    x = np.zeros(256 * 8, np.float32)
    with pytest.raises(z.ZenHipError):
        g.process_host(x, perc=x)
    g2 = z.HPR(44100.0, 256, 2.0, o.OUTPUT_PERCUSSIVE, z.TIME_CAUSAL, True, 2, 0)
    with pytest.raises(z.ZenHipError):
        g2.process_host(x, perc=np.zeros_like(x))


# ---------------------------------------------------------------------------- zen_hip_hpri_process_sink
@pytest.mark.parametrize("n,rng_len,want", [(3 * 1323000 + 77, 0, (True, True)), (700001, 4096 * 40, (True, True)),
                                            (700001, 4096 * 40, (False, True)), (30000, 0, (True, False)), (1500000, 4096 * 31, (True, True))])
def test_offline_ranges_handed_to_a_sink(z, n, rng_len, want):
    """HPRIOffline::process builds its result vectors from this (round 6): every range of the clip is handed over exactly
    once per wanted output, in ascending order, from pinned staging memory -- and the samples are those of the whole-clip call."""
    fs = 44100.0
    x = _stream(n, n % 1000)
    g = z.HPRIOffline(fs, 4096, 256, 2.0, 2.0)
    z.set_option("offline_range", rng_len)
    try:
        h, p, ranges = g.process_sink(x, want)
        rh, rp, _ = g.process(x)
    finally:
        z.set_option("offline_range", 0)
    for got, ref, w, rr in ((h, rh, want[0], ranges[0]), (p, rp, want[1], ranges[1])):
        if not w:
            assert got is None and rr == []
            continue
        assert np.array_equal(got, ref, equal_nan=True)
        pos = 0
        for b, c in rr:
            assert b == pos and c > 0
            pos += c
        assert pos == n
    if rng_len:
        assert len(ranges[1]) == -(-n // rng_len)      # more ranges than staging slots: the slots are reused
