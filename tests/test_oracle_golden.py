"""Pins the CPU oracle against the reference's own known-answer tests.

Source of the vectors: libzen/mfilt.test.cu (sevagh/Zen).  The fixture (`:315-337`) is a zero matrix whose
middle ROW (i == x/2) is 5 and whose middle COLUMN (j == y/2) is 8, layout data[i*y + j], i = time,
j = frequency.  Expectations are the CPU ones (`:407-591`): time median keeps the 8-column and erases the
5-row, frequency median the converse.  With IPP's replicate border the result holds everywhere (the
reference's own GPU-copybord tests `:701-886` assert the everywhere form); the reference CPU tests only
look inside a sub-range, which is checked literally as well.
"""
import numpy as np
import pytest

from oracle import oracle as o

SHAPES = [(9, 9, 3), (10, 20, 5), (1024, 128, 5)]  # mfilt.test.cu:375-405 CPU fixtures


def stripes(x, y):
    d = np.zeros((x, y), np.float32)
    d[x // 2, :] = 5
    d[:, y // 2] = 8   # j == y/2 is assigned last in the reference loop, so the crossing holds 8
    return d


@pytest.mark.parametrize("x,y,f", SHAPES)
@pytest.mark.parametrize("direction", [o.TIME_CAUSAL, o.TIME_ANTICAUSAL])
def test_time_median_keeps_column_erases_row(x, y, f, direction):
    res = o.median_filter(stripes(x, y), f, direction)
    # literal reference assertions (mfilt.test.cu:407-424 causal, :536-553 anticausal)
    for i in range(x):
        for j in range(y):
            if direction == o.TIME_CAUSAL:
                inside = j == y // 2 and i > (3 if (x, y) == (9, 9) else 5)
            else:
                inside = j == y // 2 and 2 < i < x - 3
            if inside:
                assert res[i, j] == 8
            elif j != y // 2:
                assert res[i, j] == 0
    # replicate border => holds everywhere
    exp = np.zeros((x, y), np.float32)
    exp[:, y // 2] = 8
    assert np.array_equal(res, exp)


@pytest.mark.parametrize("x,y,f", SHAPES)
def test_frequency_median_keeps_row_erases_column(x, y, f):
    res = o.median_filter(stripes(x, y), f, o.FREQUENCY)
    lim = 3 if (x, y) == (9, 9) else 5
    for i in range(x):          # mfilt.test.cu:464-483, :485-503, :505-523
        for j in range(y):
            if i == x // 2 and j < y - lim:
                assert res[i, j] == 5
            elif i != x // 2:
                assert res[i, j] == 0
    exp = np.zeros((x, y), np.float32)
    exp[x // 2, :] = 5
    assert np.array_equal(res, exp)


@pytest.mark.parametrize("direction", [o.FREQUENCY, o.TIME_CAUSAL, o.TIME_ANTICAUSAL])
def test_degenerate_filter_too_big(direction):
    # mfilt.test.cu:525-534 : MedianFilterCPU(9, 9, 171, dir) throws ZgException
    with pytest.raises(o.OracleError) as e:
        o.median_filter(np.zeros((9, 9), np.float32), 171, direction)
    assert e.value.code == o.E_FILTER_TOO_BIG
    with pytest.raises(o.OracleError):
        o.box_filter(np.zeros((9, 9), np.float32), 171, direction)  # box.h:243-250


def test_unused_large_square_fixture():
    # mfilt.test.cu:93-99 declares 1024x1024 / f=21 but never uses it; same property must hold
    x = y = 1024
    d = stripes(x, y)
    exp_t = np.zeros((x, y), np.float32)
    exp_t[:, y // 2] = 8
    assert np.array_equal(o.median_filter(d, 21, o.TIME_ANTICAUSAL), exp_t)
    exp_f = np.zeros((x, y), np.float32)
    exp_f[x // 2, :] = 5
    assert np.array_equal(o.median_filter(d, 21, o.FREQUENCY), exp_f)


def test_even_length_is_made_odd():
    # mfilt.h:305 : len += 1 - len%2, after the too-big check on the original length
    rng = np.random.default_rng(3)
    d = rng.uniform(0, 1, (16, 32)).astype(np.float32)
    assert np.array_equal(o.median_filter(d, 4, o.FREQUENCY), o.median_filter(d, 5, o.FREQUENCY))
    from scipy.ndimage import median_filter as sp_median
    assert np.array_equal(o.median_filter(d, 16, o.TIME_CAUSAL), sp_median(d, size=(17, 1), mode="nearest"))
    with pytest.raises(o.OracleError):
        o.median_filter(d, 17, o.TIME_CAUSAL)     # 17 > 16 rows: throws although 16 -> 17 is accepted


# ---- FFT: libzen/fftw.test.cu:16 AllowableFFTError = 2e-4 abs, n = 64 / 1024 / 16384 (:103-130) ----
@pytest.mark.parametrize("n", [64, 1024, 16384])
def test_fft_forward_inverse_within_reference_tolerance(n):
    rng = np.random.default_rng(n)
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)  # fftw.test.cu:18-36
    x64 = x.astype(np.complex128)
    fwd = o.fft_c2c(x)
    assert np.abs(fwd.real - np.fft.fft(x64).real).max() <= 2e-4
    assert np.abs(fwd.imag - np.fft.fft(x64).imag).max() <= 2e-4
    inv = o.fft_c2c(x, inverse=True)            # unnormalised: n * ifft
    ref = np.fft.ifft(x64) * n
    assert np.abs(inv.real - ref.real).max() <= 2e-4
    assert np.abs(inv.imag - ref.imag).max() <= 2e-4
    # IPP_FFT_NODIV_BY_ANY (fftw.h:70,86): backward(forward(x)) == nfft * x
    rt = o.fft_c2c(fwd, inverse=True)
    assert np.abs(rt / n - x).max() <= 2e-4


@pytest.mark.parametrize("n", [64, 1024])
def test_fft_huge_range_finiteness(n):
    # fftw.test.cu:38-58 "huge" inputs uniform(FLT_MIN, FLT_MAX): only finiteness has to agree
    rng = np.random.default_rng(7)
    x = (rng.uniform(0, 1, n) * 3.0e38 + 1j * rng.uniform(0, 1, n) * 3.0e38).astype(np.complex64)
    got = o.fft_c2c(x)
    with np.errstate(all="ignore"):
        ref = np.fft.fft(x.astype(np.complex128)).astype(np.complex64)
    both = np.isfinite(got.real) & np.isfinite(ref.real)
    # where both are finite they agree loosely; non-finite outputs are allowed to differ in kind
    assert both.sum() == 0 or np.allclose(got.real[both], ref.real[both], rtol=1e-3)


def test_fft_rejects_non_power_of_two():
    with pytest.raises(o.OracleError):
        o.fft_c2c(np.zeros(48, np.complex64))


# ---------------------------------------------------------------- box filter: the reference's own vectors
def box_column_case(x, y, f):
    """libzen/box.test.cu:30-60, :124-199 (the three enabled BoxFilter*UnitTestGPU.CausalTime cases): a zero
    matrix whose middle column is 8, reciprocal (1/x: the zeros become +inf), time-direction box filter,
    reciprocal again ((f+1)/x).  Expected: 8*(f+1) on the column (32 for f = 3, 48 for f = 5), 0 elsewhere."""
    d = np.zeros((x, y), np.float32)
    d[:, y // 2] = 8
    with np.errstate(divide="ignore"):
        rec = (np.float32(1.0) / d).astype(np.float32)
    exp = np.zeros((x, y), np.float32)
    exp[:, y // 2] = 8 * (f + 1)
    return rec, exp


@pytest.mark.parametrize("x,y,f", SHAPES)
def test_box_reference_column_vectors(x, y, f):
    rec, exp = box_column_case(x, y, f)
    res = o.box_filter(rec, f, o.TIME_CAUSAL)
    with np.errstate(divide="ignore"):
        back = (np.float32(f + 1.0) / res).astype(np.float32)
    assert np.array_equal(back, exp)


def test_ipp_probe_reports_what_it_found():
    """BASELINE.md section 3 / SURVEY 8(d): a configure-time probe for ipp.h + libipp*; where IPP exists the literal
    calls of the reference are run next to the restatement (oracle/ipp_check.c).  This pool has no IPP: the probe must
    say so, name where it looked, and leave the port as the baseline."""
    from oracle import ipp_probe
    p = ipp_probe.check()
    assert isinstance(p["found"], bool) and len(p["searched"]) >= 3
    if p["found"]:
        assert "report" in p or "error" in p
        if "report" in p:                      # the median is an order statistic: IPP and the restatement must agree exactly
            for c in p["report"]["cases"]:
                if c["case"].startswith("median"):
                    assert c["differing"] == 0, c
    else:
        assert "report" not in p
