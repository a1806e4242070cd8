"""The index arithmetic of rt_wide.hip's two-step transforms, modelled in float32 numpy against the oracle's FFT.

rt_wide.hip cuts the radix-2 decimation-in-time DAG of an N = M*J point transform into
  step A: J independent M-point transforms of the decimated sequences x[j + n*J] that read every J-th twiddle, and
  step B: for every column kappa < M a J-point transform whose stage-t twiddle of frequency q is
          tw[(q * J/2^t) * M + kappa * J/2^t]   (the index of a plain J-point transform, scaled and shifted),
and claims that every butterfly is the one the one-piece transform evaluates.  This model runs both steps with
exactly those indices, one float32 operation at a time in the oracle's order (cmul: two products and one
sum / difference per component, each rounded), and must reproduce oracle/zen_oracle.c's transform bit for bit --
forward and inverse, for the sizes the kernel is built for."""
import numpy as np
import pytest

from oracle import oracle as o

f32 = np.float32


def cmul(wr, wi, br, bi):
    return (wr * br - wi * bi).astype(f32), (wr * bi + wi * br).astype(f32)


def dit_stages(xr, xi, tw_index, twr, twi, first_stage, n_stages, inverse):
    """Radix-2 DIT stages on the LAST axis of arrays shaped (..., L) holding sequences in natural order.
    Stage s (1-based within this sub-transform) combines the two half-size sub-transforms of the even / odd
    decimation; tw_index(stage, k) names the table entry of frequency k at that stage."""
    n = xr.shape[-1]
    assert n == 1 << n_stages

    def rec(r, i, stage_size):
        if stage_size == 1:
            return r, i
        er, ei = rec(r[..., 0::2], i[..., 0::2], stage_size // 2)
        orr, oi = rec(r[..., 1::2], i[..., 1::2], stage_size // 2)
        s = int(np.log2(stage_size))
        k = np.arange(stage_size // 2)
        idx = tw_index(first_stage + s, k)
        wr, wi = twr[idx], (-twi[idx] if inverse else twi[idx])
        tr, ti = cmul(wr, wi, orr, oi)
        return (np.concatenate([(er + tr).astype(f32), (er - tr).astype(f32)], axis=-1),
                np.concatenate([(ei + ti).astype(f32), (ei - ti).astype(f32)], axis=-1))
    return rec(xr, xi, n)


@pytest.mark.parametrize("log2n", [13, 14, 15])   # 15: fft_big.hip (nfft 32768, two kernels)
@pytest.mark.parametrize("inverse", [False, True])
def test_two_step_transform_is_the_one_piece_transform(log2n, inverse):
    n, log2m = 1 << log2n, 7
    m, j_count = 1 << log2m, n >> log2m
    log2j = log2n - log2m
    rng = np.random.default_rng(log2n)
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    ref = o.fft_c2c(x, inverse=inverse)
    tw = o.twiddles(n)[: n // 2]
    twr, twi = tw.real.astype(f32), tw.imag.astype(f32)

    # step A: row j holds x[j + n*J], n < M; stage s of the M-point transform reads tw[(k * M/2^s) * J]
    a = x.reshape(m, j_count).T.copy()                       # (J, M)
    ar, ai = dit_stages(a.real.astype(f32), a.imag.astype(f32),
                        lambda s, k: (k << (log2m - s)) << log2j, twr, twi, 0, log2m, inverse)
    # step B: column kappa holds Y_7[j][kappa], j < J; stage t reads tw[(q * J/2^t) * M + kappa * J/2^t]
    out = np.empty(n, np.complex64)
    br, bi = ar.T.copy(), ai.T.copy()                        # (M, J): row kappa
    for kappa in range(m):
        yr, yi = dit_stages(br[kappa], bi[kappa],
                            lambda t, q, kappa=kappa: ((q << (log2j - t)) << log2m) + (kappa << (log2j - t)),
                            twr, twi, 0, log2j, inverse)
        out[kappa + m * np.arange(j_count)] = yr + 1j * yi   # delivers X[kappa + M*q]
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
