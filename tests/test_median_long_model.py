"""CPU models of the two wave-cooperative primitives of zen_amd/csrc/median.hip's median_long_kernel (masks no register window
holds: libzen/mfilt.h:296-305 accepts any filter_len <= the dimension), checked on their own -- the GPU tier compares the kernel
with the oracle for a handful of lengths, this pins the constructions for EVERY length:

  * wave_sort: the bitonic network in its all-ascending form (first step of a merge compares element i with its mirror image in
    the block, then half-cleaners) for the next power of two, with every comparator that reaches past n left out.  Claim: it
    sorts any n keys.  Zero-one principle: exhaustively for n <= 13, random 0/1 and random keys beyond.
  * wave_rank: number of keys < v in a sorted window, 64 probes per round.

The index arithmetic below is the kernel's, line for line."""
import itertools

import numpy as np
import pytest


def wave_sort_schedule(n):
    """[(ia, ib), ...] per step, as median.hip wave_sort visits them (comparators with ib >= n left out)."""
    P = 1
    while P < n:
        P <<= 1
    steps = []
    k = 2
    while k <= P:
        hk = k >> 1
        i = np.arange(P >> 1)
        blk, off = i // hk, i % hk
        ia, ib = blk * k + off, blk * k + k - 1 - off
        steps.append((ia[ib < n], ib[ib < n]))
        j = k >> 2
        while j >= 1:
            q = i // j
            ia = q * 2 * j + (i - q * j)
            ib = ia + j
            steps.append((ia[ib < n], ib[ib < n]))
            j >>= 1
        k <<= 1
    return steps


def run_network(x, steps):
    x = x.copy()
    for ia, ib in steps:
        a, b = x[..., ia], x[..., ib]
        x[..., ia], x[..., ib] = np.minimum(a, b), np.maximum(a, b)
    return x


@pytest.mark.parametrize("n", list(range(1, 14)))
def test_wave_sort_zero_one_exhaustive(n):
    steps = wave_sort_schedule(n)
    allv = np.array(list(itertools.product((0, 1), repeat=n)), dtype=np.int32)
    out = run_network(allv, steps)
    assert np.array_equal(out, np.sort(allv, axis=1))


@pytest.mark.parametrize("n", [14, 17, 31, 33, 63, 64, 65, 100, 127, 129, 255, 257, 1000, 2049, 4999, 16383])
def test_wave_sort_random(n):
    rng = np.random.default_rng(n)
    steps = wave_sort_schedule(n)
    reps = 400 if n <= 300 else 12
    zo = rng.integers(0, 2, (reps, n)).astype(np.int32)                      # zero-one inputs with random densities
    zo &= (rng.uniform(0, 1, (reps, 1)) < rng.uniform(0, 1, (reps, n))).astype(np.int32) | zo
    keys = rng.integers(-2**31, 2**31 - 1, (reps, n)).astype(np.int64)
    few = rng.integers(-3, 4, (reps, n)).astype(np.int64)                     # many ties
    for x in (zo, keys, few):
        assert np.array_equal(run_network(x, steps), np.sort(x, axis=1))
    for ia, ib in steps:                                                      # a step's comparators are disjoint: one wave
        idx = np.concatenate([ia, ib])                                        # instruction's loads all precede its stores
        assert idx.size == np.unique(idx).size and (ib < n).all() and (ia < ib).all()


def wave_rank(win, v):
    """median.hip wave_rank: keys < v in the sorted win, 64 probes per round."""
    lo, n, rounds = 0, len(win), 0
    lanes = np.arange(64)
    while n > 0:
        chunk = (n + 63) >> 6
        i = lo + (lanes + 1) * chunk - 1
        below = (i < lo + n) & (win[np.minimum(i, len(win) - 1)] < v)
        assert not np.any(np.diff(below.astype(int)) > 0)      # the comparisons are a prefix of the lanes (popcount of the ballot)
        c = int(below.sum())
        end = lo + n
        lo += c * chunk
        n = min(chunk - 1, end - lo)
        rounds += 1
    return lo, rounds


@pytest.mark.parametrize("w", [1, 2, 3, 63, 64, 65, 130, 200, 257, 4095, 4097, 16383, 39999])
def test_wave_rank_is_lower_bound(w):
    rng = np.random.default_rng(w)
    win = np.sort(rng.integers(-50, 50, w) if w < 300 else rng.integers(-10**6, 10**6, w))
    probes = list(rng.integers(win[0] - 2, win[-1] + 3, 200)) + [win[0], win[-1], win[0] - 1, win[-1] + 1]
    for v in probes:
        r, rounds = wave_rank(win, v)
        assert r == int(np.searchsorted(win, v, side="left"))
        assert rounds <= 4                                       # 64-ary: 3 rounds to 64^3 keys (+1 for the ragged last chunk)


def test_sliding_window_model_against_brute_force():
    """The whole slide as the kernel does it (remove the leaving key at its rank, insert the entering one at its own) on a
    short line with a replicate border: the median of every window."""
    rng = np.random.default_rng(5)
    for w, n in ((5, 40), (9, 9), (33, 70), (65, 64)):
        x = rng.integers(-4, 5, n)
        mid = w // 2

        def tap(u):
            return x[min(max(u, 0), n - 1)]
        win = np.sort(np.array([tap(u) for u in range(-mid, mid + 1)]))
        for o in range(n):
            assert win[mid] == np.sort([tap(u) for u in range(o - mid, o + mid + 1)])[mid]
            vo, vi = tap(o - mid), tap(o + 1 + mid)
            if vi == vo:
                continue
            p_out, _ = wave_rank(win, vo)
            c_in, _ = wave_rank(win, vi)
            if vi > vo:                        # positions [p_out, c_in - 1) take their upper neighbour, vi lands at c_in - 1
                win[p_out:c_in - 1] = win[p_out + 1:c_in].copy()
                win[c_in - 1] = vi
            else:                              # vi lands at c_in, positions (c_in, p_out] take their lower neighbour
                win[c_in + 1:p_out + 1] = win[c_in:p_out].copy()
                win[c_in] = vi
            assert np.all(np.diff(win) >= 0)
