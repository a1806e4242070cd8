"""CPU model of the in-place exchange mode of zen_amd/csrc/fft_dev.h (PassRunner under -DZEN_FFT_INPLACE, the 16384-point plan of
4 + 4 + 3 + 3 stages).  The GPU tier proves the variant bit-exact; this pins the three claims the
construction rests on, with the kernel's own index arithmetic:

  1. every thread of every pass reads, in place, exactly the operands the autosort (Stockham) layout hands it -- same values, same
     order -- so the butterflies cannot tell the two layouts apart;
  2. in place, a thread's stores of a pass go exactly where its loads came from, and different threads' sets are disjoint: the
     barrier between a pass's loads and its stores is not needed (only the last pass keeps one, for the next transform);
  3. with the XOR swizzle of Plan<14>::pad every wave instruction of every pass touches each of the 32 eight-byte bank pairs
     exactly twice (64 lanes x 8 bytes: the minimum).
"""
import numpy as np

LOG2N = 14
N = 1 << LOG2N
V = 16
TF = N // V
R_BITS = [4, 4, 3, 3]                       # Plan<14>: P = 4, BASE = 3, REM = 2
S = [sum(R_BITS[:p]) for p in range(len(R_BITS) + 1)]


def image_k(k, p):
    """fft_dev.h image_k: the digits c_0 .. c_{p-1} of k (c_0 lowest, r(0) bits) in the order they were produced, c_0 on top."""
    if p < 2:
        return k
    rv = np.zeros_like(k)
    for g in range(p):
        rv |= ((k >> S[g]) & ((1 << R_BITS[g]) - 1)) << (S[p] - S[g + 1])
    return rv


def pad_inplace(i):
    return i ^ ((((i >> 10) & 3) << 3) | (((i >> 13) & 1) << 2) | (((i >> 12) & 1) << 1) | ((i >> 6) & 1))


def pass_indices(p):
    """For pass p: per (thread, i, m): Stockham logical read index, in-place physical read index; per (thread, i, c): the
    Stockham logical write index and the in-place physical write index."""
    r = R_BITS[p]
    R = 1 << r
    NB = V // R
    log2J = LOG2N - S[p] - r
    J = 1 << log2J
    tf = np.arange(TF)[:, None, None]
    i = np.arange(NB)[None, :, None]
    m = np.arange(R)[None, None, :]
    b = tf + i * TF
    k, j = b >> log2J, b & (J - 1)
    st_read = (k * R + m) * J + j
    ip_read = (image_k(k, p) * R + m) * J + j
    st_write = b + m * (N // R)
    ip_write = ip_read                                  # output c goes where input m = c came from
    return st_read, ip_read, st_write, ip_write


def test_in_place_reads_the_operands_of_the_autosort_layout():
    stock = np.arange(N, dtype=np.int64)                # value ids: the input samples, in natural order in both images
    inpl = stock.copy()
    next_id = N
    for p in range(len(R_BITS)):
        st_read, ip_read, st_write, ip_write = pass_indices(p)
        a, b = stock[st_read], inpl[ip_read]
        assert np.array_equal(a, b), "pass %d: an in-place thread reads other operands than the autosort one" % p
        # the R results of a sub-transform: new ids, a function of (pass, thread, i, c) -- the same in both models because the
        # inputs were the same
        out = next_id + np.arange(a.size, dtype=np.int64).reshape(a.shape)
        next_id += a.size
        new_stock, new_inpl = np.empty_like(stock), np.empty_like(inpl)
        new_stock[st_write] = out
        new_inpl[ip_write] = out
        stock, inpl = new_stock, new_inpl
        assert np.unique(stock).size == N and np.unique(inpl).size == N       # a permutation each: nothing lost, nothing twice
    # (the last pass hands its results to the output functor by their natural index idx = b + c * N / R in both forms:
    # thread tf still owns idx = tf + slot * TF, slot = c * NB + i)
    r = R_BITS[-1]
    R, NB = 1 << r, V >> r
    tf = np.arange(TF)[:, None, None]
    i = np.arange(NB)[None, :, None]
    c = np.arange(R)[None, None, :]
    idx = (tf + i * TF) + c * (N // R)
    assert np.array_equal(idx, tf + (c * NB + i) * TF)


def test_a_thread_overwrites_only_what_it_read_itself():
    for p in range(len(R_BITS)):
        _, ip_read, _, ip_write = pass_indices(p)
        assert np.array_equal(np.sort(ip_read.reshape(TF, -1), axis=1), np.sort(ip_write.reshape(TF, -1), axis=1))
        assert np.unique(ip_read).size == N                 # the threads' sets partition the image: nobody else reads them
    # ... and the autosort layout is the counter-example that makes its first barrier necessary: a pass writes other positions
    st_read, _, st_write, _ = pass_indices(1)
    assert not np.array_equal(np.sort(st_read.reshape(TF, -1), axis=1), np.sort(st_write.reshape(TF, -1), axis=1))


def test_swizzled_image_is_bank_conflict_free_in_every_pass():
    for p in range(len(R_BITS)):
        _, ip_read, _, _ = pass_indices(p)
        if p == 0:
            # pass 0 reads its inputs from memory; its stores go to (c * J + j): the same formula with k = 0
            pass
        phys = pad_inplace(ip_read)                         # [thread, i, m]: one wave instruction = 64 consecutive threads, fixed i, m
        assert np.unique(phys).size == N and phys.max() < N  # the swizzle is a bijection of the unpadded image
        slots = phys % 32                                    # 8-byte bank pair of a float2 (64 banks of 4 bytes)
        for w0 in range(0, TF, 64):
            s = slots[w0:w0 + 64]                            # [64 lanes, i, m]
            for ii in range(s.shape[1]):
                for mm in range(s.shape[2]):
                    counts = np.bincount(s[:, ii, mm], minlength=32)
                    assert counts.min() == 2 and counts.max() == 2, (p, w0, ii, mm, counts)


def test_the_padded_image_is_why_the_first_in_place_build_lost():
    """With the shipped padding (one slot per 8 values) the digit reversal puts the lanes of a wave 1024 elements apart in the
    last two passes: one bank pair for eight or sixteen lanes (offline batch 5.93 -> 7.51 ms)."""
    _, ip_read, _, _ = pass_indices(3)
    phys = ip_read + (ip_read >> 3)
    worst = max(np.bincount(phys[0:64, i, m] % 32, minlength=32).max() for i in range(phys.shape[1]) for m in range(phys.shape[2]))
    assert worst >= 8
