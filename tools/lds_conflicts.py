#!/usr/bin/env python3
"""lds_conflicts.py -- on paper: how many lanes of a wavefront meet in one LDS bank in each pass of lfft_dev.h's plans
(zen_amd/csrc/lfft_dev.h LPlan / LPass), for a given padding period.  64 banks of 4 bytes; a lane's 8-byte access (one complex
value) touches two neighbouring banks; 64 lanes x 8 bytes are two full sweeps of the banks, so 2 lanes per bank is the best a
wavefront instruction can do.  The counters say the 4-value plan's conflict cycles are 1.1-1.5x its LDS-active cycles at nfft
1024 / 2048 (profiles/r05_pmc_latency_kernels.json); this is the table to choose a padding from before measuring it.

  python tools/lds_conflicts.py            # nfft 512..4096, 4 and 8 values per thread, padding per 4 / 8 / 16 / 32 / none
"""
import sys


def plan(log2n, log2v):
    p = (log2n + log2v - 1) // log2v
    base, rem = divmod(log2n, p)
    r = [base + (1 if i < rem else 0) for i in range(p)]
    s = [sum(r[:i]) for i in range(p)]
    return p, r, s


def worst(addrs, pad_shift):
    """addrs: element index per lane (64 lanes); returns the largest number of lanes that touch one bank"""
    hits = [0] * 64
    for a in addrs:
        slot = a + (a >> pad_shift) if pad_shift else a
        b = (2 * slot) % 64                                    # 8-byte slots: two banks each
        hits[b] += 1
        hits[(b + 1) % 64] += 1
    return max(hits)


def analyse(log2n, log2v, pad_shift):
    n, v = 1 << log2n, 1 << log2v
    tf = n // v
    p, r, s = plan(log2n, log2v)
    rows = []
    for ps in range(p):
        R = 1 << r[ps]
        nb = v // R
        log2j = log2n - s[ps] - r[ps]
        J = 1 << log2j
        rd = wr = 0
        for wave in range(max(tf // 64, 1)):
            lanes = [wave * 64 + l for l in range(min(64, tf))]
            for i in range(nb):
                for m in range(R):
                    if ps > 0:                                 # load(): img[pad((k * R + m) * J + j)]
                        rd = max(rd, worst([(((t + i * tf) >> log2j) * R + m) * J + ((t + i * tf) & (J - 1)) for t in lanes], pad_shift))
                    if ps < p - 1:                             # compute(): img[pad(b + c * (N / R))]
                        wr = max(wr, worst([t + i * tf + m * (n // R) for t in lanes], pad_shift))
        rows.append((ps, R, J, rd, wr))
    return rows


def main():
    pads = [2, 3, 4, 5, 0]
    print("lanes per bank, worst wavefront instruction of each pass (reads r / writes w; 2 is the best possible)")
    for log2n in (9, 10, 11, 12):
        for log2v in (2, 3):
            if (1 << log2n) // (1 << log2v) < 64:
                continue
            print("nfft %d, %d values per thread (%d threads):" % (1 << log2n, 1 << log2v, (1 << log2n) >> log2v))
            for pad in pads:
                rows = analyse(log2n, log2v, pad)
                txt = "  ".join("p%d(R%d,J%d) r%d w%d" % row for row in rows)
                tot = sum(max(row[3], 2) + max(row[4], 2) for row in rows)
                print("   pad %-5s sum %3d | %s" % (("per %d" % (1 << pad)) if pad else "none", tot, txt))


if __name__ == "__main__":
    sys.exit(main())
