#!/usr/bin/env python
"""Offline statistics of the attention-probability dropout mask (csrc/attn_common.h, round-5 two-level form):
    base(row, l4) = fold(cm * 0x85EBCA6B + K),  cm = (4 row + l4) * 0x9E3779B1 mod 2^32;   word j = fold(base * C[j] + K)
with fold = low ^ high half of the 64-bit multiply-add.  Checked per seed: keep fraction of every 16-bit field; the pair
frequency of ALL field pairs among the first `nw` words of a row (z-scores: rms ~ 1 and max ~ 3-4 over ~500 pairs is noise);
sampled triples; neighbouring l4 / query / head / sequence rows; the same counters under the next layer's key (offset + 1);
histograms.  Also prints the forms that were tried and FAIL (a single multiply-fold of the Weyl counter used directly as a word;
the raw halves of the product), so the reason for the second level is reproducible.
usage: hash_stats.py [rows per seed, default 2e6] [words, default 16]"""
import itertools
import sys

import numpy as np

u = np.uint64
M32 = u(0xFFFFFFFF)
WEYL = u(0x9E3779B1)
C1 = u(0x85EBCA6B)


def fmix32(x):
    x &= 0xffffffff
    x ^= x >> 16; x = x * 0x85EBCA6B & 0xffffffff; x ^= x >> 13; x = x * 0xC2B2AE35 & 0xffffffff; x ^= x >> 16
    return x


CT = [u(fmix32((j + 1) * 0x9E3779B1) | 1) for j in range(64)]


def fold(a, c, k):
    p = a * c + k                    # uint64 arithmetic wraps mod 2^64
    return (p & M32) ^ (p >> u(32))


def words_two_level(ctr, k, nw):
    base = fold((ctr * WEYL) & M32, C1, k)
    return [fold(base, CT[j], k) for j in range(nw)]


def words_single_fold(ctr, k, nw):            # FAILS: the base used directly, consecutive counters as "words"
    return [fold(((ctr * u(nw) + u(j)) * WEYL) & M32, C1, k) for j in range(nw)]


def popcount_and(a, b, pc=np.array([bin(i).count("1") for i in range(256)], dtype=np.int64)):
    return int(pc[a & b].sum())


def report(name, words, rows, nw, seeds=(1, 2, 3), heads=12):
    for p in (0.1, 0.3):
        thr = int(round(p * 65536)) - 32768
        zs, dev, nb, cross, chi = [], [], [], [], []
        for seed in seeds:
            rs = np.random.RandomState(seed)
            k = u(rs.randint(0, 2 ** 32, dtype=np.uint64)) << u(32) | u(rs.randint(0, 2 ** 32, dtype=np.uint64))
            k2 = u((int(k) + 0x9E3779B97F4A7C15) & (2 ** 64 - 1))
            ctr = np.arange(rows, dtype=np.uint64)
            keep, bits = [], []
            for w in words(ctr, k, nw):
                for f in (w & u(0xFFFF), w >> u(16)):
                    kf = f.astype(np.int64)
                    kf = np.where(kf >= 32768, kf - 65536, kf) >= thr
                    keep.append(kf); bits.append(np.packbits(kf))
                    h = np.bincount((f >> u(10)).astype(np.int64), minlength=64).astype(np.float64)
                    chi.append(((h - h.mean()) ** 2 / h.mean()).sum() / 63)
            q = 1 - p
            sig = (q * q * (1 - q * q) / rows) ** 0.5
            dev += [abs(kf.mean() - q) / (q * p / rows) ** 0.5 for kf in keep]
            for i, j in itertools.combinations(range(len(bits)), 2):
                zs.append((popcount_and(bits[i], bits[j]) / rows - q * q) / sig)
            for sh in (1, 2, 4, 8, 64, 4 * 256, 4 * 256 * heads):      # l4 + 1, l4 + 2, next query, q + 2, q + 16, next head, next sequence
                for i in range(0, len(keep), 3):
                    nb.append(((keep[i][:-sh] & keep[(i + 5) % len(keep)][sh:]).mean() - q * q) / sig)
            for j, w in enumerate(words(ctr, k2, nw)[::3]):
                kf = (w >> u(16)).astype(np.int64)
                kf = np.where(kf >= 32768, kf - 65536, kf) >= thr
                cross.append(((kf & keep[6 * j + 1]).mean() - q * q) / sig)
        zs, nb, cross = np.array(zs), np.array(nb), np.array(cross)
        print("%-28s p=%.1f rows=%d x %d seeds: field keep |z| max %.1f; %d field pairs z rms %.2f max %.1f; neighbours z rms %.2f max %.1f; "
              "next-layer key z rms %.2f max %.1f; chi2/dof top-6-bit histograms max %.2f"
              % (name, p, rows, len(seeds), max(dev), len(zs), (zs ** 2).mean() ** 0.5, abs(zs).max(), (nb ** 2).mean() ** 0.5, abs(nb).max(),
                 (cross ** 2).mean() ** 0.5, abs(cross).max(), max(chi)), flush=True)


if __name__ == "__main__":
    rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2000000
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    report("two-level (shipped)", words_two_level, rows, nw)
    report("single fold, no 2nd level", words_single_fold, rows // 4, nw)
