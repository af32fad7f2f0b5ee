#!/usr/bin/env python
"""Offline statistics of the attention-probability dropout hash (csrc/attn_common.h): keep fraction, pairwise independence of the four
15-bit fields of a key group and of neighbouring key groups / queries / heads / sequences, field histograms -- the round-3 form
(two 32-bit multiplies per group) against the round-4 form (two full-rate 24-bit multiply-adds, v_mad_u32_u24).
usage: hash_stats.py [decisions per seed, default 8e6]"""
import sys

import numpy as np

M32 = np.uint64(0xFFFFFFFF)
WEYL = np.uint64(0x9E3779B1)


def words_r3(cm, s0, s1):
    x = (cm ^ s0) & M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x85EBCA6B)) & M32; x ^= x >> np.uint64(13)
    y = (x * np.uint64(0xC2B2AE35) + s1) & M32
    y ^= y >> np.uint64(16)
    return x, y


def words_r4(cm, s0, s1):
    t = (cm ^ s0) & M32
    t ^= t >> np.uint64(15)
    x = ((t & np.uint64(0xFFFFFF)) * np.uint64(0xEBCA6B) + (t >> np.uint64(24))) & M32
    x ^= x >> np.uint64(13)
    y = ((x & np.uint64(0xFFFFFF)) * np.uint64(0xB2AE35) + s1) & M32
    y ^= y >> np.uint64(16)
    return x, y


def fields(x, y):
    m = np.uint64(0x7FFF)
    return [(x & m), ((x >> np.uint64(16)) & m), (y & m), ((y >> np.uint64(16)) & m)]


def report(name, words, n_dec, seeds=(1, 2, 3, 4, 5), lp=192, a=12):
    worst = {}
    for seed in seeds:
        rs = np.random.RandomState(seed)
        s0, s1 = np.uint64(rs.randint(0, 2 ** 32, dtype=np.uint64)), np.uint64(rs.randint(0, 2 ** 32, dtype=np.uint64))
        ngrp = n_dec // 4
        # counters as the kernels walk them: ((n A + head) LP + query) (LP / 4) + key group
        ctr = np.arange(ngrp, dtype=np.uint64)
        cm = (ctr * WEYL) & M32
        f = fields(*words(cm, s0, s1))
        for p in (0.1, 0.3):
            thr = np.uint64(round(p * 32768))
            keep = [fi >= thr for fi in f]
            kf = np.mean([k.mean() for k in keep])
            worst.setdefault(("keep fraction - (1-p), p=%.1f" % p), []).append(abs(kf - (1 - p)))
            for i in range(4):
                for j in range(i + 1, 4):
                    worst.setdefault("field pair, p=%.1f" % p, []).append(abs((keep[i] & keep[j]).mean() - (1 - p) ** 2))
            for nm, sh in (("next key group", 1), ("next query", lp // 4), ("next head", lp * lp // 4), ("next sequence", a * lp * lp // 4)):
                for i in range(4):
                    worst.setdefault(nm + ", p=%.1f" % p, []).append(abs((keep[i][:-sh] & keep[i][sh:]).mean() - (1 - p) ** 2))
        # histogram of the top 6 bits of every field
        for fi in f:
            h = np.bincount((fi >> np.uint64(9)).astype(np.int64), minlength=64).astype(np.float64)
            e = h.sum() / 64
            worst.setdefault("chi2 / dof of the top-6-bit histogram", []).append(((h - e) ** 2 / e).sum() / 63)
    sd = (0.9 * 0.1 / (n_dec / 4)) ** 0.5
    print("%s  (%d decisions x %d seeds; one standard error of a pair frequency ~ %.1e)" % (name, n_dec, len(seeds), sd * 2))
    for k in sorted(worst):
        print("   %-48s worst %.3e  mean %.3e" % (k, max(worst[k]), float(np.mean(worst[k]))))


if __name__ == "__main__":
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 8000000
    report("round-3 hash (two v_mul_lo_u32)", words_r3, n)
    report("round-4 hash (two v_mad_u32_u24)", words_r4, n)


def words_r4a(cm, s0, s1):         # first multiply 32-bit, second 24-bit
    x = (cm ^ s0) & M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x85EBCA6B)) & M32; x ^= x >> np.uint64(13)
    y = ((x & np.uint64(0xFFFFFF)) * np.uint64(0xB2AE35) + s1) & M32
    y ^= y >> np.uint64(16)
    return x, y


def words_r4b(cm, s0, s1):         # both 24-bit, the second fed by the HIGH 24 bits of x
    t = (cm ^ s0) & M32
    t ^= t >> np.uint64(15)
    x = ((t & np.uint64(0xFFFFFF)) * np.uint64(0xEBCA6B) + (t >> np.uint64(24))) & M32
    x ^= x >> np.uint64(13)
    y = (((x >> np.uint64(8)) & np.uint64(0xFFFFFF)) * np.uint64(0xB2AE35) + (x ^ s1)) & M32
    y ^= y >> np.uint64(16)
    return x, y


def words_r4c(cm, s0, s1):         # both 24-bit, x finalised by two xorshifts
    t = (cm ^ s0) & M32
    t ^= t >> np.uint64(15)
    x = ((t & np.uint64(0xFFFFFF)) * np.uint64(0xEBCA6B) + (t >> np.uint64(24))) & M32
    x ^= x >> np.uint64(13)
    x ^= (x << np.uint64(9)) & M32
    y = ((x & np.uint64(0xFFFFFF)) * np.uint64(0xB2AE35) + s1) & M32
    y ^= y >> np.uint64(16)
    return x, y


if __name__ == "__main__" and len(sys.argv) > 2:
    report("r4a: 32-bit then 24-bit", words_r4a, n)
    report("r4b: 24-bit twice, second from x >> 8", words_r4b, n)
    report("r4c: 24-bit twice + xorshift-left", words_r4c, n)
