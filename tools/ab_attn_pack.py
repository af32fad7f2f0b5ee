#!/usr/bin/env python
"""short sequences alone against k of them packed into one attention row block under a block-diagonal mask
(modcr_build_packed_mask): S = 37 (c5's image-only pass, H = 1024) and S = 101 (PMR's, H = 768)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402

dev = torch.device("cuda")


def run(n, s, h, a, ks):
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
    wqkv = (torch.randn(3 * h, h, generator=g) * 0.03).to(dev).bfloat16()
    bqkv = torch.zeros(3 * h, device=dev)
    km = torch.ones(n, s, device=dev)
    km[1, s - 5:] = 0
    drop = (0.1, 7, 11)
    variants = {"alone": lambda: mh.qkv_attn(x, wqkv, bqkv, key_mask=km, num_heads=a, attn_dropout=drop)}
    ref, _ = mh.qkv_attn(x, wqkv, bqkv, key_mask=km, num_heads=a)
    for k in ks:
        bits = mh.build_packed_mask(km, k)
        xv = x.view(n // k, k * s, h)
        got, _ = mh.qkv_attn(xv, wqkv, bqkv, mask_bits=bits, num_heads=a)
        err = float((got.view(n, s, h).float() - ref.float()).abs().max())
        variants["k=%d (%d rows, max|diff| %.3g)" % (k, k * s, err)] = (lambda xv=xv, bits=bits: mh.qkv_attn(xv, wqkv, bqkv, mask_bits=bits, num_heads=a, attn_dropout=drop))
    res = {kk: [] for kk in variants}
    for rnd in range(3):
        for name, fn in variants.items():
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = n * (6.0 * s * h * h + 4.0 * s * s * h)
    for kk, v in res.items():
        print("N=%d S=%d H=%d  %-40s min %.1f us = %.3f of peak" % (n, s, h, kk, min(v), fl / (min(v) * 1e-6) / 2.5e15))


run(128, 37, 1024, 16, [2, 4])
run(512, 101, 768, 12, [2])
run(512, 37, 768, 12, [4])
