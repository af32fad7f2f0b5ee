#!/usr/bin/env python
"""Micro-benchmarks of the individual C-ABI kernels at BASELINE config 2 shapes (GPU box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    n, s, h, a = int(os.environ.get("N", 256)), int(os.environ.get("S", 180)), 768, 12
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
    wqkv = (torch.randn(3 * h, h, generator=g) * 0.05).to(dev).bfloat16()
    bqkv = torch.randn(3 * h, generator=g).to(dev)
    mask = torch.ones(n, s, device=dev)
    t = timeit(lambda: mh.qkv_attn(x, wqkv, bqkv, key_mask=mask, num_heads=a))
    fl = n * (6 * s * h * h + 4 * s * s * h)
    print("qkv_attn_fwd  N=%d S=%d: %.1f us  %.1f TFLOP/s (%.1f%% of 2.5 PF)" % (n, s, t * 1e6, fl / t / 1e12, fl / t / 2.5e15 * 100))
    m = n * s
    for (nn, kk, act, name) in ((h, h, 0, "proj"), (4 * h, h, 1, "ffn_up+gelu"), (h, 4 * h, 0, "ffn_down")):
        a_ = torch.randn(m, kk, generator=g).to(dev).bfloat16()
        w = (torch.randn(nn, kk, generator=g) * 0.05).to(dev).bfloat16()
        b = torch.randn(nn, generator=g).to(dev)
        out = torch.empty(m, nn, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: mh.linear(a_, w, b, act=act, out=out))
        fl = 2.0 * m * nn * kk
        print("linear %-12s M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s" % (name, m, nn, kk, t * 1e6, fl / t / 1e12))
        out32 = torch.empty(m, nn, device=dev, dtype=torch.float32)
        res = torch.randn(m, nn, generator=g).to(dev).bfloat16()
        t = timeit(lambda: mh.linear(a_, w, b, act=act, residual=res, out_dtype=mh.F32, out=out32))
        print("   + residual, fp32 out: %.1f us  %.1f TFLOP/s" % (t * 1e6, fl / t / 1e12))
    pre = torch.randn(m, h, generator=g).to(dev)
    gam, bet = torch.ones(h, device=dev), torch.zeros(h, device=dev)
    y = torch.empty(m, h, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: mh.layernorm(pre, gam, bet, 1e-12, out_dtype=mh.BF16, out=y))
    print("layernorm f32->bf16 M=%d: %.1f us  %.2f TB/s" % (m, t * 1e6, m * h * 6 / t / 1e12))


if __name__ == "__main__":
    main()
