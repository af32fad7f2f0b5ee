#!/usr/bin/env python
"""fused attention forward at the bench size with and without its training-mode side outputs (lse, Q|K|V dump): torch events
around 20 back-to-back calls each, interleaved rounds in one process"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402

dev = torch.device("cuda")
n, s, h, a = int(os.environ.get("N", 512)), int(os.environ.get("S", 180)), 768, 12
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
wqkv = (torch.randn(3 * h, h, generator=g) * 0.03).to(dev).bfloat16()
bqkv = torch.zeros(3 * h, device=dev)
km = torch.ones(n, s, device=dev)
lse = torch.empty(n, a, s, device=dev)
dump = torch.empty(mh.qkv_dump_numel(n, s, a), device=dev, dtype=torch.bfloat16)
drop = (0.1, 7, 11)
variants = {"plain": {}, "lse": dict(lse=lse), "lse+dump": dict(lse=lse, dump=dump)}
res = {k: [] for k in variants}
for rnd in range(4):
    for name, kw in variants.items():
        for _ in range(3):
            mh.qkv_attn(x, wqkv, bqkv, key_mask=km, num_heads=a, attn_dropout=drop, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            mh.qkv_attn(x, wqkv, bqkv, key_mask=km, num_heads=a, attn_dropout=drop, **kw)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 20 * 1e3)
for k, v in res.items():
    print("%-10s min %.1f us  median %.1f us" % (k, min(v), sorted(v)[len(v) // 2]))
