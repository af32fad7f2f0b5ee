#!/usr/bin/env python
"""A/B of the early prologue of the fused-attention tile kernels (csrc/attn.hip `early`: the next tile's half-tiles B1 | A0' | B0' are
issued under the current tile's context stores) in ONE process on the tuning library: MODCR_ATTN_DEBUG=64 turns it off.  Interleaved
rounds, medians; context rows compared bit for bit (the two orders stage the same bytes).
HISTORICAL: the early prologue measured 1.1-2.6 % slower (profiles/r05_ab_attn_early_prologue.log) and its code was
taken out again; debug bit 6 now belongs to tools/ab_attn.py's align-map pricing, so this script no longer compares anything."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

mh.use_tuning_library(True)
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
for n, s, h, pd, dense in ((512, 180, 768, 0.1, False), (512, 180, 768, 0.0, False), (256, 180, 768, 0.1, False), (512, 180, 768, 0.1, True),
                           (512, 101, 768, 0.1, False), (512, 106, 1024, 0.1, False), (128, 230, 1024, 0.1, False)):
    a = h // 64
    x = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
    wqkv = (torch.randn(3 * h, h, generator=g) * 0.05).to(dev).bfloat16()
    bqkv = torch.randn(3 * h, generator=g).to(dev)
    mask = torch.ones(n, s, device=dev)
    mask[::3, s - 17:] = 0
    bits = mh.pack_mask_bits((torch.rand(n, s, s, generator=g) < 0.8).float().to(dev)) if dense else None
    drop = (pd, 7, 11) if pd else None
    call = lambda: mh.qkv_attn(x, wqkv, bqkv, key_mask=None if dense else mask, mask_bits=bits, num_heads=a, attn_dropout=drop)[0]
    res, outs = {"early": [], "plain": []}, {}
    for _ in range(7):
        for name in ("early", "plain"):
            if name == "plain":
                os.environ["MODCR_ATTN_DEBUG"] = "64"
            else:
                os.environ.pop("MODCR_ATTN_DEBUG", None)
            res[name].append(timeit(call, iters=10, warm=2) * 1e6)
            outs[name] = call()
    os.environ.pop("MODCR_ATTN_DEBUG", None)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    fl = n * (6.0 * s * h * h + 4.0 * s * s * h)
    print("N=%d S=%d H=%d p=%.1f dense=%s: plain prologue %.1f us (%.4f)  early %.1f us (%.4f of 2.5 PF)  bit-equal: %s"
          % (n, s, h, pd, dense, med["plain"], fl / med["plain"] / 2.5e9, med["early"], fl / med["early"] / 2.5e9,
             torch.equal(outs["early"], outs["plain"])), flush=True)
