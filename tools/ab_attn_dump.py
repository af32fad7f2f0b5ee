#!/usr/bin/env python
"""The trainable layers' attention forward (row statistics + Q|K|V image dump) with the dump as a block in front of phase B
(MODCR_ATTN_DUMP_BLOCK=1, tuning library) and interleaved between its key tiles (default): outputs bit-equal, interleaved timing."""
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
bad = 0
for n, s, h in ((512, 180, 768), (512, 106, 1024), (512, 101, 768)):
    a = h // 64
    x = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
    wqkv = (torch.randn(3 * h, h, generator=g) * 0.05).to(dev).bfloat16()
    bqkv = torch.randn(3 * h, generator=g).to(dev)
    mask = torch.ones(n, s, device=dev)
    mask[1, s - 17:] = 0

    def run(block, pd=0.1, plain=False):
        if block:
            os.environ["MODCR_ATTN_DUMP_BLOCK"] = "1"
        else:
            os.environ.pop("MODCR_ATTN_DUMP_BLOCK", None)
        lse = None if plain else torch.empty(n, a, s, device=dev)
        dump = None if plain else torch.zeros(mh.qkv_dump_numel(n, s, a), device=dev, dtype=torch.bfloat16)
        ctx, _ = mh.qkv_attn(x, wqkv, bqkv, key_mask=mask, num_heads=a, attn_dropout=(pd, 7, 11) if pd > 0 else None, lse=lse, dump=dump)
        return ctx, lse, dump
    r0, r1 = run(True), run(False)
    torch.cuda.synchronize()
    same = all(torch.equal(u, v) for u, v in zip(r0, r1))
    t = {0: [], 1: [], 2: []}
    for _ in range(5):
        t[0].append(timeit(lambda: run(True), iters=10, warm=2) * 1e6)
        t[1].append(timeit(lambda: run(False), iters=10, warm=2) * 1e6)
        t[2].append(timeit(lambda: run(False, plain=True), iters=10, warm=2) * 1e6)
    fl = n * (6.0 * s * h * h + 4.0 * s * s * h)
    print("N=%d S=%d H=%d: bit-equal %s   dump as a block %.1f us   interleaved %.1f us (%.3f of 2.5 PF)   no dump / lse %.1f us"
          % (n, s, h, same, sorted(t[0])[2], sorted(t[1])[2], fl / sorted(t[1])[2] / 2.5e9, sorted(t[2])[2]), flush=True)
    bad += not same
print("AB_ATTN_DUMP", "FAIL" if bad else "OK")
sys.exit(1 if bad else 0)
