#!/usr/bin/env python
"""Timing-only ablations of the persistent GEMM epilogue (tuning library, MODCR_GEMM_ORDER bits: 16 no epilogue, 32 every tile
stores to tile (0,0), 128 no global stores) for the plain and the loader / storer kernels at the FFN-up shape."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

mh.use_tuning_library(True)
m, n, k = (int(v) for v in os.environ.get("SHAPE", "92160x3072x768").split("x"))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
a = torch.randn(m, k, generator=g).to(dev).bfloat16()
w = (torch.randn(n, k, generator=g) * 0.03).to(dev).bfloat16()
b = torch.randn(n, generator=g).to(dev)
out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
cases = [(s, o, act) for act in (1, 0) for s in (0, 1) for o in [int(v) for v in os.environ.get("ORDERS", "0,16,32,128").split(",")]]
res = {c: [] for c in cases}
for _ in range(int(os.environ.get("ROUNDS", 4))):
    for c in cases:
        os.environ["MODCR_GEMM_SPEC"], os.environ["MODCR_GEMM_ORDER"] = str(c[0]), str(c[1])
        res[c].append(timeit(lambda: mh.linear(a, w, b, act=c[2], out=out), iters=10, warm=2) * 1e6)
for c in cases:
    v = sorted(res[c])
    print("act=%d spec=%d order=%3d: median %.1f us  min %.1f" % (c[2], c[0], c[1], v[len(v) // 2], v[0]))
