#!/usr/bin/env python
"""Timing-only ablations of the dual-workgroup GEMM (tuning library; MODCR_GEMM_ORDER bits: 1 = nothing staged inside the K loop,
2 = staging in front of the MFMAs instead of between them) against the 256 x 256 kernel.  usage: abl_gemm_d4.py [SHAPE=92160x3072x768]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
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
cases = [("p8", 0, 0)] + [("d4 order=%s" % o, 1, int(o)) for o in os.environ.get("ORDERS", "0,2,1").split(",")]
for rnd in range(int(os.environ.get("ROUNDS", 3))):
    for name, d4, order in cases:
        os.environ["MODCR_GEMM_D4"] = str(d4)
        os.environ["MODCR_GEMM_ORDER"] = str(order)
        for act in (1, 0):
            t = timeit(lambda: mh.linear(a, w, b, act=act, out=out), iters=10, warm=2) * 1e6
            print("round %d  %-14s act=%d  %.1f us" % (rnd, name, act, t), flush=True)
os.environ["MODCR_GEMM_ORDER"] = "0"
