#!/usr/bin/env python
"""A/B of an epilogue knob of the FFN-up GEMM (tuning library, MODCR_GEMM_ORDER bits) measured on the PAIR FFN-up -> FFN-down, so a
store policy that helps the producer and hurts the consumer's reads shows as the sum.  ORDERS="0,2048"."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

mh.use_tuning_library(True)
m, h = int(os.environ.get("M", 92160)), 768
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(m, h, generator=g).to(dev).bfloat16()
w1 = (torch.randn(4 * h, h, generator=g) * 0.03).to(dev).bfloat16()
b1 = torch.randn(4 * h, generator=g).to(dev)
w2 = (torch.randn(h, 4 * h, generator=g) * 0.03).to(dev).bfloat16()
b2 = torch.randn(h, generator=g).to(dev)
inter = torch.empty(m, 4 * h, device=dev, dtype=torch.bfloat16)
out = torch.empty(m, h, device=dev, dtype=torch.float16)
orders = [int(v) for v in os.environ.get("ORDERS", "0,2048").split(",")]
res = {o: ([], []) for o in orders}


def up(o):
    os.environ["MODCR_GEMM_ORDER"] = str(o)
    mh.linear(x, w1, b1, act=1, out=inter)


def pair(o):
    up(o)
    os.environ["MODCR_GEMM_ORDER"] = "0"
    mh.linear(inter, w2, b2, out=out, out_dtype=mh.F16)


for _ in range(int(os.environ.get("ROUNDS", 5))):
    for o in orders:
        res[o][0].append(timeit(lambda: up(o), iters=10, warm=2) * 1e6)
        res[o][1].append(timeit(lambda: pair(o), iters=10, warm=2) * 1e6)
for o in orders:
    a, b = sorted(res[o][0]), sorted(res[o][1])
    print("order %5d: FFN-up alone median %.1f us (min %.1f)   FFN-up + FFN-down median %.1f us (min %.1f)" % (o, a[len(a) // 2], a[0], b[len(b) // 2], b[0]))
