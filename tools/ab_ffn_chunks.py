#!/usr/bin/env python
"""The pair FFN-up -> FFN-down (product library) run over the M rows in 1, 2, 3, 4, 6 chunks: does keeping a chunk's [rows, 4H]
intermediate inside the 256 MB Infinity Cache between the two products pay for the partial GEMM rounds of smaller launches?
M = 92160 (566 MB intermediate) and M = 51712 (318 MB).  Interleaved rounds, medians."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

h = 768
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
w1 = (torch.randn(4 * h, h, generator=g) * 0.03).to(dev).bfloat16()
b1 = torch.randn(4 * h, generator=g).to(dev)
w2 = (torch.randn(h, 4 * h, generator=g) * 0.03).to(dev).bfloat16()
b2 = torch.randn(h, generator=g).to(dev)
for m in (92160, 51712):
    x = torch.randn(m, h, generator=g).to(dev).bfloat16()
    inter = torch.empty(m, 4 * h, device=dev, dtype=torch.bfloat16)
    out = torch.empty(m, h, device=dev, dtype=torch.float16)

    def pair(nc):
        step = ((m + nc - 1) // nc + 255) // 256 * 256
        for r0 in range(0, m, step):
            r1 = min(m, r0 + step)
            mh.linear(x[r0:r1], w1, b1, act=1, out=inter[r0:r1])
            mh.linear(inter[r0:r1], w2, b2, out=out[r0:r1], out_dtype=mh.F16)

    res = {}
    for _ in range(5):
        for nc in (1, 2, 3, 4, 6):
            res.setdefault(nc, []).append(timeit(lambda: pair(nc), iters=10, warm=2) * 1e6)
    print("M=%d: " % m + "   ".join("%d chunk%s %.1f us" % (nc, "s" if nc > 1 else "", sorted(v)[len(v) // 2]) for nc, v in res.items()), flush=True)
