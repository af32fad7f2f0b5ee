#!/usr/bin/env python
"""BertSelfOutput / BertOutput sublayer (GEMM + [dropout] + residual + LayerNorm) in its eval form (residual added in the GEMM
epilogue, LayerNorm pass reads half rows) and its training form (GEMM writes half rows, the row pass applies mask + residual)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
m = int(os.environ.get("M", 92160))
for k in (768, 3072):
    n = 768
    a = torch.randn(m, k, generator=g).to(dev).bfloat16()
    w = (torch.randn(n, k, generator=g) * 0.03).to(dev).bfloat16()
    b = (torch.randn(n, generator=g) * 0.1).to(dev)
    res = torch.randn(m, n, generator=g).to(dev).bfloat16()
    gam, bet = torch.ones(n, device=dev), torch.zeros(n, device=dev)
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    ws = torch.empty((m, n), dtype=torch.float32, device=dev)
    for rnd in range(2):
        t0 = timeit(lambda: mh.linear_residual_ln(a, w, b, res, gam, bet, 1e-12, workspace=ws, out=out), iters=30)
        t1 = timeit(lambda: mh.linear_dropout_residual_ln(a, w, b, res, gam, bet, 1e-12, p=0.3, seed=1, offset=0, out=out), iters=30)
        print("M=%d K=%d: eval form %.1f us, training form %.1f us" % (m, k, t0 * 1e6, t1 * 1e6), flush=True)
