#!/usr/bin/env python
"""Time modcr_linear_fwd (bf16) on a list of M,N,K shapes: SHAPES="4096x4096x4096,46080x3072x768"."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

if os.environ.get("TUNING"):
    mh.use_tuning_library(True)      # knobs (MODCR_GEMM_ORDER ...) are read by the tuning build only
shapes = os.environ.get("SHAPES", "4096x4096x4096,8192x8192x8192,46080x3072x768,46080x768x3072,46080x768x768")
act = int(os.environ.get("ACT", 0))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
for sh in shapes.split(","):
    m, n, k = (int(v) for v in sh.split("x"))
    a = torch.randn(m, k, generator=g).to(dev).bfloat16()                 # LayerNorm-like rows, BERT-like weights
    w = (torch.randn(n, k, generator=g) * 0.03).to(dev).bfloat16()
    b = torch.randn(n, generator=g).to(dev)
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: mh.linear(a, w, b, act=act, out=out), iters=int(os.environ.get("ITERS", 20)))
    print("linear M=%d N=%d K=%d act=%d: %.1f us  %.1f TFLOP/s" % (m, n, k, act, t * 1e6, 2.0 * m * n * k / t / 1e12))
