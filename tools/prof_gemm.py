#!/usr/bin/env python
"""Run only the bf16 GEMM (FFN-down shape by default) for rocprofv3 --pmc passes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402

if os.environ.get("TUNING"):
    mh.use_tuning_library(True)
m, n, k = int(os.environ.get("M", 46080)), int(os.environ.get("NN", 768)), int(os.environ.get("K", 3072))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
a = torch.randn(m, k, generator=g).to(dev).bfloat16()
w = (torch.randn(n, k, generator=g) * 0.05).to(dev).bfloat16()
b = torch.randn(n, generator=g).to(dev)
out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
act = int(os.environ.get("ACT", 0))                 # 1 = GELU epilogue (FFN-up)
if int(os.environ.get("F16", 0)):                   # IEEE-half output: the pre-LayerNorm rows of the bf16 path
    out = torch.empty(m, n, device=dev, dtype=torch.float16)
for _ in range(6):
    mh.linear(a, w, b, act=act, out=out, out_dtype=mh.F16 if out.dtype == torch.float16 else None)
torch.cuda.synchronize()
