#!/usr/bin/env python
"""Run only the weight-gradient product dW = dY^T X of an encoder layer (FFN-up shape by default: dY [M, 3072], X [M, 768], M = 92160;
the half-TN form of linear_bf16_p8_kernel with its operand transpose, split-K reduce and bias column sums) for rocprofv3 passes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402

m, n, k = int(os.environ.get("M", 92160)), int(os.environ.get("NN", 3072)), int(os.environ.get("K", 768))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
dy = torch.randn(m, n, generator=g).to(dev).bfloat16()
x = torch.randn(m, k, generator=g).to(dev).bfloat16()
dw, db = torch.empty(n, k, device=dev), torch.empty(n, device=dev)
for _ in range(int(os.environ.get("REPS", 4))):
    mh.linear_bwd_weight(dy, x, dw, db, mfma=True)
torch.cuda.synchronize()
