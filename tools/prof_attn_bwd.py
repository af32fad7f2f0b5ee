#!/usr/bin/env python
"""A few modcr_qkv_attn_bwd calls at the bench size (N = 512, S = 180, H = 768, A = 12, attention dropout 0.1) for rocprofv3;
TUNING=1 loads the tuning library (MODCR_ATTN_BWD_DEBUG ablations of attn_bwd_mfma_kernel: results are wrong, only the clock counts)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402

if os.environ.get("TUNING"):
    mh.use_tuning_library(True)
dev = torch.device("cuda")
n, s, h, a = int(os.environ.get("N", 512)), int(os.environ.get("S", 180)), 768, 12
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
dctx = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
wqkv = (torch.randn(3 * h, h, generator=g) * 0.03).to(dev).bfloat16()
bqkv = torch.zeros(3 * h, device=dev)
km = torch.ones(n, s, device=dev)
dw, db = torch.empty(3 * h, h, device=dev), torch.empty(3 * h, device=dev)
# OLD=1: the older core (row statistics recomputed); default: the five-product core on the forward's ctx + lse
new = not os.environ.get("OLD")
pd = float(os.environ.get("PDROP", 0.1))
drop = (pd, 7, 11) if pd > 0 else None
lse = torch.empty(n, a, s, device=dev) if new else None
dump = torch.empty(mh.qkv_dump_numel(n, s, a), device=dev, dtype=torch.bfloat16) if (new and not os.environ.get("NODUMP")) else None
ctx, _ = mh.qkv_attn(x, wqkv, bqkv, key_mask=km, num_heads=a, attn_dropout=drop, lse=lse, dump=dump)
for _ in range(int(os.environ.get("REPS", 4))):
    mh.qkv_attn_bwd(dctx, x, wqkv, bqkv, dw, db, key_mask=km, num_heads=a, attn_dropout=drop, ctx=ctx if new else None, lse=lse, dump=dump)
torch.cuda.synchronize()
