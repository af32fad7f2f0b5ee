#!/usr/bin/env python
"""The attention backward of seq_enc's layers 9-11 (dense mask + chunk-mean queries + the align map's gradient: attn_bwd6_kernel<3,1,1,1>
and attn_dalign_delta_kernel) at the bench size, next to the same call without the map's gradient; run under
`rocprofv3 --kernel-trace --stats` for the kernel times.  usage: time_attn_bwd_dalign.py [N=512]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

n = int(os.environ.get("N", 512))
t, r, h, a = 80, 100, 768, 12
s = t + r
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
w = (torch.randn(3 * h, h, generator=g) * 0.05).to(dev).bfloat16()
b = torch.randn(3 * h, generator=g).to(dev)
dense = (torch.rand(n, s, s, generator=g) < 0.7).float()
dense[:, torch.arange(s), torch.arange(s)] = 1
bits = mh.pack_mask_bits(dense.to(dev))
cid = torch.full((n, t), -1, dtype=torch.int32)
cid[:, 1:61] = (torch.arange(60) // 2).to(torch.int32)
cid = cid.to(dev)
dctx = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
d_align = torch.randn(n, t, r, generator=g).to(dev)
drop = (0.1, 7, 11)
lse = torch.empty(n, a, s, device=dev)
dump = torch.empty(mh.qkv_dump_numel(n, s, a), device=dev, dtype=torch.bfloat16)
amap = torch.zeros(n, t, r, device=dev)
ctx, _ = mh.qkv_attn(x, w, b, mask_bits=bits, chunk_id=cid, num_heads=a, attn_dropout=drop, align_map=amap, align_t=t, lse=lse, dump=dump)
dw, db = torch.empty(3 * h, h, device=dev), torch.empty(3 * h, device=dev)
for name, da in (("with d_align", d_align), ("without", None)):
    f = lambda: mh.qkv_attn_bwd(dctx, x, w, b, dw, db, mask_bits=bits, chunk_id=cid, num_heads=a, attn_dropout=drop, d_align=da, align_t=t if da is not None else 0,
                                ctx=ctx, lse=lse, dump=dump)
    print("%-14s whole call %.1f us" % (name, timeit(f, iters=10, warm=2) * 1e6), flush=True)
