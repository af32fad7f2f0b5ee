#!/usr/bin/env python
"""Cycle stamps of workgroup 0 (waves 0 and 4) at the seams of every tile of the fused attention kernel (tuning library,
MODCR_ATTN_TRACE_PTR): 0 tile top, 1 phase-A K loop entered (tables, prologue issue and its wait behind it), 2 K loop done,
3 images written (barrier passed), 4 phase B done, 5 flag barrier passed, 6 context rows stored, 7 end-of-tile barrier passed."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402

mh.use_tuning_library(True)
n, s, h = int(os.environ.get("N", 512)), int(os.environ.get("S", 180)), int(os.environ.get("H", 768))
a = h // 64
drop = float(os.environ.get("DROP", 0.1))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
wqkv = (torch.randn(3 * h, h, generator=g) * 0.05).to(dev).bfloat16()
bqkv = torch.randn(3 * h, generator=g).to(dev)
mask = torch.ones(n, s, device=dev)
buf = torch.zeros(1024, dtype=torch.int64, device=dev)
kw = dict(key_mask=mask, num_heads=a)
if drop > 0:
    kw.update(attn_dropout=(drop, 1234, 0))
for _ in range(5):
    mh.qkv_attn(x, wqkv, bqkv, **kw)
os.environ["MODCR_ATTN_TRACE_PTR"] = str(buf.data_ptr())
mh.qkv_attn(x, wqkv, bqkv, **kw)
torch.cuda.synchronize()
t = buf.cpu().view(2, 64, 8)
names = ("setup+prologue", "phaseA", "images", "phaseB", "flagbar", "ctx", "endbar")
for wv in range(2):
    print("wave %d:  %s | tile" % (4 * wv, "  ".join("%14s" % nm for nm in names)))
    for it in range(32):
        e = t[wv, it]
        if e[0] == 0:
            break
        nxt = t[wv, it + 1, 0] if t[wv, it + 1, 0] else e[7]
        print("         " + "  ".join("%14d" % int(e[k + 1] - e[k]) for k in range(7)) + " | %d" % int(nxt - e[0]))
