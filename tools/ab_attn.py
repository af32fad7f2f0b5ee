#!/usr/bin/env python
"""A/B timing of attention-kernel variants in ONE process on one device (interleaved rounds; runs against
libmodcr_hip_tuning.so, the only build in which the MODCR_* knobs exist -- they are re-read per call there):  python tools/ab_attn.py "NAME=ENV1=v,ENV2=v" ...   (empty = default)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402

mh.use_tuning_library(True)

KNOBS = ("MODCR_ATTN_NOPERSIST", "MODCR_ATTN_NO_V4", "MODCR_ATTN_DEBUG", "MODCR_ATTN_HCONC", "MODCR_ATTN_NO_V4L")
variants = []
for a in sys.argv[1:] or ["default="]:
    name, _, envs = a.partition("=")
    variants.append((name, dict(e.split("=") for e in envs.split(",") if e)))
n, s = int(os.environ.get("N", 256)), int(os.environ.get("S", 180))
h = int(os.environ.get("H", 768))
a = h // 64
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
wqkv = (torch.randn(3 * h, h, generator=g) * 0.05).to(dev).bfloat16()
bqkv = torch.randn(3 * h, generator=g).to(dev)
mask = torch.ones(n, s, device=dev)
fl = n * (6 * s * h * h + 4 * s * s * h)
kw = dict(key_mask=mask)
if os.environ.get("AB_CALL") == "phase3":        # seq_enc layers 9-11: dense mask bits + chunk-mean queries + align map
    t = 80
    dense = (torch.rand(n, s, s, generator=g) < 0.7).float()
    cid = torch.full((n, t), -1, dtype=torch.int32)
    cid[:, 1:70] = (torch.arange(69) // 2).to(torch.int32)
    kw = dict(mask_bits=mh.pack_mask_bits(dense.to(dev)), chunk_id=cid.to(dev), align_map=torch.zeros(n, t, s - t, device=dev), align_t=t)
elif os.environ.get("AB_CALL") == "bits":
    dense = (torch.rand(n, s, s, generator=g) < 0.7).float()
    kw = dict(mask_bits=mh.pack_mask_bits(dense.to(dev)))
if float(os.environ.get("ATTN_DROPOUT", 0)) > 0:     # training-mode variants (the attention-probability dropout of the step)
    kw["attn_dropout"] = (float(os.environ["ATTN_DROPOUT"]), 7, 11)
res = {v[0]: [] for v in variants}
for rnd in range(int(os.environ.get("ROUNDS", 7))):
    for name, env in variants:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update({k: v for k, v in env.items() if k != "SIDE_POST"})
        # SIDE_POST=1 (a pseudo-knob of this tool, not of the library): the call asks for the align map AFTER the dropout
        # (MODCR_ATTN_SIDE_POST_DROPOUT, the reference's semantics and the model's default), SIDE_POST=0 / absent for the un-dropped map
        kw["side_post_dropout"] = env.get("SIDE_POST") == "1" and "attn_dropout" in kw and "align_map" in kw
        for _ in range(2):
            mh.qkv_attn(x, wqkv, bqkv, num_heads=a, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            mh.qkv_attn(x, wqkv, bqkv, num_heads=a, **kw)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 10 * 1e3)
for name, _ in variants:
    v = sorted(res[name])
    print("%-14s median %.1f us  min %.1f us  (%.1f%% of 2.5 PF at the median)" % (name, v[len(v) // 2], v[0], fl / (v[len(v) // 2] * 1e-6) / 2.5e15 * 100))
