#!/usr/bin/env python
"""A/B of the seamless-ring GEMM's L2 warm-up (MODCR_GEMM_PF, tuning library) on the CHAIN a layer runs -- LayerNorm pass (writes the
FFN-up operand, 141 MB, as in the step: it is in the Infinity Cache when FFN-up starts) -> FFN-up + GELU -> FFN-down -- so that the
operand's cache state is the step's, not that of a back-to-back micro-benchmark whose own 566 MB of output evicts it.
usage: ab_ffn_chain_pf.py [M=92160] [ROUNDS=7]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402

mh.use_tuning_library(True)
m, h = int(os.environ.get("M", 92160)), 768
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
pre = torch.randn(m, h, generator=g).to(dev).half()
res = torch.randn(m, h, generator=g).to(dev).bfloat16()
gam, bet = torch.ones(h, device=dev), torch.zeros(h, device=dev)
w1 = (torch.randn(4 * h, h, generator=g) * 0.03).to(dev).bfloat16()
b1 = torch.randn(4 * h, generator=g).to(dev)
w2 = (torch.randn(h, 4 * h, generator=g) * 0.03).to(dev).bfloat16()
b2 = torch.randn(h, generator=g).to(dev)
x = torch.empty(m, h, device=dev, dtype=torch.bfloat16)
inter = torch.empty(m, 4 * h, device=dev, dtype=torch.bfloat16)
out = torch.empty(m, h, device=dev, dtype=torch.float16)


# VARIANTS="name:KNOB=v;KNOB=v,..." (default: the warm-up off / on): knob sets applied to the FFN-up launch only
VARS = [("PF=0", {"MODCR_GEMM_PF": "0"}), ("PF=1", {"MODCR_GEMM_PF": "1"})]
if os.environ.get("VARIANTS"):
    VARS = []
    for item in os.environ["VARIANTS"].split(","):
        name, _, kv = item.partition(":")
        VARS.append((name, dict(e.split("=") for e in kv.split(";") if e)))
KNOBS = sorted({k[5:] if k.startswith("DOWN_") else k for _, d in VARS for k in d})      # "DOWN_<KNOB>": applied to the FFN-down launch instead


def chain(pf, only_up=False):
    for kn in KNOBS:
        os.environ.pop(kn, None)
    mh.layernorm(pre, gam, bet, 1e-12, residual=res, out_dtype=mh.BF16, out=x)
    os.environ.update({k: v for k, v in VARS[pf][1].items() if not k.startswith("DOWN_")})
    mh.linear(x, w1, b1, act=1, out=inter)
    for kn in KNOBS:
        os.environ.pop(kn, None)
    if not only_up:
        os.environ.update({k[5:]: v for k, v in VARS[pf][1].items() if k.startswith("DOWN_")})
        mh.linear(inter, w2, b2, out=out, out_dtype=mh.F16)
        for kn in KNOBS:
            os.environ.pop(kn, None)


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


t = {(pf, ou): [] for pf in range(len(VARS)) for ou in (True, False)}
for _ in range(int(os.environ.get("ROUNDS", 7))):
    for pf in range(len(VARS)):
        for ou in (True, False):
            t[(pf, ou)].append(timeit(lambda: chain(pf, ou)))
for ou in (True, False):
    line = "M=%d  %s:" % (m, "LN pass + FFN-up" if ou else "LN pass + FFN-up + FFN-down")
    for pf in range(len(VARS)):
        v = sorted(t[(pf, ou)])
        line += "   %s median %.1f us (min %.1f)" % (VARS[pf][0], v[len(v) // 2], v[0])
    print(line)
