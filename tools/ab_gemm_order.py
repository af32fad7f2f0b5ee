#!/usr/bin/env python
"""A/B of MODCR_GEMM_ORDER variants of the persistent GEMMs (tuning library; the knob is re-read per call) in ONE process,
interleaved rounds, outputs compared bit for bit with the default.  Round 6: bit 8 (256) = s_setprio 1 on waves 4-7 for the whole
kernel, bit 12 (4096) = on waves 0-3 (MI355X_MICROARCH.md "Two waves per SIMD" item 4).
VARIANTS="name:KNOB=v;KNOB=v,name2:..." times arbitrary knob sets instead (e.g. VARIANTS="pf1:MODCR_GEMM_PF=1,pf0:MODCR_GEMM_PF=0": the
epilogue's L2 warm-up for the next tile's activation rows on / off).
usage: ab_gemm_order.py [ORDERS=0,256,4096 | VARIANTS=...] [CASES=MxNxK:act:out,...] [ROUNDS=7]   (act 0 none / 1 gelu; out bf16 | f16)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402

mh.use_tuning_library(True)
if os.environ.get("VARIANTS"):
    orders = []
    VENV = {}
    for item in os.environ["VARIANTS"].split(","):
        name, _, kv = item.partition(":")
        orders.append(name)
        VENV[name] = dict(e.split("=") for e in kv.split(";") if e)
else:
    orders = [int(v) for v in os.environ.get("ORDERS", "0,256,4096").split(",")]
    VENV = {o: {"MODCR_GEMM_ORDER": str(o)} for o in orders}
ALLK = sorted({k for d in VENV.values() for k in d})
cases = os.environ.get("CASES", "92160x3072x768:1:bf16,92160x768x3072:0:f16,92160x768x768:0:f16,51712x3072x768:1:bf16,51712x768x3072:0:f16")
rounds = int(os.environ.get("ROUNDS", 7))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
for case in cases.split(","):
    shape, act, outn = case.split(":")
    m, n, k = (int(v) for v in shape.split("x"))
    act = int(act)
    od, tdt = (mh.BF16, torch.bfloat16) if outn == "bf16" else (mh.F16, torch.float16)
    a = torch.randn(m, k, generator=g).to(dev).bfloat16()
    w = (torch.randn(n, k, generator=g) * 0.03).to(dev).bfloat16()
    b = torch.randn(n, generator=g).to(dev)
    outs = {o: torch.empty(m, n, device=dev, dtype=tdt) for o in orders}
    t = {o: [] for o in orders}
    for rnd in range(rounds):
        for o in orders:
            for knob in ALLK:
                os.environ.pop(knob, None)
            os.environ.update(VENV[o])
            for _ in range(2):
                mh.linear(a, w, b, act=act, out_dtype=od, out=outs[o])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                mh.linear(a, w, b, act=act, out_dtype=od, out=outs[o])
            e1.record()
            torch.cuda.synchronize()
            t[o].append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * m * n * k
    base = outs[orders[0]]
    line = "M=%d N=%d K=%d act=%d out=%s:" % (m, n, k, act, outn)
    for o in orders:
        v = sorted(t[o])
        line += "  %s: median %.1f us min %.1f (%.3f of 2.5 PF)%s" % (o, v[len(v) // 2], v[0], fl / (v[len(v) // 2] * 1e-6) / 2.5e15,
                                                                          "" if o == orders[0] else (" bit-equal" if torch.equal(outs[o], base) else " DIFFERS"))
    print(line, flush=True)
for knob in ALLK:
    os.environ.pop(knob, None)
