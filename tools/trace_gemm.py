#!/usr/bin/env python
"""Cycle stamps of workgroup 0 (waves 0 and 4) at the seams of every tile of the persistent GEMM (tuning library,
MODCR_GEMM_TRACE_PTR): 0 tile top, 1 K loop entered, 2 K loop done, 3 next prologue issued, 4 own epilogue arithmetic done,
5 barrier X passed, 6 epilogue done."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402

mh.use_tuning_library(True)
m, n, k = (int(v) for v in os.environ.get("SHAPE", "92160x3072x768").split("x"))
act = int(os.environ.get("ACT", 1))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
a = torch.randn(m, k, generator=g).to(dev).bfloat16()
w = (torch.randn(n, k, generator=g) * 0.03).to(dev).bfloat16()
b = torch.randn(n, generator=g).to(dev)
out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
buf = torch.zeros(2048, dtype=torch.int64, device=dev)
for spec in [int(v) for v in os.environ.get("SPECS", "0,1").split(",")]:
    os.environ["MODCR_GEMM_SPEC"] = str(spec)
    os.environ.pop("MODCR_GEMM_TRACE_PTR", None)
    for _ in range(int(os.environ.get("HEAT", 5))):     # HEAT=4000: ~2 s of back-to-back launches first, so the stamps see the clock the chip holds under load
        mh.linear(a, w, b, act=act, out=out)
    os.environ["MODCR_GEMM_TRACE_PTR"] = str(buf.data_ptr())
    buf.zero_()
    if os.environ.get("CHAIN"):         # CHAIN=1: the operand is written by a LayerNorm pass right before the launch, as in the step
        pre_c = torch.randn(m, k, device=dev).half()
        res_c = torch.randn(m, k, device=dev).bfloat16()
        os.environ.pop("MODCR_GEMM_TRACE_PTR", None)
        for _ in range(3):
            mh.layernorm(pre_c, torch.ones(k, device=dev), torch.zeros(k, device=dev), 1e-12, residual=res_c, out_dtype=mh.BF16, out=a)
            mh.linear(a, w, b, act=act, out=out)
        mh.layernorm(pre_c, torch.ones(k, device=dev), torch.zeros(k, device=dev), 1e-12, residual=res_c, out_dtype=mh.BF16, out=a)
        os.environ["MODCR_GEMM_TRACE_PTR"] = str(buf.data_ptr())
    mh.linear(a, w, b, act=act, out=out)
    torch.cuda.synchronize()
    t = buf.cpu().view(2, 128, 8)
    print("spec=%d act=%d  (cycles; per tile: K loop = 2-1, entry = 1-0, prologue = 3-2, arith = 4-3, X = 5-4, rest = 6-5, tile = next0-0)" % (spec, act))
    for wv in range(2):
        rows = []
        for it in range(17):
            e = t[wv, it]
            if e[0] == 0:
                break
            nxt = t[wv, it + 1, 0] if it + 1 < 128 and t[wv, it + 1, 0] else e[6]
            pro = int(e[3] - e[2]) if e[3] else 0
            base = e[3] if e[3] else e[2]
            rows.append((int(e[1] - e[0]), int(e[2] - e[1]), pro, int(e[4] - base) if e[4] else 0, int(e[5] - e[4]) if e[5] else 0,
                         int(e[6] - (e[5] if e[5] else base)), int(nxt - e[0])))
        nt = len(rows)
        if nt >= 4:     # in-kernel clock: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime) from the 2nd tile's top to the last one's
            cyc, tick = int(t[wv, nt - 1, 0] - t[wv, 1, 0]), int(t[wv, nt - 1, 7] - t[wv, 1, 7])
            if tick > 0:
                print(" wave %d: in-kernel clock over tiles 1..%d = %.3f GHz (%d cycles in %.1f us)" % (4 * wv, nt - 1, cyc / tick * 0.1, cyc, tick * 0.01))
        print(" wave %d:" % (4 * wv))
        for r in rows:
            print("   entry %6d  kloop %6d  prologue %5d  arith %6d  X %6d  rest %6d  | tile %6d" % r)
