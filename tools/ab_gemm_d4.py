#!/usr/bin/env python
"""A/B of the persistent 256 x 256 GEMM against the dual-workgroup 256 x 128 kernel (linear_bf16_d4_kernel: MODCR_GEMM_D4, tuning library):
bit-equality of the two outputs, agreement with a torch fp32 product, interleaved timing rounds in ONE process, and a
cache-flushed repeat loop (every launch compared) for the sporadic-race screen.
usage: ab_gemm_spec.py [SHAPES=92160x3072x768,...] [ROUNDS=5] [STRESS=100]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

mh.use_tuning_library(True)
shapes = os.environ.get("SHAPES", "4608x768x768,92160x3072x768,51712x3072x768,92160x768x3072,46080x2304x768")
rounds = int(os.environ.get("ROUNDS", 5))
stress = int(os.environ.get("STRESS", 100))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def run(spec, a, w, b, act, out):
    os.environ["MODCR_GEMM_D4"] = str(spec)
    return mh.linear(a, w, b, act=act, out=out)


bad = 0
for sh in shapes.split(","):
    m, n, k = (int(v) for v in sh.split("x"))
    a = torch.randn(m, k, generator=g).to(dev).bfloat16()
    w = (torch.randn(n, k, generator=g) * 0.03).to(dev).bfloat16()
    b = torch.randn(n, generator=g).to(dev)
    for act in (1, 0):
        o0 = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        o1 = torch.full((m, n), 7.0, device=dev, dtype=torch.bfloat16)
        run(0, a, w, b, act, o0)
        run(1, a, w, b, act, o1)
        torch.cuda.synchronize()
        same = torch.equal(o0, o1)
        rows = torch.randint(0, m, (2048,), device=dev)
        ref = a[rows].float() @ w.float().t() + b
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        err = (o1[rows].float() - ref).abs().max().item()
        t = {0: [], 1: []}
        for _ in range(rounds):
            for spec in (0, 1):
                t[spec].append(timeit(lambda: run(spec, a, w, b, act, o1 if spec else o0), iters=10, warm=2) * 1e6)
        med = {s: sorted(v)[len(v) // 2] for s, v in t.items()}
        fl = 2.0 * m * n * k
        print("M=%d N=%d K=%d act=%d: bit-equal %s  max|err| vs fp32 %.4f   p8 %.1f us (min %.1f)  d4 %.1f us (min %.1f)  "
              "%.3f -> %.3f of 2.5 PF" % (m, n, k, act, same, err, med[0], min(t[0]), med[1], min(t[1]),
                                           fl / med[0] / 2.5e9, fl / med[1] / 2.5e9), flush=True)
        bad += (not same)
    if stress:
        o0 = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        run(0, a, w, b, 1, o0)
        nbad = 0
        for it in range(stress):
            flush.fill_(it & 255)
            o1 = torch.full((m, n), 3.0, device=dev, dtype=torch.bfloat16)
            run(1, a, w, b, 1, o1)
            if not torch.equal(o0, o1):
                nbad += 1
        print("   stress: %d cache-flushed launches, %d differ from the plain kernel" % (stress, nbad), flush=True)
        bad += nbad
print("AB_GEMM_D4", "FAIL" if bad else "OK")
sys.exit(1 if bad else 0)
