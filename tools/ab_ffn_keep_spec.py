#!/usr/bin/env python
"""modcr_ffn_up_gelu_keep_fwd with / without the seamless-ring kernel (MODCR_GEMM_SPEC, tuning library): both outputs bit-equal,
interleaved timing, cache-flushed repeat loop."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

mh.use_tuning_library(True)
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
bad = 0
for m, h in ((92160, 768), (54272, 1024)):
    x = torch.randn(m, h, generator=g).to(dev).bfloat16()
    w1 = (torch.randn(4 * h, h, generator=g) * 0.03).to(dev).bfloat16()
    b1 = torch.randn(4 * h, generator=g).to(dev)
    outs = {}
    for spec in (0, 1):
        os.environ["MODCR_GEMM_SPEC"] = str(spec)
        outs[spec] = mh.ffn_up_gelu_keep(x, w1, b1)
    torch.cuda.synchronize()
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    t = {0: [], 1: []}
    for _ in range(5):
        for spec in (0, 1):
            os.environ["MODCR_GEMM_SPEC"] = str(spec)
            t[spec].append(timeit(lambda: mh.ffn_up_gelu_keep(x, w1, b1), iters=10, warm=2) * 1e6)
    nb = 0
    os.environ["MODCR_GEMM_SPEC"] = "1"
    for it in range(int(os.environ.get("STRESS", 60))):
        flush.fill_(it & 255)
        o = mh.ffn_up_gelu_keep(x, w1, b1)
        nb += not (torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]))
    print("M=%d H=%d: bit-equal %s  plain %.1f us  seamless %.1f us   stress: %d launches differ" % (m, h, same, sorted(t[0])[2], sorted(t[1])[2], nb), flush=True)
    bad += (not same) + nb
print("AB_FFN_KEEP_SPEC", "FAIL" if bad else "OK")
sys.exit(1 if bad else 0)
