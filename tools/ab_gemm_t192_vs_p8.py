#!/usr/bin/env python
"""The 192 x 384 tile kernel (the dispatcher's choice for N = 768 / 1024 at M = 92160 / 54272) against the seamless-ring 256 x 256 kernel forced
on the same shapes (MODCR_GEMM_T192=0, tuning library): round 4, 117 vs 121 us (K = 768) and 341 vs 352 us (K = 3072) at M = 92160 -- the
whole-round heuristic of dispatch_tile stands."""
import os, sys, torch
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import modcr_hip as mh
from bench_kernels import timeit
mh.use_tuning_library(True)
dev = torch.device("cuda"); g = torch.Generator(device="cpu").manual_seed(0)
for (m, n, k) in ((92160, 768, 768), (92160, 768, 3072), (54272, 1024, 1024), (54272, 1024, 4096), (46080, 768, 768), (46080, 768, 3072)):
    a = torch.randn(m, k, generator=g).to(dev).bfloat16(); w = (torch.randn(n, k, generator=g) * 0.03).to(dev).bfloat16(); b = torch.randn(n, generator=g).to(dev)
    o = {0: torch.empty(m, n, device=dev, dtype=torch.bfloat16), -1: torch.empty(m, n, device=dev, dtype=torch.bfloat16)}
    t = {0: [], -1: []}
    for _ in range(5):
        for knob in (-1, 0):
            if knob == -1: os.environ.pop("MODCR_GEMM_T192", None)
            else: os.environ["MODCR_GEMM_T192"] = "0"
            t[knob].append(timeit(lambda: mh.linear(a, w, b, act=0, out=o[knob]), iters=10, warm=2) * 1e6)
    os.environ.pop("MODCR_GEMM_T192", None)
    med = {kk: sorted(v)[len(v) // 2] for kk, v in t.items()}
    print("M=%d N=%d K=%d: default %.1f us   256x256 forced %.1f us   equal %s" % (m, n, k, med[-1], med[0], torch.equal(o[0], o[-1])), flush=True)
