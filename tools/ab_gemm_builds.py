#!/usr/bin/env python
"""Same-process A/B of two BUILDS of the library on the FFN-up GEMM (the pattern of tools/ab_drop_hash.py): `new` =
libmodcr_hip_tuning.so as built, `old` = a side build of another state of csrc/gemm.hip, LIB_OLD=<path> (default
modcr_hip/libmodcr_hip_old.so; built by hand, not part of the product).  Interleaved rounds, medians, agreement with a torch fp32
product for both.  usage: ab_gemm_builds.py [SHAPES=92160x3072x768,...] [ROUNDS=7]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

new = mh.use_tuning_library(True)
old = mh._load(os.environ.get("LIB_OLD", os.path.join(os.path.dirname(mh.LIB_PATH), "libmodcr_hip_old.so")))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
shapes = os.environ.get("SHAPES", "92160x3072x768,51712x3072x768,54272x4096x1024")
for sh in shapes.split(","):
    m, n, k = (int(v) for v in sh.split("x"))
    a = torch.randn(m, k, generator=g).to(dev).bfloat16()
    w = (torch.randn(n, k, generator=g) * 0.03).to(dev).bfloat16()
    b = torch.randn(n, generator=g).to(dev)
    rows = torch.randint(0, m, (2048,), device=dev)
    for act in (1, 0):
        outs, res = {}, {"new": [], "old": []}
        for name, l in (("new", new), ("old", old)):
            mh._lib = l
            outs[name] = torch.full((m, n), 5.0, device=dev, dtype=torch.bfloat16)
            mh.linear(a, w, b, act=act, out=outs[name])
        torch.cuda.synchronize()
        ref = a[rows].float() @ w.float().t() + b
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        err = {nm: (outs[nm][rows].float() - ref).abs().max().item() for nm in outs}
        ndiff = int((outs["new"] != outs["old"]).sum().item())
        for _ in range(int(os.environ.get("ROUNDS", 7))):
            for name, l in (("new", new), ("old", old)):
                mh._lib = l
                res[name].append(timeit(lambda: mh.linear(a, w, b, act=act, out=outs[name]), iters=10, warm=2) * 1e6)
        mh._lib = new
        med = {kk: sorted(v)[len(v) // 2] for kk, v in res.items()}
        print("M=%d N=%d K=%d act=%d: old %.1f us (min %.1f)  new %.1f us (min %.1f)   max|err| vs fp32 old %.4f new %.4f   %d of %d outputs differ"
              % (m, n, k, act, med["old"], min(res["old"]), med["new"], min(res["new"]), err["old"], err["new"], ndiff, m * n), flush=True)
