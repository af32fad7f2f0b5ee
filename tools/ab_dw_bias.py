#!/usr/bin/env python
"""Same-process A/B of two builds on modcr_linear_bwd_weight WITH a bias gradient (the call a trainable layer makes four times):
`new` = libmodcr_hip_tuning.so as built (bias gradient = row-block partials by plain stores, folded by extra blocks of the split-K
reduction: no memset, no atomics, reproducible), `old` = LIB_OLD (default modcr_hip/libmodcr_hip_prevdb.so: a side build of the
commit before -- `git show <rev>:<file>` of csrc/ into build/old/csrc, include/modcr_hip.h into build/include, `make tuning` there;
hipMemsetAsync + float atomics; not part of the product).  Shapes: the four weight gradients of an Oscar-base layer at M = 92160 and two head shapes.
Interleaved rounds, medians of whole calls (every launch of the call)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

new = mh.use_tuning_library(True)
old = mh._load(os.environ.get("LIB_OLD", os.path.join(os.path.dirname(mh.LIB_PATH), "libmodcr_hip_prevdb.so")))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
for m, n, k, dt in ((92160, 768, 768, torch.bfloat16), (92160, 768, 3072, torch.bfloat16), (92160, 3072, 768, torch.bfloat16),
                    (92160, 2304, 768, torch.bfloat16), (512, 768, 1536, torch.float32), (512, 3840, 768, torch.float32)):
    dy = torch.randn(m, n, generator=g).to(dev).to(dt)
    x = torch.randn(m, k, generator=g).to(dev).bfloat16()
    dw, db = torch.empty(n, k, device=dev), torch.empty(n, device=dev)
    res, outs = {"new": [], "old": []}, {}
    for _ in range(7):
        for name, l in (("new", new), ("old", old)):
            mh._lib = l
            res[name].append(timeit(lambda: mh.linear_bwd_weight(dy, x, dw, db, mfma=True), iters=10, warm=2) * 1e6)
            outs[name] = (dw.clone(), db.clone())
    mh._lib = new
    med = {k_: sorted(v)[len(v) // 2] for k_, v in res.items()}
    ref_b = dy.double().sum(0).float()
    print("dW [%d x %d] + db over M = %d (%s dY): old %.1f us   new %.1f us   dW bit-equal: %s   |db - fp64|/max: old %.2e new %.2e"
          % (n, k, m, str(dt).split(".")[-1], med["old"], med["new"], torch.equal(outs["new"][0], outs["old"][0]),
             float((outs["old"][1] - ref_b).abs().max() / ref_b.abs().max()), float((outs["new"][1] - ref_b).abs().max() / ref_b.abs().max())), flush=True)
