#!/usr/bin/env python
"""Same-process A/B of two builds of the library on modcr_qkv_attn_bwd at the bench sizes (five-product core on the forward's
row statistics + Q | K | V dump, attention dropout 0.1): `new` = libmodcr_hip_tuning.so as built, `old` = LIB_OLD (default
modcr_hip/libmodcr_hip_prev.so: the tuning build of the previous commit's csrc/, built by hand under build/; not part of the
product).  The projection / weight-gradient GEMMs of the call are the same code in both, so the difference of the whole-call times
is the attention core's.  Interleaved rounds, medians; outputs compared."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

new = mh.use_tuning_library(True)
old = mh._load(os.environ.get("LIB_OLD", os.path.join(os.path.dirname(mh.LIB_PATH), "libmodcr_hip_prev.so")))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
for n, s, h, dense in ((512, 180, 768, False), (512, 180, 768, True), (512, 106, 1024, False), (512, 101, 768, False)):
    a = h // 64
    x = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
    dctx = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
    wqkv = (torch.randn(3 * h, h, generator=g) * 0.03).to(dev).bfloat16()
    bqkv = torch.zeros(3 * h, device=dev)
    km = torch.ones(n, s, device=dev)
    bits = mh.pack_mask_bits((torch.rand(n, s, s, generator=g) < 0.8).float().to(dev)) if dense else None
    drop = (0.1, 7, 11)
    res, outs = {"new": [], "old": []}, {}
    for name, l in (("new", new), ("old", old)):
        mh._lib = l
        lse = torch.empty(n, a, s, device=dev)
        dump = torch.empty(mh.qkv_dump_numel(n, s, a), device=dev, dtype=torch.bfloat16)
        ctx, _ = mh.qkv_attn(x, wqkv, bqkv, key_mask=None if dense else km, mask_bits=bits, num_heads=a, attn_dropout=drop, lse=lse, dump=dump)
        outs[name] = (lse, dump, ctx)
    for _ in range(7):
        for name, l in (("new", new), ("old", old)):
            mh._lib = l
            lse, dump, ctx = outs[name]
            dw, db = torch.empty(3 * h, h, device=dev), torch.empty(3 * h, device=dev)
            fn = lambda: mh.qkv_attn_bwd(dctx, x, wqkv, bqkv, dw, db, key_mask=None if dense else km, mask_bits=bits, num_heads=a,
                                         attn_dropout=drop, ctx=ctx, lse=lse, dump=dump)
            res[name].append(timeit(fn, iters=6, warm=1) * 1e6)
            outs[name + "_dx"] = fn()
    mh._lib = new
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    same = torch.equal(outs["new_dx"], outs["old_dx"])
    print("N=%d S=%d H=%d dense=%s: whole modcr_qkv_attn_bwd call old %.1f us, new %.1f us (core: %+.1f us); dx bit-equal: %s"
          % (n, s, h, dense, med["old"], med["new"], med["new"] - med["old"], same), flush=True)
