#!/usr/bin/env python
"""Same-process A/B of two builds of the library that differ in the attention-probability dropout hash (csrc/attn_common.h):
`new` = libmodcr_hip_tuning.so as built, `old` = a side build with the other hash form, LIB_OLD=<path to that .so> (default
modcr_hip/libmodcr_hip_r4hash.so: the tuning build of the previous round's csrc/, `git show <rev>:...` into build/old and `make tuning`
there; not part of the product).  Round 4 used it to test a 24-bit-multiply form of the hash against the shipped one (two
v_mul_lo_u32): 394.7 vs 391.6 us at N = 512 -- not adopted; round 5: the two-level multiply-fold form against round 4's.
Interleaved rounds, medians."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

new = mh.use_tuning_library(True)
old = mh._load(os.environ.get("LIB_OLD", os.path.join(os.path.dirname(mh.LIB_PATH), "libmodcr_hip_r4hash.so")))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
for n, s, h in ((512, 180, 768), (256, 180, 768), (512, 106, 1024)):
    a = h // 64
    x = torch.randn(n, s, h, generator=g).to(dev).bfloat16()
    wqkv = (torch.randn(3 * h, h, generator=g) * 0.05).to(dev).bfloat16()
    bqkv = torch.randn(3 * h, generator=g).to(dev)
    mask = torch.ones(n, s, device=dev)
    res = {"new": [], "old": [], "eval": []}
    for _ in range(7):
        for name, l, pd in (("new", new, 0.1), ("old", old, 0.1), ("eval", new, 0.0)):
            mh._lib = l
            res[name].append(timeit(lambda: mh.qkv_attn(x, wqkv, bqkv, key_mask=mask, num_heads=a, attn_dropout=(pd, 7, 11) if pd else None),
                                    iters=10, warm=2) * 1e6)
    mh._lib = new
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    fl = n * (6.0 * s * h * h + 4.0 * s * s * h)
    print("N=%d S=%d H=%d: eval %.1f us (%.3f)   training, old hash %.1f us (%.3f)   training, new hash %.1f us (%.3f of 2.5 PF)"
          % (n, s, h, med["eval"], fl / med["eval"] / 2.5e9, med["old"], fl / med["old"] / 2.5e9, med["new"], fl / med["new"] / 2.5e9), flush=True)
