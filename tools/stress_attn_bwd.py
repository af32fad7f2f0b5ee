"""Backward twin of tools/stress_attn.py: attn_bwd6_kernel (loader waves + one compute wave per 16-key tile, one barrier per
32-query block, persistent over (sequence, head) tiles) at the shapes the steps launch, cache flushed before every launch, EVERY
launch's dx / dWqkv compared bit for bit with the first launch (whose parity the tests pin).  ITERS launches per variant.
    python tools/stress_attn_bwd.py            # on the GPU box"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multimodal-context-reasoning_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import modcr_hip as mh  # noqa: E402
import test_hip_attn_bwd as B  # noqa: E402

dev = torch.device("cuda")
iters = int(os.environ.get("ITERS", "150"))
flush = torch.empty(300 << 20, dtype=torch.uint8, device=dev)


def stress(name, n, s, t, h, a, mask, chunk, pdrop, dalign):
    w, b, x, dctx, km, dense, cid = B.make_inputs(n, s, t, h, a, mask, chunk, 4243 + n + s)
    xd, wd, bd = x.to(dev).bfloat16(), w.to(dev).bfloat16(), b.to(dev)
    bits = mh.pack_mask_bits(dense.to(dev)) if dense is not None else None
    kmd = km.to(dev) if dense is None else None
    cidd = cid.to(dev) if cid is not None else None
    drop = (pdrop, 5, 77) if pdrop > 0 else None
    lse = torch.empty(n, a, s, device=dev)
    dump = torch.empty(mh.qkv_dump_numel(n, s, a), device=dev, dtype=torch.bfloat16)
    amap = torch.zeros(n, t, s - t, device=dev) if dalign else None
    ctx, _ = mh.qkv_attn(xd, wd, bd, key_mask=kmd, mask_bits=bits, chunk_id=cidd, num_heads=a, attn_dropout=drop, lse=lse, dump=dump,
                         align_map=amap, align_t=t if dalign else 0)
    d_align = (torch.randn(n, t, s - t, device=dev) * 0.05) if dalign else None
    dcd = dctx.to(dev).bfloat16()
    first, bad = None, 0
    for i in range(iters + 1):
        flush.fill_(i & 0xff)
        dw, db = torch.empty(3 * h, h, device=dev), torch.empty(3 * h, device=dev)
        dx = mh.qkv_attn_bwd(dcd, xd, wd, bd, dw, db, key_mask=kmd, mask_bits=bits, chunk_id=cidd, num_heads=a, attn_dropout=drop,
                             d_align=d_align, align_t=t if dalign else 0, ctx=ctx, lse=lse, dump=dump)
        if first is None:
            first = (dx.clone(), dw.clone())
            assert torch.isfinite(first[0].float()).all() and torch.isfinite(first[1]).all(), name + ": non-finite gradients"
            continue
        if not (torch.equal(dx, first[0]) and torch.equal(dw, first[1])):
            bad += 1
            print("   %s launch %d: dx differs in %d elements, dW in %d" % (name, i, int((dx != first[0]).sum()), int((dw != first[1]).sum())), flush=True)
    print("%s bad launches: %d of %d" % (name, bad, iters), flush=True)
    return bad


CASES = [
    # name, N, S, T, H, A, mask, chunk-mean queries, dropout, align-map gradient
    ("bwd6<3,key,drop> N=512 S=180", 512, 180, 80, 768, 12, "key", False, 0.1, False),
    ("bwd6<3,dense,drop> N=512 S=180 chunk", 512, 180, 80, 768, 12, "dense", True, 0.1, False),
    ("bwd6<3,dense,drop,dalign> N=512 S=180", 512, 180, 80, 768, 12, "dense", True, 0.1, True),
    ("bwd6<3,key,nodrop> N=512 S=180", 512, 180, 80, 768, 12, "key", False, 0.0, False),
    ("bwd6<2,key,drop> N=512 S=101", 512, 101, 1, 768, 12, "key", False, 0.1, False),
    ("bwd6<2,key,drop> N=512 S=106 H=1024", 512, 106, 10, 1024, 16, "key", False, 0.1, False),
    ("bwd6<3,key,drop> N=77 S=129", 77, 129, 60, 768, 12, "key", False, 0.2, False),
]
only = os.environ.get("ONLY")
total = 0
for c in CASES:
    if only and only not in c[0]:
        continue
    total += stress(*c)
print("TOTAL bad launches: %d" % total)
sys.exit(1 if total else 0)
