"""Attention twin of tools/stress_gemm.py: the persistent tile loop of qkv_attn4_kernel at the benched size (N = 256, S = 180,
H = 768: six tiles per workgroup), cache flushed before every launch, EVERY launch compared bit for bit with a launch that was
first checked against a torch fp32 evaluation on the device.  ITERS launches per variant (default 500).
    python tools/stress_attn.py            # on the GPU box"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multimodal-context-reasoning_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import modcr_hip as mh  # noqa: E402
import test_hip_attn_fullsize as F  # noqa: E402

dev = torch.device("cuda")
iters = int(os.environ.get("ITERS", "500"))
junk1 = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
junk2 = torch.empty(256 << 20, dtype=torch.uint8, device=dev)


def stress(name, n, t, r, h, a, mode, drop):
    sd, x, key_mask, dense, gi = F.make_case(n, t, r, h, a, mode, seed=int(os.environ.get("SEED", 5 + mode)))
    wqkv = torch.cat([sd["query.weight"], sd["key.weight"], sd["value.weight"]], 0).to(dev).to(torch.bfloat16)
    bqkv = torch.cat([sd["query.bias"], sd["key.bias"], sd["value.bias"]], 0).to(dev)
    xd = x.to(dev).to(torch.bfloat16)
    cid = None
    if gi is not None:
        c = torch.full((n, t), -1, dtype=torch.int32)
        for i, g in enumerate(gi):
            c[i, 1:1 + g.numel()] = g.to(torch.int32)
        cid = c.to(dev)
    bits = mh.pack_mask_bits(dense.to(dev)) if dense is not None else None
    km = key_mask.to(dev) if dense is None else None
    s = t + r
    lp = 128 if s <= 128 else (192 if s <= 192 else 256)
    first, bad = None, 0
    train = "dump" in name                  # a trainable layer's launch: row statistics + the Q|K|V image dump interleaved into phase B
    first_dump = None
    for i in range(iters + 1):
        junk1.copy_(junk2)
        amap = torch.zeros(n, t, r, device=dev) if mode == 3 else None
        lse = torch.empty(n, a, s, device=dev) if train else None
        dump = torch.zeros(mh.qkv_dump_numel(n, s, a), device=dev, dtype=torch.bfloat16) if train else None
        ctx, _ = mh.qkv_attn(xd, wqkv, bqkv, key_mask=km, mask_bits=bits, chunk_id=cid, align_map=amap, align_t=t if mode == 3 else 0,
                             num_heads=a, attn_dropout=(0.1, 7, 11) if drop else None, lse=lse, dump=dump)
        if train:
            if first_dump is None:
                first_dump = (lse.clone(), dump.clone())
            elif not (torch.equal(lse, first_dump[0]) and torch.equal(dump, first_dump[1])):
                bad += 1
                print("   %s launch %d: lse / dump differ from launch 0" % (name, i), flush=True)
        if first is None:
            first = ctx.clone()
            ref, _ = F.device_reference(xd, wqkv, bqkv, a, key_mask=key_mask.to(dev), dense=dense.to(dev) if dense is not None else None, cid=cid,
                                        keep_fn=(lambda ids: F.drop_keep(ids, a, s, lp, 0.1, 7, 11, dev)) if drop else None, p_drop=0.1)
            e = float((first.float() - ref).abs().max())
            per_seq = (first.float() - ref).abs().amax(dim=(1, 2))
            top = torch.topk(per_seq, 3)
            print("%s: launch 0 vs torch fp32 max|err| %.4g (scale %.3g); worst sequences %s %s" % (
                name, e, float(ref.abs().max()), top.indices.tolist(), [round(float(v), 4) for v in top.values]), flush=True)
            continue
        d = (ctx != first).any(dim=2)
        if bool(d.any()):
            bad += 1
            nz = torch.nonzero(d)
            dh = (ctx != first).view(n, s, a, -1).any(dim=3)                     # [n, row, head]
            heads = sorted(set(int(v) for v in torch.nonzero(dh)[:, 2].tolist()))
            seqs = sorted(set(int(v) for v in nz[:, 0].tolist()))
            rows = sorted(set(int(v) for v in nz[:, 1].tolist()))
            print("   %s launch %d: %d rows differ; seqs %s heads %s rows %s..%s max|diff| %.4g" % (
                name, i, int(d.sum()), seqs[:8], heads, rows[0], rows[-1], float((ctx.float() - first.float()).abs().max())), flush=True)
    print("%s bad launches: %d of %d" % (name, bad, iters), flush=True)
    return bad


CASES = [
    ("<1,192,1> N=256 S=180", 256, 80, 100, 768, 12, 1, 1),
    ("<1,192,0> N=256 S=180", 256, 80, 100, 768, 12, 1, 0),
    ("<2,192,1> N=256 S=180", 256, 80, 100, 768, 12, 2, 1),
    ("<3,192,1> N=256 S=180", 256, 80, 100, 768, 12, 3, 1),
    ("<3,192,0> N=256 S=180", 256, 80, 100, 768, 12, 3, 0),
    ("<1,128,1> N=256 S=101", 256, 1, 100, 768, 12, 1, 1),
    ("<3,128,0> N=256 S=101", 256, 50, 51, 768, 12, 3, 0),
    ("<3,128,1> N=256 S=101", 256, 50, 51, 768, 12, 3, 1),
    ("<2,128,1> N=256 S=101", 256, 50, 51, 768, 12, 2, 1),
    ("<1,192,1> N=512 S=180", 512, 80, 100, 768, 12, 1, 1),
    ("<1,192,1> + lse + dump N=512 S=180", 512, 80, 100, 768, 12, 1, 1),
    ("<2,192,1> + lse + dump N=256 S=180", 256, 80, 100, 768, 12, 2, 1),
    ("<1,128,1> + lse + dump N=512 S=106 H=1024", 512, 96, 10, 1024, 16, 1, 1),
    ("<1,256,1> N=128 S=230 H=1024", 128, 194, 36, 1024, 16, 1, 1),
    ("<2,256,0> N=128 S=230 H=1024", 128, 194, 36, 1024, 16, 2, 0),
    ("<3,256,1> N=128 S=230 H=1024", 128, 194, 36, 1024, 16, 3, 1),
]
only = os.environ.get("ONLY")
total = 0
for c in CASES:
    if only is None or only in c[0]:
        total += stress(*c)
print("TOTAL bad launches:", total)
sys.exit(1 if total else 0)
