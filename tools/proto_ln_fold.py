#!/usr/bin/env python
"""LayerNorm-fold PROTOTYPE, consumer side (tuning library only): the fused attention kernel reading the
PRE-LayerNorm rows of the previous sublayer with gamma folded into Wqkv and rstd * acc - rstd * mu * c + d applied in its image pass
(`qkv_attn4_kernel<..., FOLD = 1>`, entry modcr_tuning_qkv_attn_fold_fwd) against the shipped call on the LayerNorm-ed rows:
accuracy of both against an fp32 evaluation, and the kernel time of the two in one process (interleaved rounds, medians)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from bench_kernels import timeit  # noqa: E402

lib = mh.use_tuning_library(True)
fn = lib.modcr_tuning_qkv_attn_fold_fwd
fn.restype = ctypes.c_int32
fn.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int32] * 4 + [ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
n, s, h = int(os.environ.get("N", 512)), 180, 768
a = h // 64
pre = torch.randn(n, s, h, generator=g) + 0.5 * torch.randn(n, s, 1, generator=g)
pre[..., ::97] *= 6.0
gamma, beta = 1.0 + 0.1 * torch.randn(h, generator=g), 0.1 * torch.randn(h, generator=g)
w, b = torch.randn(3 * h, h, generator=g) * 0.05, 0.1 * torch.randn(3 * h, generator=g)
mask = torch.ones(n, s)
mask[::3, s - 17:] = 0
pre_b = pre.to(dev).bfloat16()
pf = pre_b.float()
mu, var = pf.mean(-1, keepdim=True), pf.var(-1, unbiased=False, keepdim=True)
rstd = (var + 1e-12).rsqrt()
y = ((pf - mu) * rstd * gamma.to(dev) + beta.to(dev)).bfloat16()            # what the LayerNorm pass hands the shipped kernel
stats = torch.cat([rstd, rstd * mu], -1).reshape(-1, 2).contiguous()         # (rstd, rstd * mu) per row
wfold = (w * gamma[None, :]).to(dev).bfloat16()
cfold = wfold.float().sum(-1).contiguous()
dvec = (w @ beta + b).to(dev).contiguous()
wb, bb, km = w.to(dev).bfloat16(), b.to(dev), mask.to(dev)
ctx_f = torch.empty_like(pre_b)


def fold(p=0.0):
    rc = fn(pre_b.data_ptr(), wfold.data_ptr(), dvec.data_ptr(), cfold.data_ptr(), stats.data_ptr(), km.data_ptr(), ctx_f.data_ptr(),
            n, s, h, a, p, 7, 11, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.modcr_last_error()
    return ctx_f


def shipped(p=0.0):
    return mh.qkv_attn(y, wb, bb, key_mask=km, num_heads=a, attn_dropout=(p, 7, 11) if p else None)[0]


# accuracy on a few sequences against fp32 (LayerNorm of the fp32 rows, fp32 projection and attention)
idx = [0, 3, 77, n - 1]
x32 = torch.nn.functional.layer_norm(pre[idx], (h,), gamma, beta, 1e-12)
qkv = x32 @ w.t() + b
q, k, v = [t.view(len(idx), s, a, 64).transpose(1, 2) for t in qkv.split(h, -1)]
add = (1.0 - mask[idx])[:, None, None, :] * -10000.0
ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0 + add, -1) @ v).transpose(1, 2).reshape(len(idx), s, h)
scale = float(ref.abs().max())
for name, got in (("shipped (LayerNorm pass -> bf16 rows -> kernel)", shipped()), ("folded  (pre rows, gamma in W, stats in the image pass)", fold())):
    err = (got[idx].float().cpu() - ref).abs()
    print("%-58s max|err| %.3e (%.2e of scale %.2f)  rms %.3e" % (name, float(err.max()), float(err.max()) / scale, scale, float(err.pow(2).mean().sqrt())), flush=True)
d = (shipped().float() - fold().float()).abs().max().item()
print("shipped vs folded, all %d sequences: max|diff| %.3e" % (n, d))
d = (shipped(0.1).float() - fold(0.1).float()).abs().max().item()
print("the same with attention dropout 0.1 (one mask): max|diff| %.3e" % d, flush=True)
res = {}
for _ in range(7):
    for name, f in (("shipped eval", lambda: shipped()), ("folded eval", lambda: fold()), ("shipped train", lambda: shipped(0.1)), ("folded train", lambda: fold(0.1))):
        res.setdefault(name, []).append(timeit(f, iters=10, warm=2) * 1e6)
med = {k_: sorted(v_)[len(v_) // 2] for k_, v_ in res.items()}
print("N=%d S=%d H=%d kernel + launch, medians of 7 x 10: " % (n, s, h) + ", ".join("%s %.1f us" % kv for kv in med.items()), flush=True)
