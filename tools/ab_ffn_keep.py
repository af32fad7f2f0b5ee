#!/usr/bin/env python
"""trainable FFN (BertIntermediate + BertOutput, forward + backward) at config 3's row count with the GELU input kept
(modcr_ffn_up_gelu_keep_fwd, modcr_ffn_down_residual_ln_gelu_bwd, modcr_ffn_up_du_bwd) against the recompute route
(modcr_ffn_up_gelu_fwd, modcr_linear_residual_ln_dropout_bwd, modcr_ffn_up_gelu_bwd): torch events around 10 back-to-back
forward + backward passes each, interleaved rounds in one process"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402

dev = torch.device("cuda")
m, h, i = int(os.environ.get("M", 92160)), 768, 3072
g = torch.Generator(device="cpu").manual_seed(0)
a = torch.randn(m, h, generator=g).to(dev).bfloat16()
w1 = (torch.randn(i, h, generator=g) * 0.03).to(dev).bfloat16()
w2 = (torch.randn(h, i, generator=g) * 0.02).to(dev).bfloat16()
b1, b2 = torch.zeros(i, device=dev), torch.zeros(h, device=dev)
gam, bet = torch.ones(h, device=dev), torch.zeros(h, device=dev)
dy = torch.randn(m, h, generator=g).to(dev).bfloat16()
drop = (0.1, 7, 4096)
pre = torch.empty(m, h, device=dev)


def fwd(keep):
    if keep:
        inter, u = mh.ffn_up_gelu_keep(a, w1, b1)
    else:
        inter, u = mh.linear(a, w1, b1, act=mh.ACT_GELU), None
    y = mh.linear_dropout_residual_ln(inter, w2, b2, a, gam, bet, 1e-12, drop[0], drop[1], drop[2], pre_out=pre)
    return inter, u, y


def bwd(keep, inter, u):
    dg, db = torch.zeros(h, device=dev), torch.zeros(h, device=dev)
    if keep:
        d_pre, d_u, dw2, dbw2 = mh.ffn_down_residual_ln_gelu_bwd(dy, pre, inter, w2, gam, 1e-12, u, dg, db, dropout=drop)
        return mh.ffn_up_du_bwd(d_u, a, w1, dx_residual=d_pre)
    d_pre, d_inter, dw2, dbw2 = mh.linear_residual_ln_bwd(dy, pre, inter, w2, gam, 1e-12, dg, db, dropout=drop)
    return mh.ffn_up_gelu_bwd(d_inter, a, w1, b1, dx_residual=d_pre)


res = {True: [], False: []}
resf = {True: [], False: []}
for rnd in range(4):
    for keep in (True, False):
        for _ in range(2):
            inter, u, _ = fwd(keep)
            bwd(keep, inter, u)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        for _ in range(10):
            inter, u, _ = fwd(keep)
        e[1].record()
        for _ in range(10):
            bwd(keep, inter, u)
        e[2].record()
        torch.cuda.synchronize()
        resf[keep].append(e[0].elapsed_time(e[1]) / 10 * 1e3)
        res[keep].append(e[1].elapsed_time(e[2]) / 10 * 1e3)
for keep in (True, False):
    print("%-10s forward min %.1f us   backward min %.1f us   sum %.1f us" % ("kept" if keep else "recompute", min(resf[keep]), min(res[keep]),
                                                                       min(resf[keep]) + min(res[keep])))
