#!/usr/bin/env python
"""VERDICT r04 item 7 -- folding a LayerNorm pass into the GEMMs around it, priced before building it.
The proposed form: producer GEMM epilogue writes pre = dropout(acc + b) + residual (+ per-row sum / sum-of-squares partials), the
consumer GEMM runs on `pre` with W' = diag(gamma) W and applies rstd (acc - mu c) + d in its epilogue, the next sublayer rebuilds its
residual from pre + the row statistics.  Whatever else it needs, its producer is a residual-carrying GEMM epilogue -- which the
library has (modcr_linear_fwd with a residual operand).  GPU part: that epilogue alone (no dropout mask, no statistics: a lower
bound of the producer) against what it would replace, GEMM with IEEE-half rows + the LayerNorm/dropout row pass, at the bench's
M = 92160 for the two sublayer shapes.  CPU part (always): the numerics of the folded arithmetic on one layer's data -- FFN-up
pre-activations from (i) fp32, (ii) the shipped route (LayerNorm output rounded to bf16, bf16 weights), (iii) the folded route
(pre rounded to bf16, bf16(gamma o W), fp32 correction terms)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multimodal-context-reasoning_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def numerics():
    import helpers as H
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    rs = np.random.RandomState(3)
    h, m = 768, 4096
    sd = {}
    H.layer_weights(rs, sd, "", h, 4 * h)
    sd = H.to_torch(sd)
    # pre-LayerNorm rows with the statistics of a residual stream: unit-variance features, a per-row offset, a few outlier features
    pre = torch.from_numpy(rs.standard_normal((m, h)).astype(np.float32))
    pre += torch.from_numpy(rs.standard_normal((m, 1)).astype(np.float32)) * 0.5
    pre[:, ::97] *= 6.0
    g, b = sd["attention.output.LayerNorm.weight"], sd["attention.output.LayerNorm.bias"]
    w, wb = sd["intermediate.dense.weight"], sd["intermediate.dense.bias"]
    mu, var = pre.mean(-1, keepdim=True), pre.var(-1, unbiased=False, keepdim=True)
    rstd = (var + 1e-12).rsqrt()
    ref = (((pre - mu) * rstd) * g + b).double() @ w.double().t() + wb.double()
    # (ii) shipped: pre as IEEE half, LayerNorm in fp32, output rounded to bf16, bf16 weights
    ph = pre.to(torch.float16).float()
    mu2, var2 = ph.mean(-1, keepdim=True), ph.var(-1, unbiased=False, keepdim=True)
    a = bf(((ph - mu2) * (var2 + 1e-12).rsqrt()) * g + b)
    ship = a @ bf(w).t() + wb
    # (iii) folded: pre rounded to bf16 (the MFMA operand), statistics from the SAME rounded rows, W' = bf16(gamma o W)
    pb = bf(pre)
    mu3, var3 = pb.mean(-1, keepdim=True), pb.var(-1, unbiased=False, keepdim=True)
    rstd3 = (var3 + 1e-12).rsqrt()
    wf = bf(w * g[None, :])
    c = wf.sum(-1)                                   # sum_k gamma_k W_nk as the MFMA sees it
    d = w @ b + wb
    fold = rstd3 * (pb @ wf.t() - mu3 * c[None, :]) + d
    scale = float(ref.abs().max())
    for name, got in (("shipped route (LN output -> bf16)", ship), ("folded route (pre -> bf16, gamma in W)", fold)):
        err = (got.double() - ref).abs()
        print("numerics, FFN-up pre-activation [%d x %d], max|ref| %.2f: %-40s max|err| %.3e (%.2e of scale)  rms %.3e"
              % (m, 4 * h, scale, name, float(err.max()), float(err.max()) / scale, float(err.pow(2).mean().sqrt())), flush=True)


def timing():
    import modcr_hip as mh
    from bench_kernels import timeit
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(0)
    m, h = 92160, 768
    for k, what in ((768, "BertSelfOutput (proj)"), (3072, "BertOutput (FFN-down)")):
        a = torch.randn(m, k, generator=g).to(dev).bfloat16()
        w = (torch.randn(h, k, generator=g) * 0.03).to(dev).bfloat16()
        b = torch.zeros(h, device=dev)
        res = torch.randn(m, h, generator=g).to(dev).bfloat16()
        gam, bet = torch.ones(h, device=dev), torch.zeros(h, device=dev)
        t_cur = timeit(lambda: mh.linear_dropout_residual_ln(a, w, b, res, gam, bet, 1e-12, p=0.3, seed=1, offset=2), iters=20, warm=3) * 1e6
        t_gemm = timeit(lambda: mh.linear(a, w, b, out_dtype=mh.F16), iters=20, warm=3) * 1e6
        t_res = timeit(lambda: mh.linear(a, w, b, residual=res), iters=20, warm=3) * 1e6
        print("%s, M = %d, K = %d: shipped GEMM (half rows) + LayerNorm/dropout pass %.1f us (GEMM alone %.1f, the pass %.1f); "
              "GEMM with the residual in its epilogue (the folded form's producer WITHOUT its dropout mask and row statistics) %.1f us "
              "-> the most the fold can return on this sublayer: %.1f us, before the consumer's and the next sublayer's added epilogue work"
              % (what, m, k, t_cur, t_gemm, t_cur - t_gemm, t_res, t_cur - t_res), flush=True)


if __name__ == "__main__":
    numerics()
    if torch.cuda.is_available():
        timing()
