"""Host-side glue between the reference's module API and libmodcr_hip: weight packing and the
per-layer launch sequence.  No arithmetic happens here -- every tensor op is a C-ABI call.

One encoder layer (CaptionBertLayer, modeling_transfomres.py:481-489 / v10:140-150) is four calls:
    modcr_qkv_attn_fwd            fused QKV projection + masked attention (+chunk-mean query)
    modcr_proj_residual_ln_fwd    BertSelfOutput
    modcr_ffn_up_gelu_fwd         BertIntermediate
    modcr_ffn_down_residual_ln_fwd BertOutput
"""
import torch

import modcr_hip as mh


def _dev(t, device, dtype=None):
    t = t.detach()
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.to(device).contiguous()


def pack_linear(sd, name, device, dtype, pad_k=None):
    """nn.Linear weights -> (W [out,in] in `dtype`, bias fp32).  pad_k zero-pads the input dim."""
    w = sd[name + ".weight"].detach().to(torch.float32)
    if pad_k is not None and pad_k != w.shape[1]:
        w = torch.nn.functional.pad(w, (0, pad_k - w.shape[1]))
    b = sd.get(name + ".bias")
    return _dev(w, device, dtype), (None if b is None else _dev(b, device, torch.float32))


def pack_ln(sd, name, device):
    return _dev(sd[name + ".weight"], device, torch.float32), _dev(sd[name + ".bias"], device, torch.float32)


PACK_FUSED = True        # tools may clear it for an A/B run: device-resident fp32 parameters are re-packed by two launches per layer


def pack_layer(sd, prefix, device, dtype):
    """Packs one BERT layer's 16 tensors; q/k/v are concatenated to one [3H,H] matrix."""
    p = prefix + "attention.self."
    names = [p + "query.weight", p + "key.weight", p + "value.weight", prefix + "attention.output.dense.weight",
             prefix + "intermediate.dense.weight", prefix + "output.dense.weight", p + "query.bias", p + "key.bias", p + "value.bias"]
    src = [sd[k].detach() for k in names]
    if PACK_FUSED and all(t.is_cuda and t.device == device and t.dtype == torch.float32 and t.is_contiguous() for t in src) and dtype in (torch.bfloat16, torch.float32):
        # device-resident fp32 parameters (the trainable layers re-pack after every optimizer step): the six matrices and the
        # q | k | v bias go through ONE conversion launch (+ one copy launch for the bias) instead of two torch.cat and four casts
        h, i = src[0].shape[1], src[4].shape[0]
        wqkv = torch.empty((3 * h, h), dtype=dtype, device=device)
        wo, w1, w2 = (torch.empty(t.shape, dtype=dtype, device=device) for t in src[3:6])
        bqkv = torch.empty((3 * h,), dtype=torch.float32, device=device)
        mh.convert_segments([(src[0], wqkv[:h]), (src[1], wqkv[h:2 * h]), (src[2], wqkv[2 * h:]), (src[3], wo), (src[4], w1), (src[5], w2)])
        mh.convert_segments([(src[6], bqkv[:h]), (src[7], bqkv[h:2 * h]), (src[8], bqkv[2 * h:])])
        layer = {"wqkv": wqkv, "bqkv": bqkv, "wo": wo, "w1": w1, "w2": w2}
        f32 = lambda k: _dev(sd[k], device, torch.float32)
        layer["bo"], layer["b1"], layer["b2"] = f32(prefix + "attention.output.dense.bias"), f32(prefix + "intermediate.dense.bias"), f32(prefix + "output.dense.bias")
        layer["ln1_g"], layer["ln1_b"] = pack_ln(sd, prefix + "attention.output.LayerNorm", device)
        layer["ln2_g"], layer["ln2_b"] = pack_ln(sd, prefix + "output.LayerNorm", device)
        return layer
    wqkv = torch.cat([sd[p + "query.weight"], sd[p + "key.weight"], sd[p + "value.weight"]], dim=0)
    bqkv = torch.cat([sd[p + "query.bias"], sd[p + "key.bias"], sd[p + "value.bias"]], dim=0)
    layer = {"wqkv": _dev(wqkv, device, dtype), "bqkv": _dev(bqkv, device, torch.float32)}
    layer["wo"], layer["bo"] = pack_linear(sd, prefix + "attention.output.dense", device, dtype)
    layer["ln1_g"], layer["ln1_b"] = pack_ln(sd, prefix + "attention.output.LayerNorm", device)
    layer["w1"], layer["b1"] = pack_linear(sd, prefix + "intermediate.dense", device, dtype)
    layer["w2"], layer["b2"] = pack_linear(sd, prefix + "output.dense", device, dtype)
    layer["ln2_g"], layer["ln2_b"] = pack_ln(sd, prefix + "output.LayerNorm", device)
    return layer


class Workspace:
    """Caller-owned scratch reused across layers (the library allocates nothing)."""

    def __init__(self):
        self.bufs = {}

    def get(self, key, nbytes, device):
        b = self.bufs.get(key)
        if b is None or b.numel() * 4 < nbytes or b.device != device:
            b = torch.empty(((nbytes + 3) // 4,), dtype=torch.float32, device=device)
            self.bufs[key] = b
        return b


def layer_forward(layer, x, num_heads, eps, key_mask=None, mask_bits=None, hist=None, chunk_id=None,
                  want_probs=False, align_map=None, align_t=0, ws=None):
    """x [N,S,H] -> y [N,S,H] (, probs).  Mask: key_mask [N,P+S] 0/1 or mask_bits [N,S,LW]."""
    n, s, h = x.shape
    ws = ws or Workspace()
    need = mh.lib().modcr_qkv_attn_workspace(n, s, 0 if hist is None else hist.shape[1], h, mh.dt_of(x))
    wsa = ws.get("attn", need, x.device) if need else None
    ctx, probs = mh.qkv_attn(x, layer["wqkv"], layer["bqkv"], key_mask=key_mask, mask_bits=mask_bits,
                             hist=hist, chunk_id=chunk_id, want_probs=want_probs, align_map=align_map,
                             align_t=align_t, num_heads=num_heads, workspace=wsa)
    pre = ws.get("preln", n * s * h * 4, x.device)
    a = mh.linear_residual_ln(ctx, layer["wo"], layer["bo"], x, layer["ln1_g"], layer["ln1_b"], eps, pre)
    inter = mh.linear(a.view(n * s, h), layer["w1"], layer["b1"], act=mh.ACT_GELU)
    y = mh.linear_residual_ln(inter, layer["w2"], layer["b2"], a, layer["ln2_g"], layer["ln2_b"], eps, pre)
    return (y, probs) if want_probs else y


# ---- trainable encoder layer: forward that keeps what the backward needs, and the backward -------------------------
# (autograd of CaptionBertLayer for the trainable-encoder variants: SURVEY 8f-1 / 8f-4, BASELINE config 3.)
# Saved per layer: x, ctx, a, inter in the storage dtype, the two pre-LayerNorm rows in fp32 and the softmax row statistics
# lse [N,A,S]; q/k/v are recomputed by modcr_qkv_attn_lse_bwd (the probabilities from them and lse), the GELU input by one
# extra GEMM.

SAVE_QKV = True          # tools may clear it for an A/B run: the trainable layers' forward dumps its Q | K | V images for the backward
PRE_F16 = True           # the pre-LayerNorm rows a trainable layer keeps for its backward are IEEE half on the bf16 route; tools may clear it
KEEP_GELU_INPUT = True   # the trainable layers' FFN-up also writes its pre-activation rows (modcr_ffn_up_gelu_keep_fwd); tools may clear it


def backward_memory_groups(model, sequences, text_len, regions, roberta_len=96, prefix_len=10):
    """(layers, sequences, seq_len, hidden) of every group of TRAINABLE encoder layers of a built ModCR model, for
    configure_backward_memory(groups=): the two Oscar encoders when calec.train_encoders is set (12 + 12 layers over text + regions),
    the prefix RoBERTa body when its layers require gradients (its own hidden size and sequence length: tokens + prefix)."""
    groups = []
    calec = getattr(model, "calec", None)
    if calec is not None and getattr(calec, "train_encoders", False):
        for enc in (calec.global_enc, calec.seq_enc):
            cfg = enc.config
            groups.append((cfg.num_hidden_layers, sequences, text_len + regions, cfg.hidden_size))
    rob = getattr(model, "roberta", None)
    layers = getattr(getattr(rob, "encoder", None), "layer", None)
    if layers is not None and any(p.requires_grad for p in layers.parameters()):
        hid = next(iter(layers.parameters())).shape[-1]
        groups.append((len(layers), sequences, roberta_len + prefix_len, hid))
    return groups


def configure_backward_memory(mode="keep", device=None, sequences=None, seq_len=None, hidden=None, layers=None, groups=None):
    """How much the trainable layers keep for their backward (run_*_ModCR.py --modcr_backward_memory):
      keep       (default) every trainable layer keeps its Q | K | V images (N x A x 3 x tile x 64 bf16: ~453 MB per layer at 128
                 examples, S = 180) and the bf16 GELU input (M x 4H: ~566 MB per layer at M = 92160): ~12-18 GB for config 3, more
                 for the 24-layer RoBERTa body; the backward then recomputes nothing (config 3: 198 -> 154 ms per step, round 3);
      recompute  neither is kept: the attention backward re-projects q/k/v from x and the FFN backward re-runs the up product
                 (the round-2 routes) -- for batches that no longer fit;
      auto       keep if the estimate fits in half of the device's free memory (torch.cuda.mem_get_info), else recompute.  The
                 estimate sums over `groups` = [(layers, sequences, seq_len, hidden), ...] (backward_memory_groups(model, ...): every
                 group of trainable layers with its own shape), or the single group given by the four scalar arguments.
    Returns the mode chosen."""
    global SAVE_QKV, KEEP_GELU_INPUT
    if mode == "auto":
        choice = "keep"
        if groups is None and sequences and seq_len and hidden and layers:
            groups = [(layers, sequences, seq_len, hidden)]
        if device is not None and groups:
            need = 0
            for l_, n_, s_, h_ in groups:
                tile = 128 if s_ <= 128 else (192 if s_ <= 192 else 256)
                need += l_ * (n_ * 3 * tile * h_ * 2 + n_ * s_ * 4 * h_ * 2)
            free, _ = torch.cuda.mem_get_info(device)
            if need > free // 2:
                choice = "recompute"
        mode = choice
    if mode not in ("keep", "recompute"):
        raise ValueError("modcr_backward_memory: keep | recompute | auto, got %r" % (mode,))
    SAVE_QKV = KEEP_GELU_INPUT = (mode == "keep")
    return mode


def _sub_ln_fwd(a_in, w, b, resid, gamma, beta, eps, p, dt):
    """LN(dropout(a_in.W^T + b) + resid): returns (pre-LN rows, output, (p, seed, offset) or None).  One C-ABI call:
    the GEMM, then ONE row pass that applies the mask, adds the residual, writes the fp32 pre-LN rows the backward wants and
    normalises (round 1: fp32-out GEMM + dropout pass + add pass + LayerNorm pass = 28 bytes per element, 10 now)."""
    m, n = a_in.reshape(-1, a_in.shape[-1]).shape[0], w.shape[0]
    # IEEE-half copies where the backward runs its bf16 route (N in {256, 512, 768, 1024}): the forward's row pass writes and the
    # LayerNorm backward reads 2 bytes per element instead of 4; the forward itself still normalises the fp32 values
    half = PRE_F16 and dt == mh.BF16 and n % 256 == 0 and n <= 1024
    pre = torch.empty((m, n), dtype=torch.float16 if half else torch.float32, device=a_in.device)
    drop = None
    if p > 0.0:
        seed, off = mh.DROPOUT.take(m * n)
        drop = (p, seed, off)
    y = mh.linear_dropout_residual_ln(a_in, w, b, resid, gamma, beta, eps, *(drop or (0.0, 0, 0)), pre_out=pre)
    return pre.view(*resid.shape), y, drop


def attn_dropout_supported(x, num_heads):
    """shapes whose attention kernels carry the attention-probability dropout, forward and backward: every call with S <= 256 (the
    bf16 tile / older kernels and the MFMA / exact backward cores; the exact-fp32 route since round 5)"""
    n, s, h = x.shape
    return s <= 256


def layer_forward_train(layer, x, num_heads, eps, key_mask=None, mask_bits=None, chunk_id=None, p=0.0, attn_p=0.0,
                        align_map=None, align_t=0, side_post_dropout=True):
    """p: hidden_dropout_prob of BertSelfOutput / BertOutput in training mode (a_bert:369-373, :446-451);
    attn_p: attention_probs_dropout_prob (modeling_bert.py:69), applied inside the attention kernels;
    align_map [N,T,R] fp32 (+= head-summed text->region probabilities of this layer, v10:982), align_t = T;
    side_post_dropout: the map sums the probabilities AFTER that dropout (v10:94-106, the reference's semantics) and its gradient
    re-enters under the same mask; False = the un-dropped probabilities"""
    n, s, h = x.shape
    adrop = None
    if attn_p > 0.0:
        if not attn_dropout_supported(x, num_heads):
            raise NotImplementedError("attention-probability dropout in a trainable layer needs S <= 256 (S=%d, dtype=%s)" % (s, x.dtype))
        seed, off = mh.DROPOUT.take(n * num_heads * s * s)
        adrop = (float(attn_p), seed, off)
    # row statistics of the softmax for the five-product attention backward (csrc/attn_bwd.hip), where the forward runs a tile kernel
    lse = torch.empty((n, num_heads, s), dtype=torch.float32, device=x.device) if mh.lse_supported(x, num_heads) else None
    # ... and the Q | K | V images the kernel held in LDS (72 KB per sequence and head at S = 180: 453 MB per layer at 128 examples,
    # 11 GB over 24 layers of the 288 GB), so that the backward recomputes no projection.
    dump = None
    if lse is not None and SAVE_QKV:
        dump = torch.empty((mh.qkv_dump_numel(n, s, num_heads),), dtype=torch.bfloat16, device=x.device)
    post = bool(side_post_dropout and adrop is not None and align_map is not None)
    ctx, _ = mh.qkv_attn(x, layer["wqkv"], layer["bqkv"], key_mask=key_mask, mask_bits=mask_bits, chunk_id=chunk_id,
                         num_heads=num_heads, attn_dropout=adrop, align_map=align_map, align_t=align_t, lse=lse, dump=dump,
                         side_post_dropout=post)
    x2 = x.reshape(n * s, h)
    dt = mh.dt_of(x)
    pre1, a, drop1 = _sub_ln_fwd(ctx.reshape(n * s, h), layer["wo"], layer["bo"], x2, layer["ln1_g"], layer["ln1_b"], eps, p, dt)
    # the GELU input is kept where the persistent GEMM takes the shape (566 MB per layer at M = 92160: 6.8 GB over 12 layers):
    # the backward then recomputes no FFN-up product
    u = None
    if KEEP_GELU_INPUT and mh.ffn_keep_supported(a, layer["w1"], layer["b1"]):
        inter, u = mh.ffn_up_gelu_keep(a, layer["w1"], layer["b1"])
    else:
        inter = mh.linear(a, layer["w1"], layer["b1"], act=mh.ACT_GELU)
    pre2, y, drop2 = _sub_ln_fwd(inter, layer["w2"], layer["b2"], a, layer["ln2_g"], layer["ln2_b"], eps, p, dt)
    saved = dict(x=x, ctx=ctx, pre1=pre1, a=a, inter=inter, u=u, pre2=pre2, num_heads=num_heads, eps=eps,
                 key_mask=key_mask, mask_bits=mask_bits, chunk_id=chunk_id, drop1=drop1, drop2=drop2, adrop=adrop,
                 align_t=align_t if align_map is not None else 0, lse=lse, dump=dump, side_post=post)
    return y.view(n, s, h), saved


def _sub_ln_bwd(dy, pre, a_in, w, gamma, eps, dgamma, dbeta, drop, mfma, dw_out=None, db_out=None):
    """backward of _sub_ln_fwd: (d_pre = residual-branch gradient, d_a_in, dW, dbias); dw_out / db_out: write them there"""
    n = w.shape[0]
    if drop is None or (mfma and a_in.dtype == torch.bfloat16 and n % 256 == 0 and n <= 1024):
        return mh.linear_residual_ln_bwd(dy, pre, a_in, w, gamma, eps, dgamma, dbeta, dropout=drop, dw_out=dw_out, db_out=db_out)
    d_pre = mh.layernorm_bwd(dy, pre, gamma, eps, dgamma, dbeta)          # exact-fp32 route: separate passes
    d_sub = mh.dropout(d_pre, *drop)                       # same (seed, offset) as the forward: same mask
    dw = dw_out if dw_out is not None else torch.empty(w.shape, dtype=torch.float32, device=w.device)
    db = db_out if db_out is not None else torch.empty((w.shape[0],), dtype=torch.float32, device=w.device)
    mh.linear_bwd_weight(d_sub, a_in, dw, db, mfma=mfma)
    da = mh.linear_bwd_input(d_sub, w, out_dtype=mh.dt_of(a_in), mfma=mfma)
    return d_pre, da, dw, db


def layer_backward(layer, saved, dy, mfma=True, d_align=None, outs=None):
    """dy [N,S,H] -> (dx [N,S,H] in x's dtype, {HF parameter name: fp32 gradient}).  Four C-ABI composites:
    modcr_ffn_down_residual_ln_bwd, modcr_ffn_up_gelu_bwd, modcr_proj_residual_ln_bwd, modcr_qkv_attn_bwd
    (the two residual-gradient sums ride in the epilogues of the dX GEMMs).  `mfma` is implied by the storage dtype (bf16 = MFMA route).
    outs: {HF parameter name: fp32 tensor} -- gradients listed there are WRITTEN into the given tensor (dense weights / biases: a view of
    the flat gradient buffer on its first use in an accumulation window) or ACCUMULATED into it (the four LayerNorm parameters: the
    kernels add) instead of into a fresh tensor; the returned dict holds the same tensor objects."""
    outs = outs or {}
    x, ctx, a, inter = saved["x"], saved["ctx"], saved["a"], saved["inter"]
    n, s, h = x.shape
    m = n * s
    eps = saved["eps"]
    dev, f32 = x.device, torch.float32
    g = {}

    def zeros(*shape):
        return torch.zeros(*shape, dtype=f32, device=dev)

    dy2 = dy.reshape(m, h)
    if dy2.dtype != f32 and not (mfma and h % 256 == 0 and h <= 1024):
        dy2 = mh.convert(dy2, mh.F32)          # (the bf16 route's LayerNorm backward reads bf16 gradients directly)
    # BertOutput: y = LN(inter.W2^T + b2 + a)
    ln_names = ("output.LayerNorm.weight", "output.LayerNorm.bias", "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias")
    if all(nm in outs for nm in ln_names):
        dg2, db2, dg1, db1 = (outs[nm] for nm in ln_names)       # accumulated in place
    else:
        lnz = zeros(4, h)                  # the four LayerNorm-parameter gradients (the kernels accumulate into them): one fill
        dg2, db2, dg1, db1 = lnz[0], lnz[1], lnz[2], lnz[3]
    u = saved.get("u")
    if u is not None:
        # kept GELU input: the dX product of BertOutput leaves d_u = (d_sub.W2) * gelu'(u) and BertIntermediate needs two products
        # (the bias gradient of BertIntermediate = colsum(d_u) comes out of the same epilogue; dW1 is then formed transposed with d_u
        # token-major: no transpose of the [M, 4H] operand)
        d_pre2, d_u, dw2, dbw2, db_u = mh.ffn_down_residual_ln_gelu_bwd(dy2, saved["pre2"], inter, layer["w2"], layer["ln2_g"], eps, u,
                                                                         dg2, db2, dropout=saved.get("drop2"), want_db_u=True,
                                                                         dw_out=outs.get("output.dense.weight"), db_out=outs.get("output.dense.bias"),
                                                                         db_u_out=outs.get("intermediate.dense.bias"))
        d_a, dw1, dbw1 = mh.ffn_up_du_bwd(d_u, a, layer["w1"], dx_residual=d_pre2, db1=db_u, dw_out=outs.get("intermediate.dense.weight"))
    else:
        d_pre2, d_inter, dw2, dbw2 = _sub_ln_bwd(dy2, saved["pre2"], inter, layer["w2"], layer["ln2_g"], eps, dg2, db2,
                                                 saved.get("drop2"), mfma, dw_out=outs.get("output.dense.weight"), db_out=outs.get("output.dense.bias"))
        # BertIntermediate: inter = gelu(a.W1^T + b1)
        # (the residual-branch gradient d_pre2 is added in the epilogue of the dX GEMM: no pass of its own)
        d_a, dw1, dbw1 = mh.ffn_up_gelu_bwd(d_inter, a, layer["w1"], layer["b1"], dx_residual=d_pre2,
                                            dw_out=outs.get("intermediate.dense.weight"), db_out=outs.get("intermediate.dense.bias"))
    g["output.LayerNorm.weight"], g["output.LayerNorm.bias"] = dg2, db2
    g["output.dense.weight"], g["output.dense.bias"] = dw2, dbw2
    g["intermediate.dense.weight"], g["intermediate.dense.bias"] = dw1, dbw1
    # BertSelfOutput: a = LN(ctx.Wo^T + bo + x)
    d_pre1, d_ctx, dwo, dbo = _sub_ln_bwd(d_a, saved["pre1"], ctx.reshape(m, h), layer["wo"], layer["ln1_g"], eps, dg1, db1,
                                          saved.get("drop1"), mfma, dw_out=outs.get("attention.output.dense.weight"),
                                          db_out=outs.get("attention.output.dense.bias"))
    g["attention.output.LayerNorm.weight"], g["attention.output.LayerNorm.bias"] = dg1, db1
    g["attention.output.dense.weight"], g["attention.output.dense.bias"] = dwo, dbo
    # self-attention
    dwqkv, dbqkv = outs.get("attention.self.qkv.weight"), outs.get("attention.self.qkv.bias")      # (spans of the flat gradient buffer)
    dwqkv = dwqkv.view(3 * h, h) if dwqkv is not None else torch.empty(3 * h, h, dtype=f32, device=dev)
    dbqkv = dbqkv if dbqkv is not None else torch.empty(3 * h, dtype=f32, device=dev)
    dx = mh.qkv_attn_bwd(d_ctx.view(n, s, h), x, layer["wqkv"], layer["bqkv"], dwqkv, dbqkv,
                         key_mask=saved["key_mask"], mask_bits=saved["mask_bits"], chunk_id=saved["chunk_id"],
                         num_heads=saved["num_heads"], attn_dropout=saved.get("adrop"),
                         d_align=d_align if saved.get("align_t") else None, align_t=saved.get("align_t", 0),
                         dx_residual=d_pre1.view(n, s, h),
                         ctx=ctx if saved.get("lse") is not None else None, lse=saved.get("lse"), dump=saved.get("dump"),
                         side_post_dropout=bool(saved.get("side_post")))
    for i, nm in enumerate(("query", "key", "value")):
        g["attention.self.%s.weight" % nm] = dwqkv[i * h:(i + 1) * h]
        g["attention.self.%s.bias" % nm] = dbqkv[i * h:(i + 1) * h]
    return dx.view(n, s, h), g
