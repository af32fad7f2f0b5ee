"""CPU oracle for the ModCR hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A clean-room fp32/fp64 restatement (plain torch CPU ops, functional style, weights passed as a
state-dict with the reference's HF key names) of the arithmetic on the path SURVEY.md section 8
scopes.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module, and only as the checker.  The product (multimodal-context-reasoning_amd/) never does: it
fails loudly when the HIP library is missing.

Pinning: tests/test_oracle_golden.py checks every function below against tests/golden/*.npz, which
tools/gen_golden.py produced by importing the reference's own modules from /root/reference in the
build container (shim list: tools/ref_shims.py).  The one piece that cannot be pinned is the
RoBERTa-with-prefix body (`local_transformers`, absent from the reference tree): callers pass it
as a stub, see SURVEY.md section 8(c) -- "parity unpinned" for that boundary only.

Every function cites the reference file:line it follows.  Paths are relative to /root/reference;
`a_bert` = a_transformers.zip!a_transformers/modeling_bert.py, `v10` =
modeling/modeling_vcr_chunkalign_v10.py.
"""
import math

import torch
import torch.nn.functional as F

NEG = -10000.0


def _lin(x, sd, name):
    w = sd[name + ".weight"]
    b = sd.get(name + ".bias")
    return F.linear(x, w, b)


def _ln(x, sd, name, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def gelu_erf(x):
    """transformers.activations.ACT2FN['gelu'] (exact erf form), used at a_bert:425-437."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def extend_mask(mask):
    """modeling_transfomres.py:628-641 / v10:289-314: [N,S]->[N,1,1,S], [N,T,T]->[N,1,T,T], (1-m)*-1e4."""
    if mask.dim() == 2:
        m = mask[:, None, None, :]
    elif mask.dim() == 3:
        m = mask[:, None, :, :]
    else:
        raise NotImplementedError
    return (1.0 - m) * NEG


def chunk_mean_query(q, gather_index):
    """v10:66-78.  q [N,S,H] (before the head split); gather_index: list of int64 [len_n].

    Rows 1..len_n of sample n are replaced by the mean query of the chunk they belong to; row 0,
    rows > len_n and all image rows are left untouched.  chunk_len comes from len(offsets[n][c])
    in the reference, which equals the count of tokens gather_index maps to c
    (Data/VCRChunkAlign.py:666-670)."""
    out = q.clone()
    for n, gi in enumerate(gather_index):
        gi = gi.to(torch.int64)
        ln = gi.numel()
        if ln == 0:
            continue
        nchunk = int(gi.max().item()) + 1
        seg = q[n, 1:ln + 1]
        acc = torch.zeros(nchunk, q.shape[-1], dtype=q.dtype).index_add(0, gi, seg)
        cnt = torch.zeros(nchunk, dtype=q.dtype).index_add(0, gi, torch.ones(ln, dtype=q.dtype))
        mean = acc / cnt[:, None]
        out[n, 1:ln + 1] = mean[gi]
    return out


def self_attention(x, mask_add, sd, prefix, num_heads, history_state=None, gather_index=None):
    """modeling_bert.py:34-75 (global_enc) and v10:55-107 (seq_enc, do_chunk_cross when
    gather_index is given).  x [N,S,H]; mask_add additive, broadcastable to [N,A,S,L];
    history_state [N,P,H] is prepended for K and V only (modeling_bert.py:36-40).
    Returns (ctx [N,S,H], probs [N,A,S,L])."""
    n, s, h = x.shape
    d = h // num_heads
    xs = x if history_state is None else torch.cat([history_state, x], dim=1)
    q = _lin(x, sd, prefix + "query")
    k = _lin(xs, sd, prefix + "key")
    v = _lin(xs, sd, prefix + "value")
    if gather_index is not None:
        q = chunk_mean_query(q, gather_index)

    def split(t):
        return t.view(n, t.shape[1], num_heads, d).permute(0, 2, 1, 3)

    scores = torch.matmul(split(q), split(k).transpose(-1, -2)) / math.sqrt(d)
    scores = scores + mask_add
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, split(v)).permute(0, 2, 1, 3).reshape(n, s, h)
    return ctx, probs


def self_output(ctx, x, sd, prefix, eps):
    """a_bert:362-373 BertSelfOutput: LN(dense(ctx) + x)  (dropout = identity in eval)."""
    return _ln(_lin(ctx, sd, prefix + "dense") + x, sd, prefix + "LayerNorm", eps)


def ffn(a, sd, prefix, eps):
    """a_bert:425-451 BertIntermediate + BertOutput: LN(dense2(gelu(dense1(a))) + a)."""
    inter = gelu_erf(_lin(a, sd, prefix + "intermediate.dense"))
    return _ln(_lin(inter, sd, prefix + "output.dense") + a, sd, prefix + "output.LayerNorm", eps)


def bert_layer(x, mask_add, sd, prefix, num_heads, eps, history_state=None, gather_index=None):
    """modeling_transfomres.py:481-489 / v10:140-150 CaptionBertLayer."""
    ctx, probs = self_attention(x, mask_add, sd, prefix + "attention.self.", num_heads,
                                history_state, gather_index)
    a = self_output(ctx, x, sd, prefix + "attention.output.", eps)
    return ffn(a, sd, prefix, eps), probs


def embeddings(input_ids, token_type_ids, sd, prefix, eps, position_ids=None, pad_token_id=0):
    """a_bert:184-211 BertEmbeddings: LN(word + type + pos).  The word table is nn.Embedding(..., padding_idx=config.pad_token_id)
    (a_bert:171): same values, but row pad_token_id receives NO gradient -- it matters once the encoders are trained (the padded
    positions do reach the loss: the cross-attention of v10:856-870 reads them unmasked)."""
    t = input_ids.shape[1]
    if position_ids is None:
        position_ids = torch.arange(t)[None, :]
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    e = (torch.nn.functional.embedding(input_ids, sd[prefix + "word_embeddings.weight"], padding_idx=pad_token_id)
         + sd[prefix + "token_type_embeddings.weight"][token_type_ids]
         + sd[prefix + "position_embeddings.weight"][position_ids])
    return _ln(e, sd, prefix + "LayerNorm", eps)


def img_embed(img_feats, sd, prefix, cfg):
    """modeling_transfomres.py:676-681 / v10:339-343: LN_img(img_embedding(img)) if use_img_layernorm."""
    e = _lin(img_feats, sd, prefix + "img_embedding")
    if cfg.get("use_img_layernorm", 1):
        e = _ln(e, sd, prefix + "LayerNorm", cfg.get("img_layer_norm_eps", 1e-12))
    return e


def pooler(h, sd, prefix):
    """a_bert:634-646 BertPooler: tanh(dense(h[:,0]))."""
    return torch.tanh(_lin(h[:, 0], sd, prefix + "dense"))


def bert_img_model(sd, prefix, cfg, input_ids, token_type_ids=None, attention_mask=None,
                   img_feats=None, encoder_history_states=None, position_ids=None):
    """modeling_transfomres.py:614-694 BertImgModel.forward -> (seq_out, pooled, attentions)."""
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    mask_add = extend_mask(attention_mask.to(sd[prefix + "pooler.dense.weight"].dtype))
    eps = cfg["layer_norm_eps"]
    h = embeddings(input_ids, token_type_ids, sd, prefix + "embeddings.", eps, position_ids)
    if encoder_history_states:
        assert img_feats is None, "Cannot take image features while using encoder history states"
    if img_feats is not None:
        h = torch.cat([h, img_embed(img_feats, sd, prefix, cfg)], dim=1)
    atts = []
    for i in range(cfg["num_hidden_layers"]):
        hist = None if encoder_history_states is None else encoder_history_states[i]
        h, p = bert_layer(h, mask_add, sd, prefix + "encoder.layer.%d." % i,
                          cfg["num_attention_heads"], eps, history_state=hist)
        atts.append(p)
    return h, pooler(h, sd, prefix + "pooler."), tuple(atts)


def seq_phase_mask(input_mask, chunk_attention_mask, t, r, phase):
    """v10:179-206.  Additive mask [N,1,S,S] for phase 1 (layers 0-2), 2 (3-8, broadcast
    [N,1,1,S]) or 3 (9-11).  input_mask [N,S] 0/1, chunk_attention_mask [N,T,T] 0/1."""
    pad = extend_mask(input_mask)                    # [N,1,1,S]
    if phase == 2:
        return pad
    s = t + r
    cm = extend_mask(chunk_attention_mask)           # [N,1,T,T]
    m = pad.repeat(1, 1, s, 1)
    m[:, :, :t, :t] = cm
    if phase == 1:
        m[:, :, t:, :t] = NEG                        # v10:183 regions do not see text
    else:
        eye = torch.eye(r, dtype=pad.dtype)
        img = torch.cat([torch.zeros(r, t, dtype=pad.dtype), eye], dim=-1)
        m[:, :, t:, :] = ((1.0 - img) * NEG)[None, None]   # v10:199-204 regions see only self
    return m


def seq_bert_img_model(sd, prefix, cfg, input_ids, token_type_ids, chunk_attention_mask,
                       input_mask, img_feats, gather_index, position_ids=None):
    """v10:275-360 SeqBertImgModel.forward + v10:171-232 CaptionBertEncoder.forward.
    -> ((seq_out, pooled, attentions), chunk_hidden_states)."""
    dt = sd[prefix + "pooler.dense.weight"].dtype
    eps = cfg["layer_norm_eps"]
    t = input_ids.shape[1]
    r = img_feats.shape[1]
    h = embeddings(input_ids, token_type_ids, sd, prefix + "embeddings.", eps, position_ids)
    h = torch.cat([h, img_embed(img_feats, sd, prefix, cfg)], dim=1)
    im = input_mask.to(dt)
    cm = chunk_attention_mask.to(dt)
    masks = {p: seq_phase_mask(im, cm, t, r, p) for p in (1, 2, 3)}
    # v10:166-168 hard-codes the 12-layer schedule [0,1,2] / [3..8] / [9,10,11].  Other depths are NOT reference
    # behaviour: the build scales it (first quarter / middle half / last quarter) for the 24-layer Oscar-large shape
    # class of BASELINE configs[4]; restated here so that shape has a checker too.
    nl = cfg["num_hidden_layers"]
    q = 3 if nl == 12 else max(1, nl // 4)
    atts = []
    chunk_hidden = None
    for i in range(nl):
        phase = 1 if i < q else (2 if i < nl - q else 3)
        if i == nl - q:
            chunk_hidden = h                        # v10:196-197 hidden ENTERING the first cross-modal layer (9 of 12)
        layer_in = h
        h, p = bert_layer(h, masks[phase], sd, prefix + "encoder.layer.%d." % i,
                          cfg["num_attention_heads"], eps,
                          gather_index=gather_index if phase == 3 else None)
        if cfg.get("add_local_residual", False) and phase == 3:
            h = h + layer_in                         # v10:212-215
        atts.append(p)
    if cfg.get("add_residual", False):
        h = h + chunk_hidden                         # v10:221-223
    return (h, pooler(h, sd, prefix + "pooler."), tuple(atts)), chunk_hidden


def cross_attention_lyx(q_in, kv, sd, prefix, num_heads=8):
    """v10:692-797 with the arguments ClsLayer_lyx passes (v10:857): no mask, tau=1, eval dropout.
    q_in [N,1,E], kv [N,L,E] -> [N,1,E]."""
    n, tq, e = q_in.shape
    d = e // num_heads
    q = _lin(q_in, sd, prefix + "q_proj") * (d ** -0.5)
    k = _lin(kv, sd, prefix + "k_proj")
    v = _lin(kv, sd, prefix + "v_proj")

    def shape(t):
        return t.view(n, -1, num_heads, d).transpose(1, 2)

    w = torch.softmax(torch.matmul(shape(q), shape(k).transpose(-1, -2)), dim=-1)
    o = torch.matmul(w, shape(v)).transpose(1, 2).reshape(n, tq, e)
    return _lin(o, sd, prefix + "out_proj")


def cls_layer_lyx(kv, cls, sd, prefix, eps):
    """v10:856-870 ClsLayer_lyx.forward: LN(attn + cls) -> BertIntermediate -> BertOutput."""
    att = cross_attention_lyx(cls[:, None, :], kv, sd, prefix + "cross_attention.")[:, 0]
    c = _ln(att + cls, sd, prefix + "LayerNorm", eps)
    return ffn(c, sd, prefix, eps)


def align_loss_fn(atts_last3, t, total_label, align_pos):
    """v10:981-987: CE applied to a softmax output (double softmax), mean over selected rows."""
    w = torch.stack(atts_last3, dim=1).sum(1).sum(1)[:, :t, t:]
    w = w.masked_fill(w == 0, -1e5)
    w = torch.softmax(w, dim=-1)
    sel = align_pos == 1
    return F.cross_entropy(w[sel], total_label[sel].to(torch.int64))


def chunkalign_ensemble(sd, prefix, cfg, input_ids, img_feat, input_mask, token_type_ids,
                        chunk_attention_mask, gather_index, align_pos=None, total_label=None):
    """v10:891-997 ChunkAlign_CLS_enc4_align_ensemble.forward -> (CLS_ensem, align_loss, extras).
    `extras` (not in the reference's return) exposes intermediates for kernel-level parity."""
    t = input_ids.shape[1]
    g_out, g_cls, _ = bert_img_model(sd, prefix + "global_enc.", cfg, input_ids, token_type_ids,
                                     input_mask, img_feat)
    (s_out, s_cls, s_atts), chunk_hidden = seq_bert_img_model(
        sd, prefix + "seq_enc.", cfg, input_ids, token_type_ids, chunk_attention_mask,
        input_mask, img_feat, gather_index)
    cls = _lin(torch.cat([g_cls, s_cls], dim=-1), sd, prefix + "cls_ensemble_1")
    kv = torch.cat([g_out[:, 1:t], s_out[:, 1:t], chunk_hidden[:, 1:t]], dim=1)
    cls0 = cls
    for i in range(2):
        cls = cls_layer_lyx(kv, cls, sd, prefix + "cls_layer_lyx.%d." % i, cfg["layer_norm_eps"])
    loss = None
    if total_label is not None:
        loss = align_loss_fn(list(s_atts[-3:]), t, total_label, align_pos)
    extras = dict(global_out=g_out, global_cls=g_cls, seq_out=s_out, seq_cls=s_cls,
                  chunk_hidden=chunk_hidden, kv=kv, cls0=cls0, seq_atts=s_atts)
    return cls, loss, extras


def cls_layer2(kv, cls, word_mask, sd, prefix, eps):
    """v10:816-837 ClsLayer2.forward (neg=False, tau=1; dropouts = identity in eval): ONE unscaled attention head of the
    projected [CLS] over align_k_proj(kv), which is both keys and values; dense + LN(+cls); BertIntermediate / BertOutput."""
    q = _lin(cls.unsqueeze(1), sd, prefix + "cls_q_proj")
    k = _lin(kv, sd, prefix + "align_k_proj")
    w = torch.softmax(q @ k.transpose(1, 2) + word_mask, dim=-1)
    out = _lin((w @ k).squeeze(1), sd, prefix + "dense")
    a = _ln(out + cls, sd, prefix + "LayerNorm", eps)
    return ffn(a, sd, prefix, eps), w


def chunkalign_enc4_align(sd, prefix, cfg, input_ids, img_feat, input_mask, token_type_ids, chunk_attention_mask,
                          gather_index, label, align_pos, total_label, num_labels=4):
    """v10:1029-1084 ChunkAlign_CLS_enc4_align.forward: both encoders WITH gradients, cls_ensemble, three ClsLayer2 over
    [global | chunk-align | chunk-hidden] text states with the word mask, binary classifier + CE, 4-way regrouping
    (binary_to_mp, v10:363-373), align loss.  -> (loss_cls_0, matched_0, align_loss)."""
    t = input_ids.shape[1]
    eps = cfg["layer_norm_eps"]
    gseq, gpool, _ = bert_img_model(sd, prefix + "global_enc.", cfg, input_ids, token_type_ids, input_mask, img_feat)
    (sseq, spool, atts), ch = seq_bert_img_model(sd, prefix + "seq_enc.", cfg, input_ids, token_type_ids,
                                                 chunk_attention_mask, input_mask, img_feat, gather_index)
    cls = _lin(torch.cat((gpool, spool), -1), sd, prefix + "cls_ensemble")
    kv = torch.cat((gseq[:, 1:t], sseq[:, 1:t], ch[:, 1:t]), dim=1)
    wm = (1.0 - input_mask[:, 1:t].to(cls.dtype).unsqueeze(1)) * -10000.0
    wm = torch.cat((wm, wm, wm), -1)
    for i in range(3):
        cls, _ = cls_layer2(kv, cls, wm, sd, prefix + "cls_layer.%d." % i, eps)
    logits = _lin(cls, sd, prefix + "classifier")
    loss = F.cross_entropy(logits.view(-1, 2), label)
    mp = torch.softmax(logits, dim=1)[:, 1].reshape(-1, num_labels)
    matched = mp.max(dim=-1)[1] == label.reshape(-1, num_labels).argmax(-1)
    return loss, matched, align_loss_fn(list(atts[-3:]), t, total_label, align_pos)


def mapping_network(x, sd, prefix):
    """modeling_ensemble.py:439-457: Dropout -> Linear(768,3840) -> Tanh -> Dropout -> Linear(3840,5120)."""
    return _lin(torch.tanh(_lin(x, sd, prefix + "1")), sd, prefix + "4")


def mc_ce(logits, label):
    """modeling_ensemble.py:528-537: CrossEntropyLoss() with float (probability) targets, mean over B."""
    return -(label * torch.log_softmax(logits, dim=-1)).sum(-1).mean()


def abstract_specific(sd, cfg, batch, roberta_fn):
    """modeling_ensemble.py:459-539 Abstract_Specific.forward.
    roberta_fn(r_ids, r_tt, r_mask, prefix_emb [N,10,1024], prompt_mask [N,10]) -> pooled [N,1024]
    stands in for the missing local_transformers RoBERTa (parity unpinned there)."""
    input_ids, img_feat, input_mask = batch["input_ids"], batch["img_feat"], batch["input_mask"]
    n = input_ids.shape[0]
    r = img_feat.shape[1]
    img_mask = torch.cat([input_mask[:, :1], input_mask[:, -r:]], dim=-1)
    with torch.no_grad():                                   # modeling_ensemble.py:466: the image-only pass never carries a gradient
        img_out, _, _ = bert_img_model(sd, "calec.global_enc.", cfg, input_ids[:, :1], None, img_mask,
                                       img_feat)
    prefix_vision = mapping_network(img_out[:, 0], sd, "mapping_network_vision.").reshape(n, 5, 1024)
    cls, align_loss, extras = chunkalign_ensemble(
        sd, "calec.", cfg, input_ids, img_feat, input_mask, batch.get("token_type_ids"),
        batch["chunk_attention_mask"], batch["gather_index"], batch.get("align_pos"),
        batch.get("total_label"))
    align_prompt = mapping_network(cls, sd, "mapping_network_alignment.").view(n, 5, 1024)
    prefix_emb = torch.cat([prefix_vision, align_prompt], dim=1)
    prompt_mask = input_mask[:, :1].repeat(1, 10)
    pooled = roberta_fn(batch.get("roberta_input_ids"), batch.get("roberta_token_type_ids"),
                        batch.get("roberta_attention_mask"), prefix_emb, prompt_mask)
    logits = _lin(pooled, sd, "abst_confidence_scorer").view(-1, 4)
    loss = None
    if batch.get("label") is not None:
        loss = mc_ce(logits, batch["label"].view(logits.shape).to(logits.dtype))
    extras = dict(extras, prefix_emb=prefix_emb, cls=cls, img_cls=img_out[:, 0], align_loss=align_loss)
    return loss, (None, None, loss, None), logits, extras


def roberta_prefix(sd, prefix, cfg, input_ids, token_type_ids, attention_mask, prompt_embeddings, input_mask):
    """CPU restatement of modeling/roberta_prefix.py::RobertaPrefixModel -- NOT of reference code: the reference's
    prefix RoBERTa (local_transformers, modeling_ensemble.py:501-503) is absent from the reference tree (SURVEY 8c,
    parity unpinned), so this pins the build's own documented splice: embeddings = LN(word + pos + type) with
    RoBERTa position ids, prompt embeddings inserted after <s> (after the LN), mask extended at the same place,
    BERT layer arithmetic (a_bert:238-451), pooler tanh(dense(h[:, 0])).  Returns (sequence_output, pooled)."""
    pad = cfg.get("pad_token_id", 1)
    eps = cfg["layer_norm_eps"]
    nonpad = (input_ids != pad).to(torch.int64)
    pos = torch.cumsum(nonpad, dim=1) * nonpad + pad
    # (RobertaEmbeddings: word and position tables are nn.Embedding(..., padding_idx=pad): no gradient for those rows)
    e = (torch.nn.functional.embedding(input_ids, sd[prefix + "embeddings.word_embeddings.weight"], padding_idx=pad)
         + torch.nn.functional.embedding(pos, sd[prefix + "embeddings.position_embeddings.weight"], padding_idx=pad)
         + sd[prefix + "embeddings.token_type_embeddings.weight"][token_type_ids])
    e = _ln(e, sd, prefix + "embeddings.LayerNorm", eps)
    mask = attention_mask.to(torch.float32)
    if prompt_embeddings is not None:
        e = torch.cat([e[:, :1], prompt_embeddings, e[:, 1:]], dim=1)
        mask = torch.cat([mask[:, :1], input_mask.to(torch.float32), mask[:, 1:]], dim=1)
    h = e
    for i in range(cfg["num_hidden_layers"]):
        h, _ = bert_layer(h, extend_mask(mask), sd, prefix + "encoder.layer.%d." % i, cfg["num_attention_heads"], eps)
    return h, pooler(h, sd, prefix + "pooler.")


# ---- train-step arithmetic ---------------------------------------------------------------------
# The reference constructs `transformers.AdamW(grouped_parameters, lr=..., eps=args.adam_epsilon)` (run_PMR_ModCR.py:24,137)
# from the transformers 4.x line it vendors alongside a_transformers.zip.  That class is a third-party dependency that
# is neither under /root/reference nor in the installed transformers (5.15 removed it), so its PUBLISHED algorithm is
# restated here -- transformers v4.x src/transformers/optimization.py::AdamW.step, defaults betas=(0.9, 0.999),
# weight_decay=0.0, correct_bias=True:
#     exp_avg    = b1 * exp_avg    + (1 - b1) * grad
#     exp_avg_sq = b2 * exp_avg_sq + (1 - b2) * grad^2
#     denom      = sqrt(exp_avg_sq) + eps                  (eps OUTSIDE any bias correction)
#     step_size  = lr * sqrt(1 - b2^t) / (1 - b1^t)
#     p         -= step_size * exp_avg / denom
#     p         -= lr * weight_decay * p                   (after the update; weight_decay is 0 in the reference)
# Anchors in the reference: the constructor call (:137), the two lr groups (:127-136), the schedules (:138-145),
# clip_grad_norm_ on every micro-batch (:216), optimizer.step(); scheduler.step() (:224-225).
def hf_adamw_step(p, grad, state, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
    """one transformers.AdamW update of tensor p (in place) with its state dict {'step','exp_avg','exp_avg_sq'}"""
    if not state:
        state["step"], state["exp_avg"], state["exp_avg_sq"] = 0, torch.zeros_like(p), torch.zeros_like(p)
    b1, b2 = betas
    state["step"] += 1
    state["exp_avg"] = b1 * state["exp_avg"] + (1.0 - b1) * grad
    state["exp_avg_sq"] = b2 * state["exp_avg_sq"] + (1.0 - b2) * grad * grad
    denom = state["exp_avg_sq"].sqrt() + eps
    step_size = lr
    if correct_bias:
        step_size = step_size * math.sqrt(1.0 - b2 ** state["step"]) / (1.0 - b1 ** state["step"])
    p -= step_size * state["exp_avg"] / denom
    if weight_decay > 0.0:
        p -= lr * weight_decay * p
    return p


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (run_PMR_ModCR.py:216): total 2-norm over all gradients,
    coefficient min(1, max_norm / (total + 1e-6)); returns (scaled grads, total norm)."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    coef = min(1.0, max_norm / (total + 1e-6))
    return [g * coef for g in grads], total


def linear_schedule(step, t_total, warmup_steps=0):
    """transformers.get_linear_schedule_with_warmup (run_PMR_ModCR.py:142-143): lr multiplier after `step` steps"""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1, t_total - warmup_steps)))


def constant_schedule(step, warmup_steps=0):
    """transformers.get_constant_schedule_with_warmup (run_PMR_ModCR.py:139-140)"""
    if step < warmup_steps:
        return float(step) / float(max(1.0, warmup_steps))
    return 1.0


def train_steps_hf(params, names, grads_per_step, learning_rate, adam_epsilon, t_total, max_grad_norm=1.0,
                   scheduler="linear", warmup_steps=0):
    """The reference's optimisation loop on given gradients (run_PMR_ModCR.py:127-145,216,224-225): two lr groups
    ('seq_enc' x 0.1), clip, transformers.AdamW, schedule.  params: list of fp32 tensors (updated in place);
    grads_per_step: list (one entry per step) of lists of gradients."""
    states = [dict() for _ in params]
    for t, grads in enumerate(grads_per_step):
        grads, _ = clip_grad_norm(grads, max_grad_norm)
        f = linear_schedule(t, t_total, warmup_steps) if scheduler == "linear" else constant_schedule(t, warmup_steps)
        for p, n, g, st in zip(params, names, grads, states):
            hf_adamw_step(p, g, st, learning_rate * (0.1 if "seq_enc" in n else 1.0) * f, eps=adam_epsilon)
    return params
