"""The two Oscar encoders WITH gradients (`ChunkAlign_CLS_enc4_align`, v10:1016-1084, runs global_enc
and seq_enc outside torch.no_grad(); BASELINE config 3 "full fwd+bwd").  Same arithmetic and the same C-ABI forward
entries as the frozen path (modeling_transfomres.py / modeling_vcr_chunkalign_v10.py in this package); what changes is
that every block is a torch.autograd.Function whose backward is the matching C-ABI backward entry:

    embeddings   word + position + type lookups (torch gathers; backward = modcr_embedding_bwd, sorted-segment sums),
                 LayerNorm = modcr_layernorm_fwd / _bwd
    regions      modcr_cast_pad, img_embedding through LinearFn (MFMA GEMM over the 64-padded feature dim),
                 LayerNorm as above
    layers       BertLayerFn: modcr_qkv_attn_fwd / modcr_qkv_attn_bwd with the padding mask (global_enc, seq_enc
                 layers 3-8), the phase-1 bit mask (seq_enc 0-2) or the phase-3 bit mask + chunk-mean queries
                 (seq_enc 9-11), then the projection / FFN composites
    pooler       LinearFn with the tanh epilogue on the [CLS] row

The align map (sum of the last three seq_enc layers' text->region probabilities over all heads) is a differentiable
output here (hip_autograd.AlignMapFn): the align loss of v10:1067-1073 back-propagates through the attention
probabilities into both the layers above and below (modcr_qkv_attn_dropout_bwd's d_align input).  The ModCR objective
itself uses the MC-CE only (run_PMR_ModCR.py:206-215).
"""
import torch
import torch.nn.functional as F

import modcr_hip as mh
from . import hip_autograd as ag
from . import hip_layers
from .bert_primitives import PackCache, _pad64, compute_dtype


def _layer_params(layer):
    named = dict(layer.named_parameters())
    return [named[k] for k in ag.BertLayerFn.NAMES]


def _packed(model, i, layer, params, device, dtype):
    cache = model.__dict__.setdefault("_train_cache", PackCache())
    return cache.get(("train_layer", i, dtype), params,
                     lambda: hip_layers.pack_layer(dict(zip(ag.BertLayerFn.NAMES, params)), "", device, dtype))


def embed(model, input_ids, token_type_ids, position_ids, img_feats):
    """BertEmbeddings (a_bert:195-211) ++ img_embedding + LayerNorm (modeling_transfomres.py:676-684), dropout of the
    concatenation in training mode; returns [N, T+R, H] in the storage dtype with a grad_fn."""
    cfg = model.config
    emb = model.embeddings
    n, t = input_ids.shape
    h = cfg.hidden_size
    dt = compute_dtype(cfg)
    if position_ids is None:
        position_ids = emb.position_ids[:, :t]
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    e = ag.EmbeddingSumFn.apply(input_ids, position_ids, token_type_ids, emb.word_embeddings.weight, emb.position_embeddings.weight,
                                emb.token_type_embeddings.weight, emb.word_embeddings.padding_idx, emb.position_embeddings.padding_idx)
    e = ag.LayerNormFn.apply(e.reshape(n * t, h), None, emb.LayerNorm.weight, emb.LayerNorm.bias, emb.eps).view(n, t, h)
    if img_feats is not None:
        if not model.use_img_layernorm:
            raise NotImplementedError("use_img_layernorm=False: the Oscar checkpoints ModCR loads set it (run_PMR_ModCR.py:720)")
        r, d = img_feats.shape[1], img_feats.shape[2]
        kp = _pad64(d) if dt == torch.bfloat16 else d
        src = mh.cast_pad(img_feats, kp, mh.BF16 if dt == torch.bfloat16 else mh.F32)
        w = model.img_embedding.weight
        if kp != d:
            w = F.pad(w, (0, kp - d))                      # zero columns; the gradient is sliced back by autograd
        v = ag.linear(src, w, model.img_embedding.bias)
        v = ag.LayerNormFn.apply(v, None, model.LayerNorm.weight, model.LayerNorm.bias, cfg.img_layer_norm_eps)
        e = torch.cat((e, v.view(n, r, h)), dim=1)
    e = ag.dropout(e.contiguous(), model.dropout.p, model.training)
    return e if dt == torch.float32 else ag.ToBf16Fn.apply(e)


def pool(model, hidden):
    """BertPooler (a_bert:634-646) on the [CLS] row, fp32 with a grad_fn"""
    cls = hidden[:, 0].contiguous()
    if cls.dtype != torch.float32:
        cls = ag.ToF32Fn.apply(cls)
    return ag.linear(cls, model.pooler.dense.weight, model.pooler.dense.bias, act=mh.ACT_TANH)


def _run_layer(model, i, layer, hidden, key_mask=None, mask_bits=None, chunk_id=None, align=None):
    """align = (map buffer [N,T,R], T, gradient holder) for the layers whose probabilities feed the align map"""
    cfg = model.config
    params = _layer_params(layer)
    p = cfg.hidden_dropout_prob if model.training else 0.0
    ap = getattr(cfg, "attention_probs_dropout_prob", 0.0) if model.training else 0.0
    packed = _packed(model, i, layer, params, hidden.device, hidden.dtype)
    if align is not None:
        # (a copy: the cached dict itself stays clean)
        packed = dict(packed, _align=align, _side_post=bool(getattr(cfg, "modcr_align_map_post_dropout", True)))
    return ag.BertLayerFn.apply(hidden, key_mask, mask_bits, chunk_id, cfg.num_attention_heads, cfg.layer_norm_eps, float(p), float(ap),
                                packed, *params)


def global_encoder(model, input_ids, token_type_ids, attention_mask, position_ids, img_feats):
    """BertImgModel.forward with gradients: (sequence_output, pooled_output)"""
    hidden = embed(model, input_ids, token_type_ids, position_ids, img_feats)
    mask = attention_mask.to(torch.float32).contiguous()
    for i, layer in enumerate(model.encoder.layer):
        hidden = _run_layer(model, i, layer, hidden, key_mask=mask)
    return hidden, pool(model, hidden)


def seq_encoder(model, input_ids, token_type_ids, chunk_mask, input_mask, position_ids, img_feats, chunk_id, want_align_map=True):
    """SeqBertImgModel.forward with gradients: ((sequence_output, pooled_output), chunk_hidden_states, align_map).  The
    three mask phases are those of CaptionBertEncoder.hip_forward (v10:153-232).  align_map [N,T,R] = head-summed text ->
    region probabilities of layers 9-11 (v10:982 / :1067), differentiable: its gradient re-enters those layers' attention
    backward (the align loss of ChunkAlign_CLS_enc4_align, v10:1067-1073)."""
    enc = model.encoder
    hidden = embed(model, input_ids, token_type_ids, position_ids, img_feats)
    im = input_mask.to(torch.float32).contiguous()
    cm = chunk_mask.to(torch.float32).contiguous()
    bits1 = mh.build_phase_mask(im, cm, 1)
    bits3 = None
    chunk_hidden_states = None
    n, t = input_ids.shape
    r = hidden.shape[1] - t
    align = None
    if want_align_map and r > 0:
        align = (torch.zeros((n, t, r), dtype=torch.float32, device=hidden.device), t, {})
    for i, layer in enumerate(enc.layer):
        if i in enc.cross_modal_layers:
            if i == enc.cross_modal_layers[0]:
                chunk_hidden_states = hidden
                bits3 = mh.build_phase_mask(im, cm, 3)
            former = hidden
            hidden = _run_layer(model, i, layer, hidden, mask_bits=bits3, chunk_id=chunk_id, align=align)
            if enc.add_local_residual:                      # v10:212-215
                hidden = ag.ResidualAddFn.apply(hidden, former)
        elif i >= enc.cross_chunk_attention_layers[0]:
            hidden = _run_layer(model, i, layer, hidden, key_mask=im)
        else:
            hidden = _run_layer(model, i, layer, hidden, mask_bits=bits1)
    if enc.add_residual:                                    # v10:221-223
        hidden = ag.ResidualAddFn.apply(hidden, chunk_hidden_states)
    amap = None
    if align is not None:
        hidden, amap = ag.AlignMapFn.apply(hidden, align[0], align[2])
    return (hidden, pool(model, hidden)), chunk_hidden_states, amap
