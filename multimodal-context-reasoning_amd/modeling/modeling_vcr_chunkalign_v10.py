"""Drop-in for the hot-path classes of the reference's modeling/modeling_vcr_chunkalign_v10.py:

  CaptionBertSelfAttention / Attention / Layer / Encoder (:45-232)   seq_enc with the three mask
  SeqBertImgModel (:235-360)                                          phases + chunk-mean queries
  cross_attention_lyx (:660-797), ClsLayer2 (:801-837), ClsLayer_lyx (:840-870)
  ChunkAlign_CLS_enc4_align_ensemble (:872-997)

Same constructor / forward signatures, positional return tuples and state-dict keys; all
arithmetic in libmodcr_hip.  The reference's per-sample Python loop (v10:69-77), the three
materialised [N,1,S,S] fp32 masks (v10:179-206) and the retained [N,12,S,S] probability tensors
(config.output_attentions) are replaced by: a packed chunk-id tensor, bit masks built once per
batch, and a head/layer-summed text->image map accumulated inside the attention kernel.
"""
import torch
from torch import nn
from torch.nn.utils.rnn import pad_sequence

import modcr_hip as mh
from . import hip_autograd as ag
from .bert_primitives import (BertEmbeddings, BertIntermediate, BertOutput, BertPooler, BertPreTrainedModel,
                              BertSelfOutput, EncoderOutputs, PackCache, additive_to_binary, compute_dtype)
from .hip_layers import Workspace
from .modeling_bert import CaptionBertSelfAttention as _GlobalSelfAttention, split_additive_mask
from .modeling_transfomres import CaptionBertLayer as _GlobalLayer, ImgEmbedMixin

BertLayerNorm = torch.nn.LayerNorm


def pack_chunk_ids(gather_index, t, device):
    """list[N] of int64 [len_n] (Data/VCRChunkAlign.py:666-670) -> int32 [N,T]: chunk id of text
    token t (gather_index[n][t-1] for t in 1..len_n), -1 elsewhere (v10:70-77 touches only those rows)."""
    n = len(gather_index)
    cid = torch.full((n, t), -1, dtype=torch.int32, device=device)
    gi = [g.to(device=device, dtype=torch.int32) for g in gather_index]
    if n and max(g.numel() for g in gi) > 0:
        padded = pad_sequence(gi, batch_first=True, padding_value=-1)
        cid[:, 1:1 + padded.shape[1]] = padded[:, :t - 1]
    return cid


class CaptionBertSelfAttention(_GlobalSelfAttention):
    """v10:45-107: adds do_chunk_cross (query of every text token := mean query of its chunk)."""

    def __init__(self, config):
        super(CaptionBertSelfAttention, self).__init__(config)
        self.hidden_size = config.hidden_size

    def forward(self, hidden_states, attention_mask, head_mask=None, history_state=None, do_chunk_cross=False,
                offsets=None, gather_index=None):
        if head_mask is not None:
            raise NotImplementedError("head_mask is always None on the ModCR path")
        n, s, _ = hidden_states.shape
        l = s + (0 if history_state is None else history_state.shape[1])
        km, bits = split_additive_mask(attention_mask, n, s, l)
        cid = None
        if do_chunk_cross:
            t = 1 + max(g.numel() for g in gather_index) + 1
            cid = pack_chunk_ids(gather_index, min(t, s), hidden_states.device)
        ctx, probs = self.hip_forward(hidden_states, km, bits, hist=history_state, chunk_id=cid,
                                      want_probs=self.output_attentions)
        return (ctx, probs) if self.output_attentions else (ctx,)


class CaptionBertAttention(nn.Module):
    def __init__(self, config):
        super(CaptionBertAttention, self).__init__()
        self.self = CaptionBertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask, head_mask=None, history_state=None, do_chunk_cross=False,
                offsets=None, gather_index=None):
        self_outputs = self.self(input_tensor, attention_mask, head_mask=head_mask, history_state=history_state,
                                 do_chunk_cross=do_chunk_cross, offsets=offsets, gather_index=gather_index)
        attention_output = self.output(self_outputs[0], input_tensor)
        return (attention_output,) + self_outputs[1:]


class CaptionBertLayer(_GlobalLayer):
    def __init__(self, config):
        nn.Module.__init__(self)
        self.attention = CaptionBertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask, head_mask=None, history_state=None, do_chunk_cross=False,
                offsets=None, gather_index=None):
        attention_outputs = self.attention(hidden_states, attention_mask, head_mask=head_mask,
                                           history_state=history_state, do_chunk_cross=do_chunk_cross,
                                           offsets=offsets, gather_index=gather_index)
        attention_output = attention_outputs[0]
        layer_output = self.output(self.intermediate(attention_output), attention_output)
        return (layer_output,) + attention_outputs[1:]


class CaptionBertEncoder(nn.Module):
    """v10:153-232.  Layers 0-2: chunk-local text mask; 3-8: plain padding mask; 9-11: chunk mask,
    regions see only themselves, chunk-mean queries.  Returns (outputs, chunk_hidden_states)."""

    def __init__(self, config):
        super(CaptionBertEncoder, self).__init__()
        self.output_attentions = config.output_attentions
        self.output_hidden_states = config.output_hidden_states
        self.materialize = getattr(config, "modcr_materialize_attentions", False)
        self.layer = nn.ModuleList([CaptionBertLayer(config) for _ in range(config.num_hidden_layers)])
        self.num_hidden_layers = config.num_hidden_layers
        self.add_residual = config.add_residual
        self.add_local_residual = config.add_local_residual
        self.chunk_attention_layers = [0, 1, 2, ]
        self.cross_chunk_attention_layers = [3, 4, 5, 6, 7, 8]
        self.cross_modal_layers = [9, 10, 11]
        if config.num_hidden_layers != 12:
            # The reference hard-codes the 12-layer schedule above (v10:166-168).  For other depths (the 24-layer
            # Oscar-large shape class of BASELINE configs[4]; small test models) this build scales it: first quarter
            # chunk-local, middle half plain padding mask, last quarter cross-modal with chunk-mean queries.
            q = max(1, config.num_hidden_layers // 4)
            nl = config.num_hidden_layers
            self.chunk_attention_layers = list(range(0, q))
            self.cross_chunk_attention_layers = list(range(q, nl - q))
            self.cross_modal_layers = list(range(nl - q, nl))
        self.max_hypo = config.max_hypo

    def hip_forward(self, x, input_mask, chunk_mask, chunk_id, hypo_len, img_len, encoder_history_states=None,
                    want_align_map=True, ws=None, last_rows=None):
        """x [N,S,H]; input_mask [N,S] 0/1; chunk_mask [N,T,T] 0/1; chunk_id int32 [N,T].  last_rows = k (opt-in): the final hidden
        states come back as [N,k,H] (CaptionBertLayer.hip_forward's out_rows on the last layer); chunk_hidden_states and the align
        map are unaffected."""
        ws = ws or Workspace()
        n = x.shape[0]
        if (self.output_attentions and self.materialize) or self.output_hidden_states or encoder_history_states is not None:
            last_rows = None
        nl = len(self.layer)
        bits1 = mh.build_phase_mask(input_mask, chunk_mask, 1)
        bits3 = None
        amap = None
        want = self.output_attentions and self.materialize
        all_hidden, all_att = (), ()
        chunk_hidden_states = None
        for i, layer in enumerate(self.layer):
            if self.output_hidden_states:
                all_hidden = all_hidden + (x,)
            hist = None if encoder_history_states is None else encoder_history_states[i]
            if i in self.cross_modal_layers:
                if i == self.cross_modal_layers[0]:
                    chunk_hidden_states = x                       # v10:196-197
                    bits3 = mh.build_phase_mask(input_mask, chunk_mask, 3)
                    if want_align_map and img_len > 0:
                        amap = torch.zeros((n, hypo_len, img_len), dtype=torch.float32, device=x.device)
                former = x
                k_out = last_rows if i == nl - 1 else None
                x, probs = layer.hip_forward(x, mask_bits=bits3, hist=hist, chunk_id=chunk_id, want_probs=want,
                                             align_map=amap, align_t=hypo_len if amap is not None else 0, ws=ws, out_rows=k_out)
                if self.add_local_residual:                       # v10:212-215: the cross-modal layers add their input
                    if k_out is not None and k_out < former.shape[1]:
                        former = former[:, :k_out].contiguous()
                    x = mh.add(mh.convert(x, mh.F32), former, out_dtype=mh.dt_of(former))
            elif i not in self.chunk_attention_layers:
                x, probs = layer.hip_forward(x, key_mask=input_mask, hist=hist, want_probs=want, ws=ws,
                                             out_rows=last_rows if i == nl - 1 else None)
            else:
                x, probs = layer.hip_forward(x, mask_bits=bits1, hist=hist, want_probs=want, ws=ws,
                                             out_rows=last_rows if i == nl - 1 else None)
            if self.output_attentions:
                all_att = all_att + (probs,)
        if self.add_residual:                                     # v10:221-223: + the hidden states that entered layer 9
            chs = chunk_hidden_states if x.shape[1] == chunk_hidden_states.shape[1] else chunk_hidden_states[:, :x.shape[1]].contiguous()
            x = mh.add(mh.convert(x, mh.F32), chs, out_dtype=mh.dt_of(x))
        if self.output_hidden_states:
            all_hidden = all_hidden + (x,)
        outputs = (x,)
        if self.output_hidden_states:
            outputs = outputs + (all_hidden,)
        if self.output_attentions:
            outputs = outputs + (all_att,)
        outputs = EncoderOutputs(outputs)
        outputs.align_map = amap
        return outputs, chunk_hidden_states

    def forward(self, hidden_states, chunk_attention_mask, gather_index, img_mask, input_mask, hypo_len, img_len,
                head_mask=None, encoder_history_states=None, offsets=None):
        """Reference signature: the three masks arrive ADDITIVE and extended (v10:296-314)."""
        n = hidden_states.shape[0]
        im = additive_to_binary(input_mask.reshape(n, hypo_len + img_len))
        cm = additive_to_binary(chunk_attention_mask.reshape(n, hypo_len, hypo_len))
        cid = pack_chunk_ids(gather_index, hypo_len, hidden_states.device)
        return self.hip_forward(hidden_states, im, cm, cid, hypo_len, img_len, encoder_history_states)


class SeqBertImgModel(BertPreTrainedModel, ImgEmbedMixin):
    """ Expand from BertModel to handle image region features as input (seq_enc). """

    def __init__(self, config):
        super(SeqBertImgModel, self).__init__(config)
        self.embeddings = BertEmbeddings(config)
        self.encoder = CaptionBertEncoder(config)
        self.pooler = BertPooler(config)
        self.img_dim = config.img_feature_dim
        self.img_feature_type = config.img_feature_type
        self.use_img_layernorm = getattr(config, "use_img_layernorm", None)
        self.img_embedding = nn.Linear(self.img_dim, self.config.hidden_size, bias=True)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        if self.use_img_layernorm:
            self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.img_layer_norm_eps)
        self._cache = PackCache()
        self._ws = Workspace()
        self.init_weights()
        self.max_hypo = config.max_hypo
        self.edge_dense = nn.Embedding(1, config.hidden_size)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, input_mask=None, position_ids=None,
                head_mask=None, img_feats=None, img_mask=None, encoder_history_states=None, offsets=None,
                gather_index=None, modcr_last_rows=None):
        """attention_mask = chunk_attention_mask [N,T,T] 0/1, input_mask [N,T+R] 0/1 (v10:903-907).
        Returns ((sequence_output, pooled_output, attentions), chunk_hidden_states).  modcr_last_rows = k (not in the reference's
        signature; frozen route only): sequence_output is [N,k,H], see BertImgModel.forward."""
        if head_mask is not None:
            raise NotImplementedError("head_mask is never set on the ModCR path")
        if attention_mask is None or attention_mask.dim() != 3:
            raise NotImplementedError          # v10:289-294: seq_enc is always driven with the 3-D chunk mask
        if input_mask is None or input_mask.dim() != 2:
            raise NotImplementedError
        if encoder_history_states:
            assert img_feats is None, "Cannot take image features while using encoder history states"
        n, t = input_ids.shape
        r = img_feats.shape[1]
        if getattr(self, "trainable", False) and torch.is_grad_enabled() and not encoder_history_states:
            # seq_enc with gradients (v10:1016-1084): structured masks and chunk-mean queries through modcr_qkv_attn_bwd
            from . import trainable_encoders
            cid = gather_index if torch.is_tensor(gather_index) else pack_chunk_ids(gather_index, t, input_ids.device)
            (seq, pooled), chunk_hidden_states, amap = trainable_encoders.seq_encoder(
                self, input_ids, token_type_ids, attention_mask, input_mask, position_ids, img_feats, cid)
            att = ((None,) * len(self.encoder.layer),) if self.encoder.output_attentions else ()
            outputs = EncoderOutputs((seq, pooled) + att)
            outputs.align_map = amap                        # differentiable (hip_autograd.AlignMapFn)
            return outputs, chunk_hidden_states
        dt = compute_dtype(self.config)
        x = torch.empty((n, t + r, self.config.hidden_size), dtype=dt, device=input_ids.device)
        drop = None
        if self.training and self.dropout.p > 0.0:          # BertEmbeddings.dropout (a_bert:210) and self.dropout (v10:343): same p,
            drop = (self.dropout.p,) + tuple(mh.DROPOUT.take(x.numel()))      # one mask over x, applied by the passes that write its rows
        self.embeddings(input_ids, token_type_ids, position_ids, out=x, dropout=drop)
        self.embed_regions(img_feats, x, t, dropout=drop)
        cid = gather_index if torch.is_tensor(gather_index) else pack_chunk_ids(gather_index, t, input_ids.device)
        encoder_outputs, chunk_hidden_states = self.encoder.hip_forward(
            x, input_mask.to(torch.float32), attention_mask.to(torch.float32), cid, t, r, encoder_history_states,
            ws=self._ws, last_rows=modcr_last_rows)
        sequence_output = encoder_outputs[0]
        pooled_output = self.pooler(sequence_output)
        outputs = EncoderOutputs((sequence_output, pooled_output,) + tuple(encoder_outputs[1:]))
        outputs.align_map = encoder_outputs.align_map
        return outputs, chunk_hidden_states


class cross_attention_lyx(nn.Module):
    """Multi-headed attention from 'Attention Is All You Need' paper -- the form ClsLayer_lyx uses
    (v10:857): one query token per sequence, keys/values = key_value_states, no mask, tau = 1."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, is_decoder=False, bias=True):
        super(cross_attention_lyx, self).__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.dropout = dropout
        self.head_dim = embed_dim // num_heads
        if (self.head_dim * num_heads) != self.embed_dim:
            raise ValueError("embed_dim must be divisible by num_heads (got `embed_dim`: %d and `num_heads`: %d)."
                             % (self.embed_dim, num_heads))
        self.scaling = self.head_dim ** -0.5
        self.is_decoder = is_decoder
        self.k_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.v_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)

    def forward(self, hidden_states, key_value_states=None, past_key_value=None, attention_mask=None,
                layer_head_mask=None, output_attentions=False, neg_type=False, tau=1.0, prior_score=None):
        """hidden_states [N,1,E] fp32; key_value_states [N,L,E].  Returns (attn_output [N,1,E], None, None)."""
        if (key_value_states is None or past_key_value is not None or attention_mask is not None or
                layer_head_mask is not None or neg_type or tau != 1.0 or prior_score is not None):
            raise NotImplementedError("only the call pattern of ClsLayer_lyx.forward (v10:857) is on the ModCR path")
        n, tgt_len, e = hidden_states.shape
        if tgt_len != 1:
            raise NotImplementedError("tgt_len must be 1 (CLS query)")
        if isinstance(key_value_states, (tuple, list)):
            # row blocks of FROZEN bf16 states (ChunkAlign_CLS_enc4_align_ensemble.forward hands the three text-row views of
            # v10:913 over without concatenating them): the reassociated form, keys and values are never projected
            return self._forward_frozen_rows(hidden_states.reshape(n, e), list(key_value_states)), None, None
        if self.takes_row_blocks([key_value_states]):
            return self._forward_frozen_rows(hidden_states.reshape(n, e), [key_value_states]), None, None
        l = key_value_states.shape[1]
        kv2 = key_value_states.reshape(n * l, e)
        kvd = mh.dt_of(kv2)
        q = ag.linear(hidden_states.reshape(n, e), self.q_proj.weight, self.q_proj.bias)
        k = ag.linear(kv2, self.k_proj.weight, self.k_proj.bias, out_dtype=kvd).view(n, l, e)
        v = ag.linear(kv2, self.v_proj.weight, self.v_proj.bias, out_dtype=kvd).view(n, l, e)
        att = ag.AlignAttnFn.apply(q, k, v, self.num_heads, self.scaling, float(self.dropout) if self.training else 0.0)
        out = ag.linear(att, self.out_proj.weight, self.out_proj.bias)
        return out.view(n, 1, e), None, None


    def takes_row_blocks(self, blocks):
        """True when _forward_frozen_rows serves these states: bf16, no gradient wanted, 8 heads, E <= 1024 (csrc/clsattn.hip)"""
        return (not ag.EXACT and self.num_heads == 8 and self.embed_dim <= 1024 and self.embed_dim % 64 == 0 and
                all(t.dtype == torch.bfloat16 and not t.requires_grad and t.stride(2) == 1 and t.stride(1) == blocks[0].stride(1)
                    and t.data_ptr() % 8 == 0 and t.stride(0) % 4 == 0 for t in blocks))

    def _forward_frozen_rows(self, cls, blocks):
        """score[h][j] = q_h.(Wk_h x_j + bk_h) = (Wk_h^T q_h).x_j + const_h; out_h = Wv_h (sum_j p'[h][j] x_j) + bv_h sum_j p'[h][j]:
        two few-row GEMMs on [N*heads, E] rows around modcr_cls_xattn_fwd instead of two [N*L, E] projections (and, in the
        backward, instead of their K = N*L weight-gradient products).  Same function of the parameters as v10:741-795."""
        n, e = cls.shape
        h, d = self.num_heads, self.head_dim
        hm = getattr(self, "_head_rows", None)
        if hm is None or hm.device != cls.device:
            # [h, E]: row h = the reference's q scaling (v10:751) on head h's slice, 0 elsewhere
            hm = (torch.arange(e, device=cls.device).div(d, rounding_mode="floor")[None, :] ==
                  torch.arange(h, device=cls.device)[:, None]).to(torch.float32) * self.scaling
            self._head_rows = hm
        q = ag.linear(cls, self.q_proj.weight, self.q_proj.bias)
        qe = (q.view(n, 1, e) * hm).view(n * h, e)                                   # q_h in its own slice, one row per head
        qt = ag.linear(qe, self.k_proj.weight.t().contiguous(), None)               # [N*h, E]: row (n, h) = Wk_h^T q_h
        ctx, ssum = ag.ClsXAttnFn.apply(qt.view(n, h, e), self.k_proj.bias, h, float(self.dropout) if self.training else 0.0,
                                        *blocks)
        av = ag.linear(ctx.view(n * h, e), self.v_proj.weight, None)                # row (n, h) = Wv ctx_h; its slice h is wanted
        att = torch.diagonal(av.view(n, h, h, d), dim1=1, dim2=2).permute(0, 2, 1)  # [N, h, d]
        att = att + ssum.unsqueeze(-1) * self.v_proj.bias.view(h, d)
        out = ag.linear(att.reshape(n, e), self.out_proj.weight, self.out_proj.bias)
        return out.view(n, 1, e)


class _BertLayerParams(nn.Module):
    """BertLayer members that ClsLayer2 / ClsLayer_lyx inherit but never call: kept so the state
    dict has the reference's keys (cls_layer*.{i}.attention.*)."""

    def __init__(self, config):
        super(_BertLayerParams, self).__init__()
        self.self = _GlobalSelfAttention(config)
        self.output = BertSelfOutput(config)


class ClsLayer2(nn.Module):
    """v10:801-837: ONE unscaled attention head of the projected [CLS] over align_k_proj(states) (keys AND values) with
    the additive word mask, dense + LN(+cls), BertIntermediate, BertOutput.  Constructed but never called by
    ChunkAlign_CLS_enc4_align_ensemble (v10:882); the layer of ChunkAlign_CLS_enc4_align (v10:1025, :1052-1055).
    The attention is modcr_align_attn_fwd/bwd with heads = 1, scale = 1, k == v and the mask as its key bias."""

    def __init__(self, config):
        super(ClsLayer2, self).__init__()
        self.attention = _BertLayerParams(config)
        self.cls_q_proj = nn.Linear(config.hidden_size, config.hidden_size)
        self.align_k_proj = nn.Linear(config.hidden_size, config.hidden_size)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        self.eps = config.layer_norm_eps

    def forward(self, self_chunk_align, cls, word_mask, neg=False, tau=1.0):
        """self_chunk_align [N,L,H] (storage dtype), cls [N,H] fp32, word_mask additive [N,1,L] -> (layer_output [N,H],
        None): the attention weights are not materialised (the caller at v10:1055 drops them)."""
        if neg or tau != 1.0:
            raise NotImplementedError("ClsLayer2: neg / tau are never set by ChunkAlign_CLS_enc4_align (v10:1055)")
        if self_chunk_align.dtype == torch.float32:
            ag.set_exact(True)          # fp32 encoder states = parity mode
        n, l, e = self_chunk_align.shape
        kv2 = self_chunk_align.reshape(n * l, e)
        q = ag.linear(cls, self.cls_q_proj.weight, self.cls_q_proj.bias)
        k = ag.linear(kv2, self.align_k_proj.weight, self.align_k_proj.bias, out_dtype=mh.dt_of(kv2)).view(n, l, e)
        p = float(self.dropout.p) if self.training else 0.0
        att = ag.AlignAttnFn.apply(q, k, k, 1, 1.0, p, word_mask.reshape(n, l).to(torch.float32).contiguous())   # v10:822-829
        out = ag.linear(att, self.dense.weight, self.dense.bias)
        out = ag.dropout(out, self.dropout.p, self.training)                    # v10:832
        c = ag.LayerNormFn.apply(out, cls, self.LayerNorm.weight, self.LayerNorm.bias, self.eps)
        inter = ag.linear(c, self.intermediate.dense.weight, self.intermediate.dense.bias, act=mh.ACT_GELU)
        o = ag.linear(inter, self.output.dense.weight, self.output.dense.bias)
        o = ag.dropout(o, self.output.dropout.p, self.training)
        return ag.LayerNormFn.apply(o, c, self.output.LayerNorm.weight, self.output.LayerNorm.bias, self.eps), None


class ClsLayer_lyx(nn.Module):
    """v10:840-870: LN(cross_attention(cls, kv) + cls) -> BertIntermediate -> BertOutput, trainable."""

    def __init__(self, config):
        super(ClsLayer_lyx, self).__init__()
        self.attention = _BertLayerParams(config)
        self.ensemble = nn.Linear(config.hidden_size * 2, 1)
        self.cross_attention = cross_attention_lyx(config.hidden_size, 8, dropout=0.1, is_decoder=True)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        self.eps = config.layer_norm_eps

    def forward(self, self_chunk_align, cls, word_mask=None, prior_score=None, cls_2=None):
        """self_chunk_align: the reference's [N, L, H] tensor, or a tuple of [N, rows, H] row blocks standing for their
        concatenation along dim 1 (frozen bf16 states only: cross_attention_lyx.takes_row_blocks)"""
        if not isinstance(self_chunk_align, (tuple, list)) and self_chunk_align.dtype == torch.float32:
            ag.set_exact(True)          # fp32 encoder states = parity mode
        att = self.cross_attention(cls.unsqueeze(1), self_chunk_align, tau=1.0, neg_type=False,
                                   prior_score=prior_score)[0].squeeze(1)
        att = ag.dropout(att, self.dropout.p, self.training)                   # v10:861
        c = ag.LayerNormFn.apply(att, cls, self.LayerNorm.weight, self.LayerNorm.bias, self.eps)
        inter = ag.linear(c, self.intermediate.dense.weight, self.intermediate.dense.bias, act=mh.ACT_GELU)
        o = ag.linear(inter, self.output.dense.weight, self.output.dense.bias)
        o = ag.dropout(o, self.output.dropout.p, self.training)                # BertOutput.dropout (a_bert:448)
        return ag.LayerNormFn.apply(o, c, self.output.LayerNorm.weight, self.output.LayerNorm.bias, self.eps)


class ChunkAlign_CLS_enc4_align_ensemble(nn.Module):
    def __init__(self, global_enc, seq_enc, num_labels):
        super(ChunkAlign_CLS_enc4_align_ensemble, self).__init__()
        self.global_enc = global_enc
        self.seq_enc = seq_enc
        hg, hs = self.global_enc.config.hidden_size, self.seq_enc.config.hidden_size
        self.cls_ensemble_1 = nn.Linear(hg + hs, hg)
        self.num_labels = num_labels
        self.cls_layer_num = 2
        self.cls_layer = nn.ModuleList([ClsLayer2(self.global_enc.config) for _ in range(self.cls_layer_num)])
        self.cls_layer_lyx = nn.ModuleList([ClsLayer_lyx(self.global_enc.config) for _ in range(self.cls_layer_num)])
        self.classifier = nn.Linear(hg, 2)
        self.fusion_align = nn.Linear(hg * 2, 1024)
        self.prior = nn.Linear(hg, 1)
        self.cls_loss_fct = nn.CrossEntropyLoss()
        self.train_encoders = False

    def set_train_encoders(self, flag=True):
        """Not in the reference's ensemble class, which always wraps both encoders in torch.no_grad() (v10:893-912):
        flag=True runs them WITH gradients, as ChunkAlign_CLS_enc4_align does (v10:1016-1084)."""
        self.train_encoders = bool(flag)
        self.global_enc.trainable = self.seq_enc.trainable = self.train_encoders
        return self

    def align_loss_from_map(self, attn_weight, total_label, align_pos):
        """v10:983-987 on the [N,T,R] map (the reference discards this value, modeling_ensemble.py:484): CrossEntropyLoss over
        the rows with align_pos == 1, written as a masked mean of the per-row losses -- the reference's boolean row selection
        makes the host wait for the device (nonzero), after which every later launch of the step is exposed to launch latency."""
        attn_weight = attn_weight.masked_fill(attn_weight == 0, -1e5)
        attn_weight = torch.softmax(attn_weight, dim=-1)
        sel = (align_pos == 1).reshape(-1)
        r = attn_weight.shape[-1]
        label = total_label.reshape(-1).to(dtype=torch.int64)
        per_row = torch.nn.functional.cross_entropy(attn_weight.reshape(-1, r), torch.where(sel, label, torch.zeros_like(label)),
                                                    reduction="none")
        selw = sel.to(per_row.dtype)
        return (per_row * selw).sum() / selw.sum()          # no selected row: 0 / 0 = nan, as the reference's mean over nothing

    def forward(self, input_ids, img_feat, input_mask=None, label=None, token_type_ids=None, position_ids=None,
                head_mask=None, encoder_history_states=None, offsets=None, chunk_attention_mask=None,
                gather_index=None, align_pos=None, total_label=None, abstract_hidden_states=None,
                global_outputs=None):
        # global_outputs (not in the reference's signature): the global_enc outputs of exactly this call when the
        # caller has already computed them (Abstract_Specific batches this pass with its image-only pass)
        hypo_len = input_ids.size(1)
        ag.set_exact(getattr(self.global_enc.config, "modcr_dtype", "bf16") == "fp32")
        # opt-in (config.modcr_last_layer_rows): everything below reads text rows and the pooled [CLS] only, so the frozen encoders'
        # last layers need not run their token-wise blocks over the region rows
        lr = {}
        if (getattr(self.global_enc.config, "modcr_last_layer_rows", False) and not (self.train_encoders and torch.is_grad_enabled())
                and encoder_history_states is None):
            lr = {"modcr_last_rows": hypo_len}
        with (torch.enable_grad() if self.train_encoders and torch.is_grad_enabled() else torch.no_grad()):
            outputs = global_outputs if global_outputs is not None else self.global_enc(
                input_ids, img_feats=img_feat, attention_mask=input_mask, position_ids=position_ids,
                token_type_ids=token_type_ids, head_mask=head_mask, encoder_history_states=encoder_history_states, **lr)
            global_output = outputs[0]
            global_CLS = outputs[1]
            img_mask = input_mask[:, hypo_len:]
            seq_outputs, chunk_hidden_states = self.seq_enc(input_ids, img_feats=img_feat, img_mask=img_mask,
                                                            input_mask=input_mask, attention_mask=chunk_attention_mask,
                                                            position_ids=position_ids, token_type_ids=token_type_ids,
                                                            head_mask=head_mask, offsets=offsets,
                                                            gather_index=gather_index, **lr)
            chunk_CLS = seq_outputs[1]
            chunk_align = seq_outputs[0][:, 1:hypo_len]
            global_hypo = global_output[:, 1:hypo_len]
            chunk_hidden = chunk_hidden_states[:, 1:hypo_len]
            # [global | chunk-align | chunk-hidden] along the token axis (v10:913): plain copies, or -- frozen bf16 states --
            # the three views themselves (cross_attention_lyx reads them in place)
            blocks = (global_hypo, chunk_align, chunk_hidden)
            if all(m.cross_attention.takes_row_blocks(blocks) for m in self.cls_layer_lyx):
                self_chunk_align_ = blocks
            else:
                self_chunk_align_ = torch.cat(blocks, dim=1)
            if global_CLS.dtype != torch.float32:          # frozen route: bf16 pooler rows; trainable route: fp32 with grad
                global_CLS, chunk_CLS = mh.convert(global_CLS, mh.F32), mh.convert(chunk_CLS, mh.F32)
            cls_in = torch.cat((global_CLS, chunk_CLS), -1)
        CLS_ensem = ag.linear(cls_in, self.cls_ensemble_1.weight, self.cls_ensemble_1.bias)
        for layer_module in self.cls_layer_lyx:
            CLS_ensem = layer_module(self_chunk_align_, CLS_ensem, None, None, None)
        align_loss = None
        if total_label is not None and seq_outputs.align_map is not None:
            # trainable encoders: the loss keeps its graph (v10:1067-1073 adds it to the objective of ChunkAlign_CLS_enc4_align);
            # the ensemble's own use computes and discards it under no_grad (v10:981-987)
            with (torch.enable_grad() if self.train_encoders and torch.is_grad_enabled() else torch.no_grad()):
                align_loss = self.align_loss_from_map(seq_outputs.align_map, total_label, align_pos)
        return CLS_ensem, align_loss, ([], None)



def binary_to_mp(logit, num_labels=4):
    """v10:363-373: P(answer is right) of every (question, choice) pair, regrouped per question; [N,2] -> [N/4,4].
    A softmax over two columns of a [N,2] tensor: torch ops (no gradient flows through it at v10:1061-1065)."""
    return torch.softmax(logit, dim=1)[:, 1].reshape(-1, num_labels)


class ChunkAlign_CLS_enc4_align(nn.Module):
    """v10:1016-1084: the variant that trains BOTH Oscar encoders (no torch.no_grad() around them), three ClsLayer2 over
    [global | chunk-align | chunk-hidden] text states, a binary right / wrong classifier per (question, choice) pair and
    the align loss over the head- and layer-summed text -> region attention of seq_enc's last three layers.
    forward(...) -> (loss_cls_0, matched_0, align_loss, correct, total_sum), same signature as the reference."""

    def __init__(self, global_enc, seq_enc, num_labels):
        super(ChunkAlign_CLS_enc4_align, self).__init__()
        self.global_enc = global_enc
        self.seq_enc = seq_enc
        hg, hs = self.global_enc.config.hidden_size, self.seq_enc.config.hidden_size
        self.cls_ensemble = nn.Linear(hg + hs, hg)
        self.num_labels = num_labels
        self.cls_layer_num = 3
        self.cls_layer = nn.ModuleList([ClsLayer2(self.global_enc.config) for _ in range(self.cls_layer_num)])
        self.classifier = nn.Linear(hg, 2)
        self.cls_loss_fct = nn.CrossEntropyLoss()
        self.global_enc.trainable = self.seq_enc.trainable = True         # both encoders with gradients (v10:1034-1046)

    def _logits(self, input_ids, img_feat, input_mask, token_type_ids, position_ids, head_mask, encoder_history_states, offsets,
                chunk_attention_mask, gather_index):
        """both encoders, cls_ensemble, the three ClsLayer2, classifier (v10:1032-1057 = :1089-1117): ([N,2] logits, seq_outputs)"""
        hypo_len = input_ids.size(1)
        ag.set_exact(getattr(self.global_enc.config, "modcr_dtype", "bf16") == "fp32")
        outputs = self.global_enc(input_ids, img_feats=img_feat, attention_mask=input_mask, position_ids=position_ids,
                                  token_type_ids=token_type_ids, head_mask=head_mask,
                                  encoder_history_states=encoder_history_states)
        global_output, global_CLS = outputs[0], outputs[1]
        img_mask = input_mask[:, hypo_len:]
        seq_outputs, chunk_hidden_states = self.seq_enc(input_ids, img_feats=img_feat, img_mask=img_mask, input_mask=input_mask,
                                                        attention_mask=chunk_attention_mask, position_ids=position_ids,
                                                        token_type_ids=token_type_ids, head_mask=head_mask, offsets=offsets,
                                                        gather_index=gather_index)
        chunk_CLS = seq_outputs[1]
        if global_CLS.dtype != torch.float32:            # frozen route under no_grad (evaluation): bf16 pooler rows
            global_CLS, chunk_CLS = mh.convert(global_CLS, mh.F32), mh.convert(chunk_CLS, mh.F32)
        CLS_ensem = ag.linear(torch.cat((global_CLS, chunk_CLS), -1), self.cls_ensemble.weight, self.cls_ensemble.bias)
        self_chunk_align = torch.cat((global_output[:, 1:hypo_len], seq_outputs[0][:, 1:hypo_len],
                                      chunk_hidden_states[:, 1:hypo_len]), dim=1)          # v10:1050-1053; plain copies
        word_mask = (1.0 - input_mask[:, 1:hypo_len].to(torch.float32).unsqueeze(1)) * -10000.0
        word_mask = torch.cat((word_mask, word_mask, word_mask), -1)
        for layer_module in self.cls_layer:
            CLS_ensem, _ = layer_module(self_chunk_align, CLS_ensem, word_mask)
        return ag.linear(CLS_ensem, self.classifier.weight, self.classifier.bias), seq_outputs

    def evaluate(self, input_ids, img_feat, input_mask=None, label=None, token_type_ids=None, position_ids=None,
                 head_mask=None, encoder_history_states=None, offsets=None, chunk_attention_mask=None, gather_index=None):
        """v10:1086-1124 -> (matched_0, pre, logit_expl)"""
        with torch.no_grad():
            logits, _ = self._logits(input_ids, img_feat, input_mask, token_type_ids, position_ids, head_mask,
                                     encoder_history_states, offsets, chunk_attention_mask, gather_index)
            logit_expl = binary_to_mp(logits, self.num_labels)
            pre = logit_expl.max(dim=-1)[1]
            matched_0 = pre == torch.argmax(label.reshape(-1, self.num_labels), -1)
        return matched_0, pre, logit_expl

    def forward(self, input_ids, img_feat, input_mask=None, label=None, token_type_ids=None, position_ids=None,
                head_mask=None, encoder_history_states=None, offsets=None, chunk_attention_mask=None,
                gather_index=None, align_pos=None, total_label=None):
        logits, seq_outputs = self._logits(input_ids, img_feat, input_mask, token_type_ids, position_ids, head_mask,
                                           encoder_history_states, offsets, chunk_attention_mask, gather_index)
        # CrossEntropyLoss over [N,2] logits with class-index labels = the soft-label CE kernel on their one-hot rows
        onehot = torch.nn.functional.one_hot(label.reshape(-1).to(torch.int64), 2).to(torch.float32)
        loss_cls_0 = ag.McCeFn.apply(logits.view(-1, 2), onehot)
        with torch.no_grad():
            pre = binary_to_mp(logits.detach(), self.num_labels).max(dim=-1)[1]
            matched_0 = pre == torch.argmax(label.reshape(-1, self.num_labels), -1)
        # align loss (v10:1067-1073) on the differentiable align map of the trainable seq_enc (the frozen route's map
        # under no_grad): a masked softmax + CE over a few selected [R]-rows, torch ops
        amap = seq_outputs.align_map
        attn_weight = torch.softmax(amap.masked_fill(amap == 0, -1e5), dim=-1)
        sel = align_pos == 1
        total_label_align = total_label[sel].to(dtype=torch.int64)
        attn_weight_align = attn_weight[sel, :]
        align_loss = self.cls_loss_fct(attn_weight_align, total_label_align)
        total_sum = int(total_label_align.size(0))
        correct = int((torch.argmax(attn_weight_align, -1) == total_label_align).sum().item())
        return loss_cls_0, matched_0, align_loss, correct, total_sum
