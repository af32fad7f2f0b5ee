"""Drop-in for the hot-path classes of the reference's modeling/modeling_transfomres.py (sic):
CaptionBertLayer (:471-502), CaptionBertEncoder (:504-562) and BertImgModel (:564-727) -- the
global_enc of ModCR.  Same constructor/forward signatures and positional return tuples."""
import torch
from torch import nn

import modcr_hip as mh
from .bert_primitives import (BertEmbeddings, BertIntermediate, BertOutput, BertPooler, BertPreTrainedModel,
                              EncoderOutputs, PackCache, additive_to_binary, compute_dtype, packed_linear,
                              packed_ln, _pad64)
from .hip_layers import Workspace
from .modeling_bert import CaptionBertAttention, split_additive_mask


FFN_SPLIT = 1      # tools may raise it for an A/B run: FFN row chunks per layer
PACK_SHORT = 64    # sequences of at most this many rows are packed several to an attention row block (0 = off)


class CaptionBertLayer(nn.Module):
    def __init__(self, config):
        super(CaptionBertLayer, self).__init__()
        self.attention = CaptionBertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def hip_forward(self, x, key_mask=None, mask_bits=None, hist=None, chunk_id=None, want_probs=False,
                    align_map=None, align_t=0, ws=None, out_rows=None, seq_len=None):
        """out_rows = k (opt-in, the LAST layer of a frozen pass whose caller consumes only the first k rows of every sequence:
        config.modcr_last_layer_rows): attention runs over all rows (keys and values of every row are needed), the token-wise
        blocks behind it -- BertSelfOutput, BertIntermediate, BertOutput -- only over rows 0..k-1; returns [N,k,H].  seq_len: the
        sequence length when x is a packed view of several short sequences per row block."""
        ws = ws or Workspace()
        n, s, h = x.shape
        need = mh.lib().modcr_qkv_attn_workspace(n, s, 0 if hist is None else hist.shape[1], h, mh.dt_of(x))
        ctx, probs = self.attention.self.hip_forward(x, key_mask, mask_bits, hist, chunk_id, want_probs, align_map,
                                                     align_t, ws.get("attn", need, x.device) if need else None)
        if out_rows is not None:
            sl = seq_len or s
            if out_rows < sl:
                ctx = ctx.reshape(-1, sl, h)[:, :out_rows].contiguous()
                x = x.reshape(-1, sl, h)[:, :out_rows].contiguous()
                n, s = ctx.shape[0], out_rows
            elif seq_len:
                ctx, x = ctx.reshape(-1, sl, h), x.reshape(-1, sl, h)
                n, s = ctx.shape[0], sl
        pre = ws.get("preln", n * s * h * 4, x.device)
        a = self.attention.output(ctx, x, pre)
        if FFN_SPLIT <= 1 or (n * s) % (8 * FFN_SPLIT):
            y = self.output(self.intermediate(a), a, pre)
        else:
            # the FFN in row chunks: BertIntermediate's [rows, 4H] output of one chunk is read back by BertOutput while
            # it still sits in the 256 MB Infinity Cache (283 MB at 46080 rows do not)
            m, rows = n * s, (n * s) // FFN_SPLIT
            a2 = a.reshape(m, h)
            y = torch.empty_like(a2)
            for c in range(FFN_SPLIT):
                ac = a2[c * rows:(c + 1) * rows]
                self.output(self.intermediate(ac), ac, pre, out=y[c * rows:(c + 1) * rows])
            y = y.view(n, s, h)
        return y, probs

    def forward(self, hidden_states, attention_mask, head_mask=None, history_state=None):
        n, s, _ = hidden_states.shape
        l = s + (0 if history_state is None else history_state.shape[1])
        km, bits = split_additive_mask(attention_mask, n, s, l)
        want = self.attention.self.output_attentions
        y, probs = self.hip_forward(hidden_states, km, bits, hist=history_state, want_probs=want)
        return (y, probs) if want else (y,)

    def attention_cal(self, hidden_states, attention_mask, head_mask=None, history_state=None):
        return self.attention(hidden_states, attention_mask, head_mask, history_state)[0]

    def forward_ffn(self, attention_output):
        return self.output(self.intermediate(attention_output), attention_output)


class CaptionBertEncoder(nn.Module):
    def __init__(self, config):
        super(CaptionBertEncoder, self).__init__()
        self.output_attentions = config.output_attentions
        self.output_hidden_states = config.output_hidden_states
        self.materialize = getattr(config, "modcr_materialize_attentions", False)
        self.layer = nn.ModuleList([CaptionBertLayer(config) for _ in range(config.num_hidden_layers)])

    def hip_forward(self, x, key_mask, encoder_history_states=None, ws=None, last_rows=None, dense_bits=None):
        """last_rows = k (opt-in): the caller reads only rows 0..k-1 of the final hidden states (ModCR: the text rows, or the
        [CLS] row of the image-only pass) -- the last layer's token-wise blocks skip the other rows and the result is [N,k,H]."""
        ws = ws or Workspace()
        all_hidden, all_att = (), ()
        want = self.output_attentions and self.materialize
        if want or self.output_hidden_states or encoder_history_states is not None:
            last_rows = None
        # Short sequences (the image-only pass: S = 1 + R rows, modeling_ensemble.py:466-471) packed k to an attention row block
        # under a block-diagonal mask, so that the 192- / 256-token tile kernels are filled (S = 37 alone runs the older kernel at
        # a third of their rate; every other op of a layer is row-wise and does not see the packing).
        n, s, h = x.shape
        pack_k, bits = 1, None
        if dense_bits is not None:
            # a per-(query, key) mask (3-D attention_mask, modeling_transfomres.py:629-630): every layer on the dense-mask kernels
            if encoder_history_states is not None:
                raise NotImplementedError("a 3-D attention_mask together with encoder_history_states")
            for i, layer in enumerate(self.layer):
                if self.output_hidden_states:
                    all_hidden = all_hidden + (x,)
                x, probs = layer.hip_forward(x, mask_bits=dense_bits, want_probs=want, ws=ws)
                if self.output_attentions:
                    all_att = all_att + (probs,)
            if self.output_hidden_states:
                all_hidden = all_hidden + (x,)
            outputs = (x,)
            if self.output_hidden_states:
                outputs = outputs + (all_hidden,)
            if self.output_attentions:
                outputs = outputs + (all_att,)
            return outputs
        if (PACK_SHORT and x.dtype == torch.bfloat16 and not want and encoder_history_states is None and not self.output_hidden_states
                and key_mask is not None and s <= PACK_SHORT):
            pack_k = mh.pack_factor(n, s)
            if pack_k > 1 and pack_k * s > 64:
                bits = mh.build_packed_mask(key_mask, pack_k)
                x = x.view(n // pack_k, pack_k * s, h)
            else:
                pack_k = 1
        for i, layer in enumerate(self.layer):
            if self.output_hidden_states:
                all_hidden = all_hidden + (x,)
            hist = None if encoder_history_states is None else encoder_history_states[i]
            last = i == len(self.layer) - 1
            if pack_k > 1:
                if last and last_rows is not None:
                    x, probs = layer.hip_forward(x, mask_bits=bits, ws=ws, out_rows=last_rows, seq_len=s)
                else:
                    x, probs = layer.hip_forward(x, mask_bits=bits, ws=ws)
                    if last:
                        x = x.view(n, s, h)
                if self.output_attentions:
                    all_att = all_att + (None,)
                continue
            x, probs = layer.hip_forward(x, key_mask=key_mask, hist=hist, want_probs=want, ws=ws,
                                         out_rows=last_rows if last else None)
            if self.output_attentions:
                all_att = all_att + (probs,)
        if self.output_hidden_states:
            all_hidden = all_hidden + (x,)
        outputs = (x,)
        if self.output_hidden_states:
            outputs = outputs + (all_hidden,)
        if self.output_attentions:
            outputs = outputs + (all_att,)
        return outputs

    def hip_forward_pair(self, rows, n, s1, s2, mask1, mask2, ws=None):
        """Two passes of the SAME encoder over different sequences (ModCR: the full [text | regions] pass of
        calec and the image-only [CLS | regions] pass of Abstract_Specific, modeling_ensemble.py:466-471) as one
        batch: `rows` [n*s1 + n*s2, H] holds both; attention runs per pass (two launches per layer), the
        token-wise blocks (BertSelfOutput, BertIntermediate, BertOutput) once over all rows -- at N=256 that is
        71936 rows = 2.9 rounds of the 192x384 GEMM tiles instead of 1.9 + 2 half-empty ones."""
        ws = ws or Workspace()
        h = rows.shape[1]
        m1 = n * s1
        ctx = torch.empty_like(rows)
        for layer in self.layer:
            xa, xb = rows[:m1].view(n, s1, h), rows[m1:].view(n, s2, h)
            layer.attention.self.hip_forward(xa, mask1, out=ctx[:m1].view(n, s1, h))
            layer.attention.self.hip_forward(xb, mask2, out=ctx[m1:].view(n, s2, h))
            pre = ws.get("preln", rows.shape[0] * h * 4, rows.device)
            a = layer.attention.output(ctx, rows, pre)
            rows = layer.output(layer.intermediate(a), a, pre)
        return rows[:m1].view(n, s1, h), rows[m1:].view(n, s2, h)

    def forward(self, hidden_states, attention_mask, head_mask=None, encoder_history_states=None):
        n, s, _ = hidden_states.shape
        p = 0 if not encoder_history_states else encoder_history_states[0].shape[1]
        km, bits = split_additive_mask(attention_mask, n, s, s + p)
        if bits is not None:
            raise NotImplementedError("global_enc always uses the broadcast padding mask (modeling_transfomres.py:628-641)")
        return self.hip_forward(hidden_states, km, encoder_history_states)


class ImgEmbedMixin(object):
    """img_embedding + LayerNorm of the region features (modeling_transfomres.py:676-681), written
    behind the text rows of each sequence (the torch.cat at :684)."""

    _cast_cache = None      # (img_feats tensor, its version, kp, dtype, epoch) -> padded copy in the storage dtype, shared by all encoders
    _epoch = 0              # bumped by every Abstract_Specific.forward: nothing computed in one model call is re-used by the next

    def region_rows(self, img_feats, dt):
        """LayerNorm(img_embedding(img_feats)) as [N*R, H] rows in the storage dtype `dt` (modeling_transfomres.py:676-680), before
        the dropout.  One ModCR step embeds the SAME region features three times (global_enc full pass, global_enc image-only pass,
        seq_enc: modeling_ensemble.py:466-471, v10:896-907).  The padded storage-dtype copy of the features is shared by
        all three, and an encoder whose weights have not changed re-uses its LayerNorm-ed region rows for a second call
        on the same tensor (global_enc's two passes).  Validity = the very same tensor object at the same version (a
        reference is held, so its memory cannot be recycled under the cache) within ONE forward of the whole model
        (`_epoch`): a common subexpression of a step is computed once, nothing is carried from step to step."""
        n, r, d = img_feats.shape
        if not self.use_img_layernorm:
            raise NotImplementedError("use_img_layernorm=False: the Oscar checkpoints ModCR loads set it (run_PMR_ModCR.py:720)")
        w, b = packed_linear(self._cache, ("img", dt), self.img_embedding, dt)
        g, be = packed_ln(self._cache, "imgln", self.LayerNorm)
        kp = _pad64(d) if dt == torch.bfloat16 else d
        rc = getattr(self, "_region_cache", None)
        ep = ImgEmbedMixin._epoch
        if (rc is not None and rc[0] is img_feats and rc[1] == img_feats._version and rc[2] is w and rc[3] is g and rc[4].dtype == dt
                and rc[5] == ep and not torch.is_grad_enabled()):
            return rc[4]
        cc = ImgEmbedMixin._cast_cache
        if cc is not None and cc[0] is img_feats and cc[1] == img_feats._version and cc[2] == kp and cc[3] == dt and cc[5] == ep:
            src = cc[4]
        else:
            src = mh.cast_pad(img_feats, kp, mh.BF16 if dt == torch.bfloat16 else mh.F32)     # fp32 [N*R,2054] -> dtype [N*R,Kp]
            ImgEmbedMixin._cast_cache = (img_feats, img_feats._version, kp, dt, src, ep)
        pre = mh.linear(src, w, b, out_dtype=mh.F32)
        rows = mh.layernorm(pre, g, be, self.config.img_layer_norm_eps, out_dtype=mh.BF16 if dt == torch.bfloat16 else mh.F32)
        if not torch.is_grad_enabled():
            self._region_cache = (img_feats, img_feats._version, w, g, rows, ep)
        return rows

    def embed_regions(self, img_feats, out, t, dropout=None):
        """the region rows behind the text rows of each sequence of `out` [N, T+R, H] (the torch.cat at :684).
        dropout = (p, seed, offset): self.dropout of the reference (:681) rides on the pass that places the rows
        (counters = flat indices of `out`; BertEmbeddings.forward(dropout=) covers the text rows under the same (seed, offset))."""
        return mh.rows_scatter_dropout(self.region_rows(img_feats, out.dtype), out, t, dropout)


class BertImgModel(BertPreTrainedModel, ImgEmbedMixin):
    """Expand from BertModel to handle image region features as input (global_enc)."""

    def __init__(self, config):
        super(BertImgModel, self).__init__(config)
        self.embeddings = BertEmbeddings(config)
        self.encoder = CaptionBertEncoder(config)
        self.pooler = BertPooler(config)
        self.img_dim = config.img_feature_dim
        self.img_feature_type = config.img_feature_type
        self.use_img_layernorm = getattr(config, "use_img_layernorm", None)
        if config.img_feature_type not in ("frcnn", "faster_r-cnn", None):
            raise NotImplementedError("img_feature_type=%r (ModCR uses region features)" % (config.img_feature_type,))
        self.img_embedding = nn.Linear(self.img_dim, self.config.hidden_size, bias=True)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        if self.use_img_layernorm:
            self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.img_layer_norm_eps)
        self._cache = PackCache()
        self._ws = Workspace()
        self.init_weights()

    def forward_pair(self, input_ids, token_type_ids, attention_mask, img_feats, img_attention_mask, position_ids=None):
        """forward(input_ids, token_type_ids, attention_mask, img_feats=...) AND forward(input_ids[:, :1],
        attention_mask=img_attention_mask, img_feats=...) (the image-only call of modeling_ensemble.py:466-471) in
        one batch; returns the two output tuples (sequence_output, pooled_output).  The region embeddings are
        computed once and copied behind the [CLS] row of the image-only sequences."""
        n, t = input_ids.shape
        r = img_feats.shape[1]
        h = self.config.hidden_size
        dt = compute_dtype(self.config)
        s1, s2 = t + r, 1 + r
        rows = torch.empty((n * s1 + n * s2, h), dtype=dt, device=input_ids.device)
        xa, xb = rows[:n * s1].view(n, s1, h), rows[n * s1:].view(n, s2, h)
        da = db = None
        if self.training and self.dropout.p > 0.0:          # BertEmbeddings.dropout + self.dropout (:681) of each of the two calls
            da = (self.dropout.p,) + tuple(mh.DROPOUT.take(xa.numel()))
            db = (self.dropout.p,) + tuple(mh.DROPOUT.take(xb.numel()))
        self.embeddings(input_ids, token_type_ids, position_ids, out=xa, dropout=da)
        self.embed_regions(img_feats, xa, t, dropout=da)
        self.embeddings(input_ids[:, :1].contiguous(), None, None, out=xb, dropout=db)
        self.embed_regions(img_feats, xb, 1, dropout=db)     # (the rows of the call above: region_rows' cache, or recomputed)
        ya, yb = self.encoder.hip_forward_pair(rows, n, s1, s2, attention_mask.to(torch.float32),
                                               img_attention_mask.to(torch.float32), self._ws)
        att = ((None,) * len(self.encoder.layer),) if self.encoder.output_attentions else ()
        return (ya, self.pooler(ya)) + att, (yb, self.pooler(yb)) + att

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, position_ids=None, head_mask=None,
                img_feats=None, encoder_history_states=None, modcr_last_rows=None):
        """modcr_last_rows = k (not in the reference's signature; frozen route only): sequence_output is [N,k,H], the first k rows of
        every sequence -- the last layer computes nothing behind its attention for the rows the caller does not read."""
        if head_mask is not None:
            raise NotImplementedError("head_mask is never set on the ModCR path")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        dense_bits = None
        if attention_mask.dim() == 3:
            # [N, S, S] 0/1 per (query, key): the reference extends it to [N, 1, S, S] and adds (1 - m) * -10000
            # (modeling_transfomres.py:629-630); ModCR itself never passes one to global_enc.  Frozen route.
            if getattr(self, "trainable", False) and torch.is_grad_enabled():
                raise NotImplementedError("3-D attention_mask on the trainable global_enc route")
            dense_bits = mh.pack_mask_bits(attention_mask.to(torch.float32))
        elif attention_mask.dim() != 2:
            raise NotImplementedError
        if encoder_history_states:
            assert img_feats is None, "Cannot take image features while using encoder history states"
        if getattr(self, "trainable", False) and torch.is_grad_enabled() and not encoder_history_states:
            # encoder with gradients (v10:1016-1084 runs it outside no_grad): same kernels behind autograd Functions
            from . import trainable_encoders
            seq, pooled = trainable_encoders.global_encoder(self, input_ids, token_type_ids, attention_mask, position_ids,
                                                            img_feats)
            att = ((None,) * len(self.encoder.layer),) if self.encoder.output_attentions else ()
            return (seq, pooled) + att
        n, t = input_ids.shape
        r = 0 if img_feats is None else img_feats.shape[1]
        dt = compute_dtype(self.config)
        x = torch.empty((n, t + r, self.config.hidden_size), dtype=dt, device=input_ids.device)
        drop = None
        if self.training and self.dropout.p > 0.0:          # BertEmbeddings.dropout (a_bert:210) and self.dropout (:681): same p,
            drop = (self.dropout.p,) + tuple(mh.DROPOUT.take(x.numel()))      # one mask over x, applied by the passes that write its rows
        self.embeddings(input_ids, token_type_ids, position_ids, out=x, dropout=drop)
        if img_feats is not None:
            self.embed_regions(img_feats, x, t, dropout=drop)
        encoder_outputs = self.encoder.hip_forward(x, None if dense_bits is not None else attention_mask.to(torch.float32),
                                                   encoder_history_states, self._ws, last_rows=modcr_last_rows, dense_bits=dense_bits)
        sequence_output = encoder_outputs[0]
        pooled_output = self.pooler(sequence_output)
        return (sequence_output, pooled_output,) + encoder_outputs[1:]
