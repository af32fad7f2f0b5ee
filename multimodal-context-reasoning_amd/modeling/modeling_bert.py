"""Drop-in for the hot-path classes of the reference's modeling/modeling_bert.py:
CaptionBertSelfAttention (:25-75) and CaptionBertAttention (:78-93), same constructor and forward
signatures and return tuples, arithmetic in libmodcr_hip (fused QKV projection + attention)."""
import warnings

import torch
from torch import nn

import modcr_hip as mh
from .bert_primitives import BertSelfAttention, BertSelfOutput, additive_to_binary


def split_additive_mask(attention_mask, n, s, l):
    """additive [N,1,1,L] -> key mask [N,L]; additive [N,1,S,L] -> packed bits [N,S,LW]."""
    if attention_mask.dim() != 4 or attention_mask.shape[1] != 1:
        raise ValueError("attention_mask must be [N,1,1,L] or [N,1,S,L], got %s" % (tuple(attention_mask.shape),))
    if attention_mask.shape[2] == 1:
        return additive_to_binary(attention_mask.reshape(n, l)), None
    return None, mh.pack_mask_bits(additive_to_binary(attention_mask.reshape(n, s, l)))


class CaptionBertSelfAttention(BertSelfAttention):
    _warned = False

    def __init__(self, config):
        super(CaptionBertSelfAttention, self).__init__(config)
        self.output_attentions = config.output_attentions

    def hip_forward(self, x, key_mask=None, mask_bits=None, hist=None, chunk_id=None, want_probs=False,
                    align_map=None, align_t=0, workspace=None, out=None):
        w, b = self.packed_qkv(x.dtype)
        drop = None
        if self.training and self.dropout.p > 0.0:      # nn.Dropout on the probabilities (modeling_bert.py:69), training mode
            n, s, h = x.shape
            if x.dtype == torch.bfloat16 and not want_probs:
                seed, off = mh.DROPOUT.take(n * self.num_attention_heads * s * (s + (0 if hist is None else hist.shape[1])))
                drop = (float(self.dropout.p), seed, off)
            elif not CaptionBertSelfAttention._warned:
                CaptionBertSelfAttention._warned = True
                warnings.warn("attention-probability dropout is implemented on the bf16 path without a probabilities output: "
                              "not applied (dtype=%s, output_attentions materialised=%s)" % (x.dtype, want_probs))
        return mh.qkv_attn(x, w, b, key_mask=key_mask, mask_bits=mask_bits, hist=hist, chunk_id=chunk_id,
                           want_probs=want_probs, align_map=align_map, align_t=align_t,
                           num_heads=self.num_attention_heads, workspace=workspace, out=out, attn_dropout=drop)

    def forward(self, hidden_states, attention_mask, head_mask=None, history_state=None):
        if head_mask is not None:
            raise NotImplementedError("head_mask is always None on the ModCR path (modeling_transfomres.py:657)")
        n, s, _ = hidden_states.shape
        l = s + (0 if history_state is None else history_state.shape[1])
        km, bits = split_additive_mask(attention_mask, n, s, l)
        ctx, probs = self.hip_forward(hidden_states, km, bits, hist=history_state,
                                      want_probs=self.output_attentions)
        return (ctx, probs) if self.output_attentions else (ctx,)


class CaptionBertAttention(nn.Module):
    def __init__(self, config):
        super(CaptionBertAttention, self).__init__()
        self.self = CaptionBertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask, head_mask=None, history_state=None):
        self_outputs = self.self(input_tensor, attention_mask, head_mask, history_state)
        attention_output = self.output(self_outputs[0], input_tensor)
        return (attention_output,) + self_outputs[1:]
