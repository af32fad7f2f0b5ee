"""Drop-in for the hot-path classes of the reference's modeling/modeling_bert.py:
CaptionBertSelfAttention (:25-75) and CaptionBertAttention (:78-93), same constructor and forward
signatures and return tuples, arithmetic in libmodcr_hip (fused QKV projection + attention)."""
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
    def __init__(self, config):
        super(CaptionBertSelfAttention, self).__init__(config)
        self.output_attentions = config.output_attentions
        # training mode: the probabilities this module returns (and the align map summed from them) are the ones AFTER nn.Dropout,
        # as in the reference (modeling_bert.py:69-74 applies self.dropout before building `outputs`); False = the un-dropped ones
        self.side_post_dropout = bool(getattr(config, "modcr_align_map_post_dropout", True))

    def hip_forward(self, x, key_mask=None, mask_bits=None, hist=None, chunk_id=None, want_probs=False,
                    align_map=None, align_t=0, workspace=None, out=None):
        w, b = self.packed_qkv(x.dtype)
        drop = None
        if self.training and self.dropout.p > 0.0:      # nn.Dropout on the probabilities (modeling_bert.py:69), training mode
            n, s, h = x.shape
            l = s + (0 if hist is None else hist.shape[1])
            # every route carries the mask for P + S <= 256 (bf16 tile / older kernels, the exact-fp32 route); a probabilities output
            # under dropout is the post-dropout one, as the reference returns it (modeling_bert.py:69-74) -- on shapes the bf16 tile
            # kernels do not serve, mh.qkv_attn runs that call on the exact-fp32 route.  What the library does not offer is an error,
            # not a silently un-dropped forward:
            if l > 256:
                raise NotImplementedError("attention-probability dropout: P + S = %d exceeds the 256 keys the attention kernels take" % l)
            if want_probs and not self.side_post_dropout:
                raise NotImplementedError("a probabilities output in training mode is the post-dropout one (modeling_bert.py:69-74); "
                                          "config.modcr_align_map_post_dropout=False offers the un-dropped align map only")
            seed, off = mh.DROPOUT.take(n * self.num_attention_heads * s * l)
            drop = (float(self.dropout.p), seed, off)
        return mh.qkv_attn(x, w, b, key_mask=key_mask, mask_bits=mask_bits, hist=hist, chunk_id=chunk_id,
                           want_probs=want_probs, align_map=align_map, align_t=align_t,
                           num_heads=self.num_attention_heads, workspace=workspace, out=out, attn_dropout=drop,
                           side_post_dropout=self.side_post_dropout and drop is not None and (want_probs or align_map is not None))

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
