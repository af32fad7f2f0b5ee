"""The reference's prefix-conditioned RoBERTa-large: a stand-in (PrefixPoolerStandIn) and the real 24-layer body
on the HIP kernels (RobertaPrefixModel).

The reference imports `RobertaModel` from `local_transformers.adapter_transformers` (a fork that
accepts `prompt_embeddings=` / `input_mask=`), which is NOT in the reference tree
(.MISSING_LARGE_BLOBS; SURVEY.md section 8c: "parity unpinned").  Its 24-layer body is row (f)-1
"next" of the scope table.  Until it is built on the same kernels, this module keeps the call
contract of modeling_ensemble.py:501-503 -- (input_ids, token_type_ids, attention_mask,
prompt_embeddings [N,10,1024], input_mask [N,10]) -> (sequence_output, pooled [N,1024]) -- with a
small trainable pooler over the prefix so the heads, the loss and every gradient path of the
ModCR step (mappers, cls_layer_lyx, cls_ensemble_1, scorer) are exercised end to end.
"""
import torch
from torch import nn

import modcr_hip as mh
from . import hip_autograd as ag


class PrefixPoolerStandIn(nn.Module):
    def __init__(self, prefix_len=10, hidden=1024):
        super().__init__()
        self.dense = nn.Linear(prefix_len * hidden, hidden)
        nn.init.normal_(self.dense.weight, std=0.02)
        nn.init.zeros_(self.dense.bias)
        self.hidden = hidden

    def forward(self, input_ids=None, token_type_ids=None, attention_mask=None, prompt_embeddings=None,
                input_mask=None):
        n = prompt_embeddings.shape[0]
        flat = prompt_embeddings.reshape(n, -1)
        # (exact-fp32 parity mode: no bf16 copy of the prefix, the VALU GEMM -- the trajectory test compares ten optimisation steps
        # with the oracle at 1e-3, and this pooler sits between every trained parameter and the loss)
        pooled = ag.linear(flat if ag.EXACT else ag.ToBf16Fn.apply(flat), self.dense.weight, self.dense.bias, act=mh.ACT_TANH)
        return None, pooled


class _RobertaLayer(nn.Module):
    """parameter container with HF key names (attention.self.{query,key,value}, attention.output.{dense,LayerNorm},
    intermediate.dense, output.{dense,LayerNorm})"""

    def __init__(self, h, inter, eps):
        super().__init__()
        self.attention = nn.Module()
        self.attention.self = nn.Module()
        for nm in ("query", "key", "value"):
            setattr(self.attention.self, nm, nn.Linear(h, h))
        self.attention.output = nn.Module()
        self.attention.output.dense = nn.Linear(h, h)
        self.attention.output.LayerNorm = nn.LayerNorm(h, eps=eps)
        self.intermediate = nn.Module()
        self.intermediate.dense = nn.Linear(h, inter)
        self.output = nn.Module()
        self.output.dense = nn.Linear(inter, h)
        self.output.LayerNorm = nn.LayerNorm(h, eps=eps)

    def ordered_params(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n in ag.BertLayerFn.NAMES]


class RobertaPrefixModel(nn.Module):
    """RoBERTa-large body with `prompt_embeddings=` / `input_mask=` (the call of modeling_ensemble.py:501-503) on the
    ModCR kernels: trainable end to end (forward AND backward of every layer are C-ABI calls).

    The reference's module (`local_transformers.adapter_transformers.models.roberta`) is absent from the reference
    tree, so WHERE the prefix vectors enter is this build's documented choice, not a pinned behaviour (SURVEY 8c,
    "parity unpinned"): the N x P x H prompt embeddings are spliced into the embedding output right after `<s>`
    (after the embedding LayerNorm, without position / type embeddings of their own), `attention_mask` is extended
    with `input_mask` at the same place, and the pooler reads `<s>` (row 0).  Layer arithmetic is BERT's
    (a_bert:238-451) with RoBERTa's hyper-parameters: 24 layers, H = 1024, 16 heads, I = 4096, eps 1e-5, position
    ids = padding_idx + cumulative count of non-pad tokens.  State-dict keys are HF RoBERTa's
    (`embeddings.*`, `encoder.layer.{i}.*`, `pooler.dense.*`)."""

    def __init__(self, vocab_size=50265, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                 intermediate_size=4096, max_position_embeddings=514, type_vocab_size=2, layer_norm_eps=1e-5,
                 pad_token_id=1, initializer_range=0.02, attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0):
        super().__init__()
        self.h, self.a, self.eps, self.pad = hidden_size, num_attention_heads, layer_norm_eps, pad_token_id
        # training-mode dropouts of the layers (roberta-large's config.json has 0.1 / 0.1); 0 = eval arithmetic
        self.attn_p, self.hidden_p = float(attention_probs_dropout_prob), float(hidden_dropout_prob)
        self.embeddings = nn.Module()
        self.embeddings.word_embeddings = nn.Embedding(vocab_size, hidden_size, padding_idx=pad_token_id)
        self.embeddings.position_embeddings = nn.Embedding(max_position_embeddings, hidden_size, padding_idx=pad_token_id)
        self.embeddings.token_type_embeddings = nn.Embedding(type_vocab_size, hidden_size)
        self.embeddings.LayerNorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        self.encoder = nn.Module()
        self.encoder.layer = nn.ModuleList([_RobertaLayer(hidden_size, intermediate_size, layer_norm_eps)
                                            for _ in range(num_hidden_layers)])
        self.pooler = nn.Module()
        self.pooler.dense = nn.Linear(hidden_size, hidden_size)
        self._cache = None
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, std=initializer_range)
                if isinstance(m, nn.Linear):
                    nn.init.zeros_(m.bias)

    def resize_token_embeddings(self, n):
        old = self.embeddings.word_embeddings
        if n == old.num_embeddings:
            return old
        new = nn.Embedding(n, old.embedding_dim, padding_idx=self.pad).to(old.weight.device)
        k = min(n, old.num_embeddings)
        with torch.no_grad():
            nn.init.normal_(new.weight, std=0.02)
            new.weight[:k] = old.weight[:k]
        self.embeddings.word_embeddings = new
        return new

    def _packed(self, i, layer, device, dtype):
        from .bert_primitives import PackCache
        from . import hip_layers
        if self._cache is None:
            self._cache = PackCache()
        params = layer.ordered_params()
        return self._cache.get(("layer", i, dtype), params,
                               lambda: hip_layers.pack_layer({n: p for n, p in layer.named_parameters()}, "", device, dtype))

    def forward(self, input_ids=None, token_type_ids=None, attention_mask=None, prompt_embeddings=None,
                input_mask=None):
        n, t = input_ids.shape
        dev = input_ids.device
        dtype = torch.float32 if ag.EXACT else torch.bfloat16
        if attention_mask is None:
            attention_mask = (input_ids != self.pad).to(torch.float32)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        # embeddings (table lookups are torch gathers, their backward modcr_embedding_bwd; the LayerNorm is the HIP kernel)
        nonpad = (input_ids != self.pad).to(torch.int64)
        position_ids = torch.cumsum(nonpad, dim=1) * nonpad + self.pad
        e = ag.EmbeddingSumFn.apply(input_ids, position_ids, token_type_ids, self.embeddings.word_embeddings.weight,
                                    self.embeddings.position_embeddings.weight, self.embeddings.token_type_embeddings.weight,
                                    self.embeddings.word_embeddings.padding_idx, self.embeddings.position_embeddings.padding_idx)
        e = ag.LayerNormFn.apply(e.reshape(n * t, self.h), None, self.embeddings.LayerNorm.weight,
                                 self.embeddings.LayerNorm.bias, self.eps).view(n, t, self.h)
        e = ag.dropout(e.contiguous(), self.hidden_p, self.training)        # RobertaEmbeddings.dropout
        mask = attention_mask.to(torch.float32)
        if prompt_embeddings is not None:
            p = prompt_embeddings.shape[1]
            pm = input_mask.to(torch.float32) if input_mask is not None else torch.ones(n, p, device=dev)
            e = torch.cat([e[:, :1], prompt_embeddings.to(e.dtype), e[:, 1:]], dim=1)
            mask = torch.cat([mask[:, :1], pm, mask[:, 1:]], dim=1)
        hidden = e if dtype == torch.float32 else ag.ToBf16Fn.apply(e.contiguous())
        mask = mask.contiguous()
        # (more than 256 rows: layer_forward_train raises -- a skipped dropout is not a failure mode of a drop-in)
        attn_p = self.attn_p if self.training else 0.0
        hid_p = self.hidden_p if self.training else 0.0
        for i, layer in enumerate(self.encoder.layer):
            hidden = ag.BertLayerFn.apply(hidden, mask, None, None, self.a, self.eps, hid_p, attn_p, self._packed(i, layer, dev, dtype),
                                          *layer.ordered_params())
        cls = ag.ToF32Fn.apply(hidden[:, 0].contiguous()) if hidden.dtype != torch.float32 else hidden[:, 0].contiguous()
        pooled = ag.linear(cls, self.pooler.dense.weight, self.pooler.dense.bias, act=mh.ACT_TANH)
        return hidden, pooled
