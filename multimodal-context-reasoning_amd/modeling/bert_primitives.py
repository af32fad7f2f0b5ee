"""BERT building blocks with the reference's names and state-dict keys, backed by libmodcr_hip.

Stands in for the vendored a_transformers/modeling_bert.py classes the reference imports
(BertEmbeddings :166-236, BertSelfAttention :238-265, BertSelfOutput :362-373, BertIntermediate
:425-437, BertOutput :440-451, BertPooler :634-646, BertPreTrainedModel :718-743).  The torch
modules here are PARAMETER CONTAINERS: they give checkpoints, optimizers and `state_dict()` the
exact key names of the reference, but no torch op ever runs on them -- every forward goes through
the C ABI (modcr_hip).  There is no CPU path.
"""
import copy
import json
import os

import torch
from torch import nn

import modcr_hip as mh


class BertConfig(object):
    """The attributes the reference reads (run_PMR_ModCR.py:717-748, v10:158-169), plus the knobs of
    this build: modcr_dtype ('bf16' | 'fp32'), modcr_materialize_attentions, modcr_align_map_post_dropout."""

    def __init__(self, **kw):
        d = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                 initializer_range=0.02, layer_norm_eps=1e-12, pad_token_id=0,
                 img_feature_dim=2054, img_feature_type="frcnn", use_img_layernorm=1,
                 img_layer_norm_eps=1e-12, output_attentions=False, output_hidden_states=False,
                 max_hypo=50, add_residual=False, add_local_residual=False,
                 modcr_dtype="bf16", modcr_materialize_attentions=False,
                 # training mode: returned probabilities / the align map are those AFTER the attention dropout (reference semantics)
                 modcr_align_map_post_dropout=True)
        d.update(kw)
        for k, v in d.items():
            setattr(self, k, v)

    @classmethod
    def from_pretrained(cls, path, **kw):
        f = os.path.join(path, "config.json") if os.path.isdir(path) else path
        with open(f) as fh:
            d = json.load(fh)
        d.update(kw)
        return cls(**d)

    def to_dict(self):
        return copy.deepcopy(self.__dict__)


def compute_dtype(config):
    return torch.float32 if getattr(config, "modcr_dtype", "bf16") == "fp32" else torch.bfloat16


class PackCache(object):
    """Device copies of parameters in the layout/dtype the kernels want, rebuilt when a parameter's
    version counter changes (optimizer step, load_state_dict)."""

    def __init__(self):
        self._store = {}

    def get(self, key, params, build):
        sig = tuple((p.data_ptr(), p._version, p.device, p.dtype) for p in params)
        hit = self._store.get(key)
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                hit = (sig, build())
            self._store[key] = hit
        return hit[1]


def _pad64(k):
    """K of a bf16 GEMM operand whose feature dimension is not MFMA-tile aligned (the 2054 region-feature columns): the next
    multiple of 64, or -- beyond 128 -- of 128, which is what the persistent tile kernels require (K % 128 == 0): 2054 -> 2176.
    (Rounds 1-4 padded to 2112 = 33 x 64 and the region embedding ran on the older ring kernel at 0.29 of peak, 2 x 228 us a step.)"""
    return (k + 63) // 64 * 64 if k <= 128 else (k + 127) // 128 * 128


def packed_linear(cache, key, lin, dtype):
    """nn.Linear -> (W [out, in padded to 64] in `dtype`, bias fp32)."""
    params = [lin.weight] + ([lin.bias] if lin.bias is not None else [])

    def build():
        w = lin.weight.detach().float()
        if dtype == torch.bfloat16 and w.shape[1] % 64:
            w = torch.nn.functional.pad(w, (0, _pad64(w.shape[1]) - w.shape[1]))
        b = None if lin.bias is None else lin.bias.detach().float().contiguous()
        return w.to(dtype).contiguous(), b
    return cache.get(key, params, build)


def packed_ln(cache, key, ln):
    return cache.get(key, [ln.weight, ln.bias],
                     lambda: (ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous()))


class BertEmbeddings(nn.Module):
    """a_bert:166-211.  forward(input_ids, token_type_ids, position_ids, out=, seq_stride=) writes
    LN(word + type + pos) into rows [0,T) of each sequence of `out` [N, S, H]."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.eps = config.layer_norm_eps

    def forward(self, input_ids=None, token_type_ids=None, position_ids=None, out=None, dtype=torch.bfloat16, dropout=None):
        """dropout = (p, seed, offset) of the caller's ONE dropout over `out` (text rows here, region rows by embed_regions): a_bert:210"""
        n, t = input_ids.shape
        h = self.word_embeddings.weight.shape[1]
        if out is None:
            out = torch.empty((n, t, h), dtype=dtype, device=input_ids.device)
        if position_ids is not None and position_ids.shape[0] != n:
            position_ids = position_ids.expand(n, t)
        mh.embed_ln(input_ids, token_type_ids, position_ids, self.word_embeddings.weight.detach(),
                    self.position_embeddings.weight.detach(), self.token_type_embeddings.weight.detach(),
                    self.LayerNorm.weight.detach(), self.LayerNorm.bias.detach(), self.eps, out, out.shape[1], dropout=dropout)
        return out


class BertSelfAttention(nn.Module):
    """a_bert:238-265: parameter container for query / key / value."""

    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = config.hidden_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
        self._cache = PackCache()

    def packed_qkv(self, dtype):
        def build():
            w = torch.cat([self.query.weight, self.key.weight, self.value.weight], 0).detach().to(dtype).contiguous()
            b = torch.cat([self.query.bias, self.key.bias, self.value.bias], 0).detach().float().contiguous()
            return w, b
        return self._cache.get(("qkv", dtype), [self.query.weight, self.key.weight, self.value.weight,
                                                self.query.bias, self.key.bias, self.value.bias], build)


def _dropout_residual_ln(hidden_states, w, b, input_tensor, gamma, beta, eps, p, out=None):
    """LN(dropout(dense(h)) + input): training-mode BertSelfOutput / BertOutput (a_bert:369-373, :446-451; dropout is
    live inside the no_grad encoders under model.train()).  One C-ABI call: GEMM -> IEEE-half rows (fp32 on the
    parity path) -> counter-based mask + residual + LayerNorm pass."""
    seed, off = mh.DROPOUT.take(input_tensor.numel())
    return mh.linear_dropout_residual_ln(hidden_states, w, b, input_tensor, gamma, beta, eps, p, seed, off, out=out)


class BertSelfOutput(nn.Module):
    """a_bert:362-373: LN(dense(ctx) + input)."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.eps = config.layer_norm_eps
        self._cache = PackCache()

    def forward(self, hidden_states, input_tensor, workspace=None):
        dt = hidden_states.dtype
        w, b = packed_linear(self._cache, ("w", dt), self.dense, dt)
        g, be = packed_ln(self._cache, "ln", self.LayerNorm)
        if self.training and self.dropout.p > 0.0:
            return _dropout_residual_ln(hidden_states, w, b, input_tensor, g, be, self.eps, self.dropout.p)
        return mh.linear_dropout_residual_ln(hidden_states, w, b, input_tensor, g, be, self.eps)


class BertIntermediate(nn.Module):
    """a_bert:425-437: gelu(dense(x))."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        if config.hidden_act != "gelu":
            raise NotImplementedError("hidden_act=%r: the ModCR checkpoints use erf-GELU" % (config.hidden_act,))
        self._cache = PackCache()

    def forward(self, hidden_states):
        dt = hidden_states.dtype
        w, b = packed_linear(self._cache, ("w", dt), self.dense, dt)
        return mh.linear(hidden_states, w, b, act=mh.ACT_GELU)


class BertOutput(nn.Module):
    """a_bert:440-451: LN(dense(inter) + input)."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.eps = config.layer_norm_eps
        self._cache = PackCache()

    def forward(self, hidden_states, input_tensor, workspace=None, out=None):
        dt = hidden_states.dtype
        w, b = packed_linear(self._cache, ("w", dt), self.dense, dt)
        g, be = packed_ln(self._cache, "ln", self.LayerNorm)
        if self.training and self.dropout.p > 0.0:
            return _dropout_residual_ln(hidden_states, w, b, input_tensor, g, be, self.eps, self.dropout.p, out=out)
        # eval mode: the same two launches with p = 0 (the row pass adds the residual: see modcr_linear_dropout_residual_ln_fwd)
        return mh.linear_dropout_residual_ln(hidden_states, w, b, input_tensor, g, be, self.eps, out=out)


class BertPooler(nn.Module):
    """a_bert:634-646: tanh(dense(h[:, 0]))."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()
        self._cache = PackCache()

    def forward(self, hidden_states):
        dt = hidden_states.dtype
        w, b = packed_linear(self._cache, ("w", dt), self.dense, dt)
        first = hidden_states[:, 0]              # strided view [N, H], row stride S*H
        return mh.linear(first, w, b, act=mh.ACT_TANH)


class BertPreTrainedModel(nn.Module):
    """The slice of HF PreTrainedModel the run scripts use: init_weights, from_pretrained,
    resize_token_embeddings (run_PMR_ModCR.py:727-764)."""
    config_class = BertConfig

    def __init__(self, config):
        super().__init__()
        self.config = config

    def _init_weights(self, module):
        """a_bert:729-743."""
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def init_weights(self):
        self.apply(self._init_weights)

    @classmethod
    def from_pretrained(cls, path, config=None, **kw):
        if config is None:
            config = BertConfig.from_pretrained(path)
        model = cls(config, **kw)
        f = os.path.join(path, "pytorch_model.bin") if os.path.isdir(path) else path
        if os.path.exists(f):
            sd = torch.load(f, map_location="cpu")
            sd = {k[5:] if k.startswith("bert.") else k: v for k, v in sd.items()}
            model.load_state_dict(sd, strict=False)
        return model

    def resize_token_embeddings(self, new_num_tokens):
        old = self.embeddings.word_embeddings
        if new_num_tokens is None or new_num_tokens == old.num_embeddings:
            return old
        new = nn.Embedding(new_num_tokens, old.embedding_dim, padding_idx=old.padding_idx)
        new.to(old.weight.device, dtype=old.weight.dtype)
        self._init_weights(new)
        k = min(old.num_embeddings, new_num_tokens)
        new.weight.data[:k] = old.weight.data[:k]
        self.embeddings.word_embeddings = new
        self.config.vocab_size = new_num_tokens
        return new


def additive_to_binary(mask):
    """The reference hands its layers ADDITIVE masks (0 / -10000).  Kernel masks are 0/1."""
    return (mask == 0).to(torch.float32)


class EncoderOutputs(tuple):
    """Tuple with the reference's positional layout plus `.align_map` (head- and layer-summed
    text->image probabilities of the last three layers, what v10:982 consumes)."""
    align_map = None
