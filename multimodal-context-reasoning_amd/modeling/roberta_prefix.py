"""Stand-in for the reference's prefix-conditioned RoBERTa-large.

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
        pooled = ag.linear(ag.ToBf16Fn.apply(flat), self.dense.weight, self.dense.bias, act=mh.ACT_TANH)
        return None, pooled
