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


def pack_layer(sd, prefix, device, dtype):
    """Packs one BERT layer's 16 tensors; q/k/v are concatenated to one [3H,H] matrix."""
    p = prefix + "attention.self."
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
