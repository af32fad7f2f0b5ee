"""ctypes binding of libmodcr_hip.so (C ABI: include/modcr_hip.h) + thin torch-tensor wrappers.

PyTorch is plumbing here: it owns device memory and the stream; every piece of arithmetic on the
hot path is a hand-written HIP kernel behind the C ABI.  There is NO fallback: if the shared
library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmodcr_hip.so")

BF16, F32, F16 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_TANH = 0, 1, 2

_c = ctypes
_vp, _i32, _i64, _f32 = _c.c_void_p, _c.c_int32, _c.c_int64, _c.c_float

# name -> (restype, argtypes); mirrors include/modcr_hip.h one to one
SIGNATURES = {
    "modcr_version": (_i32, []),
    "modcr_last_error": (_c.c_char_p, []),
    "modcr_qkv_attn_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32,
                                  _i32, _i32, _i32, _i32, _vp, _i64, _i32, _vp]),
    "modcr_qkv_attn_dropout_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32,
                                          _i32, _i32, _i32, _i32, _f32, _c.c_uint64, _c.c_uint64, _vp, _i64, _i32, _vp]),
    "modcr_qkv_attn_dump_bytes": (_i64, [_i32, _i32, _i32]),
    "modcr_qkv_attn_lse_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32,
                                      _i32, _i32, _i32, _i32, _f32, _c.c_uint64, _c.c_uint64, _vp, _i64, _i32, _vp]),
    "modcr_qkv_attn_opt_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32,
                                      _i32, _i32, _i32, _i32, _f32, _c.c_uint64, _c.c_uint64, _i32, _vp, _i64, _i32, _vp]),
    "modcr_linear_splitk_workspace": (_i64, [_i32, _i32, _i32]),
    "modcr_linear_splitk_fwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "modcr_qkv_attn_workspace": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "modcr_time_next_attn": (_i32, [_vp, _vp]),
    "modcr_chunk_mean_q_fwd": (_i32, [_vp, _i64, _i64, _vp, _i32, _i32, _i32, _i32, _vp]),
    "modcr_build_phase_mask": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "modcr_pack_mask_bits": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "modcr_build_packed_mask": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "modcr_linear_fwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _i32,
                                _i32, _i32, _i32, _i32, _vp]),
    "modcr_ffn_up_gelu_fwd": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "modcr_linear_residual_ln_fwd": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _i32,
                                            _i32, _vp, _i64, _i32, _vp]),
    "modcr_linear_dropout_residual_ln_workspace": (_i64, [_i32, _i32, _i32, _i32]),
    "modcr_linear_dropout_residual_ln_fwd": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _c.c_uint64, _c.c_uint64, _vp, _i64, _i32, _vp]),
    "modcr_proj_residual_ln_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _i32, _vp,
                                          _i64, _i32, _vp]),
    "modcr_ffn_down_residual_ln_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _i32, _i32,
                                              _vp, _i64, _i32, _vp]),
    "modcr_layernorm_fwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _f32, _vp, _i32, _i64, _i32, _i32,
                                   _i64, _vp]),
    "modcr_embed_ln_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _i32, _i32,
                                  _i64, _i32, _i32, _i32, _i32, _vp]),
    "modcr_embed_ln_dropout_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _i32, _i32,
                                          _i64, _i32, _i32, _i32, _i32, _f32, _c.c_uint64, _c.c_uint64, _vp]),
    "modcr_rows_scatter_dropout": (_i32, [_vp, _vp, _i64, _i32, _i32, _i64, _i32, _i32, _f32, _c.c_uint64, _c.c_uint64, _vp]),
    "modcr_cast_pad": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "modcr_convert": (_i32, [_vp, _i32, _vp, _i32, _i64, _vp]),
    "modcr_convert_segments": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "modcr_split3_bf16": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp]),
    "modcr_align_attn_fwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _c.c_uint64, _c.c_uint64,
                                    _vp, _i32, _vp]),
    "modcr_align_attn_bwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32,
                                    _i32, _i32, _f32, _f32, _c.c_uint64, _c.c_uint64, _i32, _vp]),
    "modcr_cls_xattn_fwd": (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _c.c_uint64, _c.c_uint64, _vp]),
    "modcr_cls_xattn_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _i32, _i32, _i32, _f32, _c.c_uint64,
                                   _c.c_uint64, _vp]),
    "modcr_mc_ce_fwd_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "modcr_linear_bwd_input_workspace": (_i64, [_i32, _i32, _i32]),
    "modcr_linear_bwd_input": (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "modcr_linear_bwd_weight_workspace": (_i64, [_i32, _i32, _i32]),
    "modcr_linear_bwd_weight": (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "modcr_layernorm_bwd": (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i64, _i32, _vp]),
    "modcr_act_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "modcr_qkv_attn_bwd_workspace": (_i64, [_i32, _i32, _i32, _i32]),
    "modcr_qkv_attn_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                  _vp, _i64, _i32, _vp]),
    "modcr_qkv_attn_dropout_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                          _f32, _c.c_uint64, _c.c_uint64, _vp, _i32, _vp, _i64, _i32, _vp]),
    "modcr_qkv_attn_lse_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                      _f32, _c.c_uint64, _c.c_uint64, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "modcr_qkv_attn_opt_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                      _f32, _c.c_uint64, _c.c_uint64, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _vp]),
    "modcr_linear_residual_ln_bwd_workspace": (_i64, [_i32, _i32, _i32]),
    "modcr_linear_residual_ln_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32,
                                            _vp, _i64, _i32, _vp]),
    "modcr_linear_residual_ln_dropout_bwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _i64, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _c.c_uint64, _c.c_uint64, _vp, _i64, _i32, _vp]),
    "modcr_layernorm_dropout_bwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _f32, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _c.c_uint64, _c.c_uint64, _vp]),
    "modcr_proj_residual_ln_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _i32, _vp]),
    "modcr_ffn_down_residual_ln_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i64,
                                              _i32, _vp]),
    "modcr_ffn_keep_supported": (_i32, [_i32, _i32, _i32, _i32]),
    "modcr_ffn_up_gelu_keep_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "modcr_ffn_down_gelu_bwd_workspace": (_i64, [_i32, _i32, _i32]),
    "modcr_ffn_down_residual_ln_gelu_bwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _c.c_uint64, _c.c_uint64, _vp, _i64, _i32, _vp]),
    "modcr_ffn_up_du_bwd_workspace": (_i64, [_i32, _i32, _i32]),
    "modcr_ffn_up_du_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _i32, _vp]),
    "modcr_ffn_up_gelu_bwd_workspace": (_i64, [_i32, _i32, _i32]),
    "modcr_ffn_up_gelu_bwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _i32, _vp]),
    "modcr_chunk_mean_q_bwd": (_i32, [_vp, _i64, _i64, _vp, _i32, _i32, _i32, _i32, _vp]),
    "modcr_dropout": (_i32, [_vp, _vp, _i64, _i32, _f32, _c.c_uint64, _c.c_uint64, _vp]),
    "modcr_dropout_residual_ln_fwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _f32, _vp, _i32, _vp, _i32, _i64, _i32, _f32, _c.c_uint64, _c.c_uint64, _vp]),
    "modcr_add": (_i32, [_vp, _vp, _i32, _vp, _i32, _i64, _vp]),
    "modcr_embedding_bwd_v": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _i64, _vp]),
    "modcr_embedding_bwd": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _vp]),
    "modcr_sumsq_f32": (_i32, [_vp, _i64, _vp, _vp]),
    "modcr_sumsq_partials": (_i32, []),
    "modcr_sumsq_f32_ordered": (_i32, [_vp, _i64, _vp, _vp, _i32, _vp]),
    "modcr_adamw_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp]),
    "modcr_adamw_hf_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp]),
}

_lib = None
_product_lib = None
TUNING_LIB_PATH = os.path.join(_HERE, "libmodcr_hip_tuning.so")


class ModcrHipError(RuntimeError):
    pass


def _load(path):
    if not os.path.exists(path):
        raise ModcrHipError(
            "%s not found -- build it with `make -C %s%s` (hipcc, gfx950). "
            "There is no CPU fallback for the ModCR hot path." % (path, os.path.join(_HERE, "..", "csrc"),
                                                                  " tuning" if path == TUNING_LIB_PATH else ""))
    l = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)          # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    return l


def lib():
    """Load libmodcr_hip.so (built by __graft_entry__.build() / csrc/Makefile).  Fails loudly."""
    global _lib, _product_lib
    if _lib is None:
        _product_lib = _load(LIB_PATH)
        _lib = _product_lib
    return _lib


def use_tuning_library(on=True):
    """tools/ and a few tests only: route every call through libmodcr_hip_tuning.so (same sources, -DMODCR_TUNING), the
    build in which the MODCR_* environment knobs of csrc/common.h exist.  The product library reads no environment
    variable.  use_tuning_library(False) switches back."""
    global _lib
    lib()
    _lib = _load(TUNING_LIB_PATH) if on else _product_lib
    return _lib


def is_tuning_library():
    return _lib is not None and _lib is not _product_lib


def _check(rc, what):
    if rc != 0:
        raise ModcrHipError("%s failed (%d): %s" % (what, rc, lib().modcr_last_error().decode()))


def dt_of(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise ModcrHipError("unsupported dtype %s" % t.dtype)


def torch_dtype(dt):
    return {BF16: torch.bfloat16, F32: torch.float32, F16: torch.float16}[dt]


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise ModcrHipError("tensor must live on the GPU (got %s)" % t.device)
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _contig(t, dtype=None):
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------
SPLITK = True        # tools may clear it for an A/B run: few-row GEMMs through the split-K entry


def linear(a, w, bias=None, act=ACT_NONE, residual=None, out_dtype=None, out=None):
    """act(a @ w.T + bias) (+ residual).  a [..., K] (last dim contiguous, uniform row stride),
    w [N, K].  Returns [..., N]."""
    dt = dt_of(w)
    if a.dtype != w.dtype:
        raise ModcrHipError("linear: activation %s vs weight %s" % (a.dtype, w.dtype))
    k = a.shape[-1]
    if a.dim() == 2 and a.stride(1) == 1:
        a2, lda = a, a.stride(0)
    else:
        a2 = a.reshape(-1, k)
        a2 = _contig(a2)
        lda = k
    m, n = a2.shape[0], w.shape[0]
    w = _contig(w)
    od = dt if out_dtype is None else out_dtype
    if out is None:
        out = torch.empty((m, n), dtype=torch_dtype(od), device=a.device)
    res2 = None
    if residual is not None:
        res2 = _contig(residual.reshape(-1, n))
    if dt == BF16 and res2 is None and 256 <= m <= 1024 and SPLITK and od != F16:
        need = lib().modcr_linear_splitk_workspace(m, n, k)         # few-row GEMMs of the heads: split-K over the chip
        if need:
            ws = _workspace("lin_splitk", need, a.device)
            _check(lib().modcr_linear_splitk_fwd(_ptr(a2), lda, _ptr(w), w.stride(0), _ptr(bias), _ptr(out), out.stride(0), m, n, k,
                                                 act, od, _ptr(ws), need, _stream()), "modcr_linear_splitk_fwd")
            return out.view(*a.shape[:-1], n) if (a.dim() != 2) else out
    _check(lib().modcr_linear_fwd(_ptr(a2), lda, _ptr(w), w.stride(0), _ptr(bias), _ptr(res2), n,
                                  dt_of(res2) if res2 is not None else 0, _ptr(out), out.stride(0), m, n,
                                  k, act, dt, od, _stream()), "modcr_linear_fwd")
    return out.view(*a.shape[:-1], n) if (a.dim() != 2) else out


def layernorm(x, gamma, beta, eps, residual=None, out_dtype=None, out=None, rows_per_group=0,
              group_stride=0):
    h = x.shape[-1]
    x2 = _contig(x.reshape(-1, h))
    m = x2.shape[0]
    od = dt_of(x2) if out_dtype is None else out_dtype
    ret = out
    if out is None:
        out = torch.empty((m, h), dtype=torch_dtype(od), device=x.device)
        ret = out.view(x.shape)
    r2 = _contig(residual.reshape(-1, h)) if residual is not None else None
    _check(lib().modcr_layernorm_fwd(_ptr(x2), dt_of(x2), _ptr(r2), dt_of(r2) if r2 is not None else 0,
                                     _ptr(gamma), _ptr(beta), float(eps), _ptr(out), od, m, h,
                                     rows_per_group, group_stride, _stream()), "modcr_layernorm_fwd")
    return ret


def _same_dtype(what, w, **tensors):
    """the C entries take ONE dtype code for the activations, the residual and the output: a tensor of another dtype would be read
    through the wrong element size (garbage, or reads past its end) and the C side cannot tell"""
    for name, t in tensors.items():
        if t is not None and t.dtype != w.dtype:
            raise ValueError("%s: %s is %s but the weights are %s" % (what, name, t.dtype, w.dtype))


def linear_residual_ln(a, w, bias, residual, gamma, beta, eps, workspace=None, out=None):
    """LN(a @ w.T + bias + residual): BertSelfOutput / BertOutput."""
    dt = dt_of(w)
    _same_dtype("linear_residual_ln", w, a=a, residual=residual, out=out)
    k = a.shape[-1]
    a2 = _contig(a.reshape(-1, k))
    m, n = a2.shape[0], w.shape[0]
    r2 = _contig(residual.reshape(-1, n))
    if workspace is None or workspace.numel() * workspace.element_size() < m * n * 4:
        workspace = torch.empty((m, n), dtype=torch.float32, device=a.device)
    if out is None:
        out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    _check(lib().modcr_linear_residual_ln_fwd(_ptr(a2), k, _ptr(_contig(w)), _ptr(bias), _ptr(r2), _ptr(gamma),
                                              _ptr(beta), float(eps), _ptr(out), m, n, k, _ptr(workspace),
                                              workspace.numel() * workspace.element_size(), dt, _stream()),
           "modcr_linear_residual_ln_fwd")
    return out.view(*residual.shape)


def linear_dropout_residual_ln(a, w, bias, residual, gamma, beta, eps, p=0.0, seed=0, offset=0, out=None, pre_out=None):
    """LN(dropout(a @ w.T + bias) + residual): BertSelfOutput / BertOutput as one C-ABI call (GEMM -> IEEE-half rows -> mask +
    residual + LayerNorm pass).  p = 0: no dropout.  pre_out: fp32 or fp16 [M,N] tensor that receives the pre-LayerNorm rows."""
    dt = dt_of(w)
    _same_dtype("linear_dropout_residual_ln", w, a=a, residual=residual, out=out)
    k = a.shape[-1]
    a2 = _contig(a.reshape(-1, k))
    m, n = a2.shape[0], w.shape[0]
    r2 = _contig(residual.reshape(-1, n))
    need = lib().modcr_linear_dropout_residual_ln_workspace(m, n, k, dt)
    ws = _workspace("lin_ln", need, a.device)
    if out is None:
        out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    if pre_out is not None and (pre_out.dtype not in (torch.float32, torch.float16) or pre_out.numel() != m * n or not pre_out.is_contiguous()):
        raise ValueError("linear_dropout_residual_ln: pre_out must be a contiguous fp32 or fp16 [M,N] tensor")
    _check(lib().modcr_linear_dropout_residual_ln_fwd(_ptr(a2), k, _ptr(_contig(w)), _ptr(bias), _ptr(r2), _ptr(gamma), _ptr(beta), float(eps),
                                                      _ptr(out), _ptr(pre_out), dt_of(pre_out) if pre_out is not None else F32, m, n, k, float(p), seed, offset,
                                                      _ptr(ws), need, dt, _stream()),
           "modcr_linear_dropout_residual_ln_fwd")
    return out.view(*residual.shape)


def pack_mask_bits(mask):
    """0/1 float mask [..., L] -> uint32 bits [..., ceil(L/32)] (as int32 storage)."""
    l = mask.shape[-1]
    m2 = _contig(mask.reshape(-1, l), torch.float32)
    bits = torch.empty((m2.shape[0], (l + 31) // 32), dtype=torch.int32, device=mask.device)
    _check(lib().modcr_pack_mask_bits(_ptr(m2), _ptr(bits), m2.shape[0], l, _stream()), "modcr_pack_mask_bits")
    return bits.view(*mask.shape[:-1], (l + 31) // 32)


def pack_factor(n, s, limit=192):
    """how many sequences of s rows share one attention row block of at most `limit` rows (a divisor of n; 1 = no packing)"""
    for k in range(limit // max(s, 1), 1, -1):
        if n % k == 0:
            return k
    return 1


def build_packed_mask(key_mask, k):
    """block-diagonal dense mask bits of k sequences per row block: key_mask [N, S] 0/1 -> bits [N / k, k S, ceil(k S / 32)]"""
    n, s = key_mask.shape
    km = _contig(key_mask, torch.float32)
    bits = torch.empty((n // k, k * s, (k * s + 31) // 32), dtype=torch.int32, device=key_mask.device)
    _check(lib().modcr_build_packed_mask(_ptr(km), _ptr(bits), n, s, k, _stream()), "modcr_build_packed_mask")
    return bits


def build_phase_mask(input_mask, chunk_mask, phase):
    n, s = input_mask.shape
    t = chunk_mask.shape[1]
    bits = torch.empty((n, s, (s + 31) // 32), dtype=torch.int32, device=input_mask.device)
    _check(lib().modcr_build_phase_mask(_ptr(_contig(input_mask, torch.float32)),
                                        _ptr(_contig(chunk_mask, torch.float32)), _ptr(bits), n, t, s - t,
                                        phase, _stream()), "modcr_build_phase_mask")
    return bits


ATTN_SIDE_POST_DROPOUT = 1      # include/modcr_hip.h: MODCR_ATTN_SIDE_POST_DROPOUT


def qkv_attn(x, wqkv, bqkv, key_mask=None, mask_bits=None, hist=None, chunk_id=None, want_probs=False,
             align_map=None, align_t=0, num_heads=None, workspace=None, out=None, attn_dropout=None, lse=None, dump=None,
             side_post_dropout=False):
    """Fused QKV projection + attention.  x [N,S,H]; returns (ctx [N,S,H], probs or None).
    attn_dropout = (p, seed, offset): training-mode dropout of the attention probabilities.
    side_post_dropout: probs / align_map are the probabilities AFTER that dropout, P o m / (1 - p), as the reference's modules
    return them (modeling_bert.py:69-74); default: the un-dropped ones (and no probs output under dropout).
    lse: fp32 [N, A, S] tensor that receives the row statistics qkv_attn_bwd(ctx=, lse=) wants (tile-kernel shapes only:
    lse_supported); dump: bf16 tensor of qkv_dump_numel(N, S, A) elements that receives the Q | K | V images (with lse)."""
    dt = dt_of(x)
    x = _contig(x)
    n, s, h = x.shape
    a = num_heads
    p = 0 if hist is None else hist.shape[1]
    if (dt == BF16 and attn_dropout is not None and attn_dropout[0] > 0.0 and side_post_dropout and (want_probs or align_map is not None)
            and not side_outputs_on_tiles(s, p, h, a)):
        # post-dropout side outputs are served by the bf16 TILE kernels only (modcr_qkv_attn_opt_fwd returns MODCR_ERR_UNSUPPORTED on
        # the older kernel's shapes: P + S <= 64, odd head counts below 193 rows, H not a multiple of 128, prefix rows): such a call
        # takes the exact-fp32 route, which carries the same mask at any length <= 256, and hands the context back in bf16
        if lse is not None or dump is not None:
            raise ValueError("qkv_attn: no row statistics on the exact-fp32 route this shape takes (S=%d P=%d A=%d H=%d)" % (s, p, a, h))
        ctx32, probs = qkv_attn(x.float(), wqkv.float(), bqkv, key_mask=key_mask, mask_bits=mask_bits,
                                hist=None if hist is None else hist.float(), chunk_id=chunk_id, want_probs=want_probs,
                                align_map=align_map, align_t=align_t, num_heads=a, attn_dropout=attn_dropout, side_post_dropout=True)
        if out is None:
            return ctx32.to(torch.bfloat16), probs
        out.copy_(ctx32)
        return out, probs
    hist = _contig(hist)
    ctx = torch.empty_like(x) if out is None else out
    probs = torch.empty((n, a, s, p + s), dtype=torch.float32, device=x.device) if want_probs else None
    need = lib().modcr_qkv_attn_workspace(n, s, p, h, dt)
    if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
        workspace = torch.empty((need // 4,), dtype=torch.float32, device=x.device)
    km = _contig(key_mask, torch.float32) if key_mask is not None else None
    chunk_t = 0 if chunk_id is None else chunk_id.shape[1]
    ap, seed, off = attn_dropout if attn_dropout is not None else (0.0, 0, 0)
    if lse is not None and (lse.dtype != torch.float32 or tuple(lse.shape) != (n, a, s) or not lse.is_contiguous()):
        raise ValueError("qkv_attn: lse must be a contiguous fp32 [N, A, S] tensor")
    if dump is not None and (dump.dtype != torch.bfloat16 or dump.numel() != qkv_dump_numel(n, s, a) or not dump.is_contiguous() or lse is None):
        raise ValueError("qkv_attn: dump must be a contiguous bf16 tensor of qkv_dump_numel(N, S, A) elements, given together with lse")
    _check(lib().modcr_qkv_attn_opt_fwd(_ptr(x), _ptr(hist), _ptr(_contig(wqkv)), _ptr(bqkv), _ptr(km), _ptr(mask_bits),
                                        _ptr(chunk_id), chunk_t, _ptr(ctx), _ptr(probs), _ptr(align_map), align_t, _ptr(lse), _ptr(dump),
                                        n, s, p, h, a, float(ap), seed, off, ATTN_SIDE_POST_DROPOUT if side_post_dropout else 0,
                                        _ptr(workspace) if need else None, need, dt, _stream()),
           "modcr_qkv_attn_fwd")
    return ctx, probs


def side_outputs_on_tiles(s, p, h, a):
    """shapes on which the bf16 TILE kernels serve a probabilities output / an align map (the dispatcher of modcr_qkv_attn_opt_fwd,
    csrc/attn.hip: 64 < S <= 256 without prefix rows, H a multiple of 128, head pairs up to 192 rows, one head per workgroup above)"""
    return p == 0 and 64 < s <= 256 and h % 128 == 0 and 256 <= h <= 8192 and (a % 2 == 0 or s > 192)


def qkv_dump_numel(n, s, a):
    """bf16 elements of qkv_attn's `dump` output: [N][A][3][LP][64]"""
    return lib().modcr_qkv_attn_dump_bytes(n, s, a) // 2


def lse_supported(x, num_heads, hist=None):
    """shapes on which qkv_attn runs a tile kernel (the ones that can write the row statistics `lse`)"""
    n, s, h = x.shape
    return (x.dtype == torch.bfloat16 and hist is None and 64 < s <= 192 and num_heads % 2 == 0 and h % 128 == 0 and h >= 256
            and h == num_heads * 64)


def embed_ln(input_ids, token_type_ids, position_ids, word, pos, typ, gamma, beta, eps, out, seq_stride, dropout=None):
    """dropout = (p, seed, offset): BertEmbeddings.dropout on the rows written, counters = flat indices of `out` [N, seq_stride, H]"""
    n, t = input_ids.shape
    h = word.shape[1]
    p, seed, off = dropout if dropout is not None else (0.0, 0, 0)
    _check(lib().modcr_embed_ln_dropout_fwd(_ptr(_contig(input_ids)), _ptr(_contig(token_type_ids)),
                                            _ptr(_contig(position_ids)), _ptr(word), _ptr(pos), _ptr(typ), _ptr(gamma),
                                            _ptr(beta), float(eps), _ptr(out), n, t, h, seq_stride, word.shape[0],
                                            pos.shape[0], typ.shape[0], dt_of(out), float(p), seed, off, _stream()), "modcr_embed_ln_dropout_fwd")
    return out


def rows_scatter_dropout(src, dst, row0, dropout=None):
    """src [N*R, H] -> dst[:, row0:row0+R] of the contiguous dst [N, S, H] under nn.Dropout (dropout = (p, seed, offset), None = plain
    copy); counters = flat indices of dst, so together with embed_ln(..., dropout=) the buffer carries the mask of ONE dropout call over it."""
    n, s, h = dst.shape
    assert dst.is_contiguous() and src.is_contiguous() and src.dtype == dst.dtype and src.numel() % (n * h) == 0
    r = src.numel() // (n * h)
    assert row0 + r <= s
    p, seed, off = dropout if dropout is not None else (0.0, 0, 0)
    _check(lib().modcr_rows_scatter_dropout(_ptr(src), _ptr(dst), n * r, h, r, s, row0, dt_of(dst), float(p), seed, off, _stream()),
           "modcr_rows_scatter_dropout")
    return dst


def cast_pad(src, kp, dtype):
    k = src.shape[-1]
    s2 = _contig(src.reshape(-1, k), torch.float32)
    dst = torch.empty((s2.shape[0], kp), dtype=torch_dtype(dtype), device=src.device)
    _check(lib().modcr_cast_pad(_ptr(s2), k, _ptr(dst), kp, s2.shape[0], k, kp, dtype, _stream()), "modcr_cast_pad")
    return dst


def split3(src, mode):
    """fp32 [M,K] -> bf16 [M,3K]: mode 0 = [hi|lo|hi] (activations), 1 = [hi|hi|lo] (weights)."""
    src = _contig(src, torch.float32)
    m, k = src.shape
    dst = torch.empty((m, 3 * k), dtype=torch.bfloat16, device=src.device)
    _check(lib().modcr_split3_bf16(_ptr(src), k, _ptr(dst), 3 * k, m, k, mode, _stream()), "modcr_split3_bf16")
    return dst


def convert(src, dtype):
    src = _contig(src)
    dst = torch.empty(src.shape, dtype=torch_dtype(dtype), device=src.device)
    _check(lib().modcr_convert(_ptr(src), dt_of(src), _ptr(dst), dtype, src.numel(), _stream()), "modcr_convert")
    return dst


def convert_segments(pairs):
    """[(src, dst), ...]: dst[:] = src (element-wise fp32 <-> bf16 conversions or copies) for all pairs in ONE launch per eight pairs
    (modcr_convert_segments).  All sources share a dtype and all destinations share a dtype; tensors contiguous, numel equal per pair."""
    if not pairs:
        return
    sdt, ddt = dt_of(pairs[0][0]), dt_of(pairs[0][1])
    for src, dst in pairs:
        if dt_of(src) != sdt or dt_of(dst) != ddt or src.numel() != dst.numel() or not (src.is_contiguous() and dst.is_contiguous()):
            raise ValueError("convert_segments: pairs must be contiguous, of equal size, with one source and one destination dtype")
    k = len(pairs)
    srcs = (_c.c_void_p * k)(*[p_[0].data_ptr() for p_ in pairs])
    dsts = (_c.c_void_p * k)(*[p_[1].data_ptr() for p_ in pairs])
    ns = (_c.c_int64 * k)(*[p_[0].numel() for p_ in pairs])
    _check(lib().modcr_convert_segments(srcs, dsts, ns, k, sdt, ddt, _stream()), "modcr_convert_segments")


def align_attn(q, k, v, heads, scale=1.0, want_probs=False, dropout=None, key_bias=None):
    """q [N,E] fp32, k/v [N,L,E] (bf16 or fp32) -> out [N,E] fp32, probs [N,heads,L] (unmasked softmax) or None.
    dropout = (p, seed, offset): training-mode dropout of the attention weights."""
    n, l, e = k.shape
    q, k, v = _contig(q, torch.float32), _contig(k), _contig(v)
    out = torch.empty_like(q)
    probs = torch.empty((n, heads, l), dtype=torch.float32, device=q.device) if want_probs else None
    p, seed, off = dropout if dropout is not None else (0.0, 0, 0)
    _check(lib().modcr_align_attn_fwd(_ptr(q), _ptr(k), _ptr(v), e, _ptr(out), _ptr(probs), n, l, e, heads,
                                      float(scale), float(p), seed, off,
                                      _ptr(_contig(key_bias, torch.float32)) if key_bias is not None else None, dt_of(k), _stream()),
           "modcr_align_attn_fwd")
    return out, probs


def align_attn_bwd(dout, q, k, v, probs, heads, scale=1.0, dropout=None):
    n, l, e = k.shape
    dout, q = _contig(dout, torch.float32), _contig(q, torch.float32)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    p, seed, off = dropout if dropout is not None else (0.0, 0, 0)
    _check(lib().modcr_align_attn_bwd(_ptr(dout), _ptr(q), _ptr(k), _ptr(v), e, _ptr(probs), _ptr(dq),
                                      _ptr(dk), _ptr(dv), e, n, l, e, heads, float(scale), float(p), seed, off, dt_of(k),
                                      _stream()),
           "modcr_align_attn_bwd")
    return dq, dk, dv


def _row_blocks(blocks):
    """host arrays of modcr_cls_xattn_*: [N, rows_i, E] bf16 views (last dim contiguous, one row stride for all) -> ctypes arrays"""
    n, _, e = blocks[0].shape
    ld = blocks[0].stride(1)
    for t in blocks:
        if t.dtype != torch.bfloat16 or t.dim() != 3 or t.shape[0] != n or t.shape[2] != e or t.stride(2) != 1 or t.stride(1) != ld:
            raise ValueError("cls_xattn: row blocks must be bf16 [N, rows, E] views with contiguous rows and one row stride")
    k = len(blocks)
    ptrs = (_c.c_void_p * k)(*[t.data_ptr() for t in blocks])
    strides = (_c.c_int64 * k)(*[t.stride(0) for t in blocks])
    rows = (_c.c_int32 * k)(*[t.shape[1] for t in blocks])
    return ptrs, strides, rows, k, ld, n, e, sum(t.shape[1] for t in blocks)


def cls_xattn(qt, blocks, heads, dropout=None):
    """Reassociated single-query cross-attention over frozen bf16 states (include/modcr_hip.h: modcr_cls_xattn_fwd).
    qt [N,heads,E] fp32; blocks: 1-3 bf16 [N,rows,E] views.  Returns (ctx [N,heads,E], ssum [N,heads], probs [N,heads,L])."""
    ptrs, strides, rows, k, ld, n, e, l = _row_blocks(blocks)
    qt = _contig(qt, torch.float32)
    ctx = torch.empty((n, heads, e), dtype=torch.float32, device=qt.device)
    ssum = torch.empty((n, heads), dtype=torch.float32, device=qt.device)
    probs = torch.empty((n, heads, l), dtype=torch.float32, device=qt.device)
    p, seed, off = dropout if dropout is not None else (0.0, 0, 0)
    _check(lib().modcr_cls_xattn_fwd(_ptr(qt), ptrs, strides, rows, k, ld, _ptr(ctx), _ptr(ssum), _ptr(probs), n, e, heads,
                                     float(p), seed, off, _stream()), "modcr_cls_xattn_fwd")
    return ctx, ssum, probs


def cls_xattn_bwd(dctx, dssum, ctx, ssum, probs, blocks, heads, dropout=None):
    ptrs, strides, rows, k, ld, n, e, l = _row_blocks(blocks)
    dctx, dssum = _contig(dctx, torch.float32), _contig(dssum, torch.float32)
    dqt = torch.empty_like(ctx)
    p, seed, off = dropout if dropout is not None else (0.0, 0, 0)
    _check(lib().modcr_cls_xattn_bwd(_ptr(dctx), _ptr(dssum), _ptr(ctx), _ptr(ssum), _ptr(probs), ptrs, strides, rows, k, ld,
                                     _ptr(dqt), n, e, heads, float(p), seed, off, _stream()), "modcr_cls_xattn_bwd")
    return dqt


def mc_ce(logits, label, want_grad=True, want_loss=True, grad_scale=None):
    """Soft-label CE over [B,C]; returns (loss scalar tensor or None, dlogits or None).
    grad_scale: 0-dim/1-element fp32 GPU tensor holding d(total)/d(loss), read on the device."""
    logits = _contig(logits, torch.float32)
    label = _contig(label, torch.float32)
    b, c = logits.shape
    loss = torch.empty((), dtype=torch.float32, device=logits.device) if want_loss else None
    dl = torch.empty_like(logits) if want_grad else None
    gs = _contig(grad_scale, torch.float32) if grad_scale is not None else None
    _check(lib().modcr_mc_ce_fwd_bwd(_ptr(logits), _ptr(label), _ptr(loss), _ptr(dl), _ptr(gs), b, c, _stream()),
           "modcr_mc_ce_fwd_bwd")
    return loss, dl


_scratch = {}


def _workspace(key, nbytes, device):
    """caller-owned scratch for the MFMA backward routes (grown on demand, reused across calls)"""
    b = _scratch.get((key, device))
    if b is None or b.numel() < nbytes:
        b = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        _scratch[(key, device)] = b
    return b


def linear_bwd_input(dy, w, out_dtype=F32, mfma=True):
    """dX = dY @ W.  dY [M,N] fp32 or bf16, W [N,K] (fp32 or bf16).  mfma=False: exact-fp32 kernel."""
    dy = _contig(dy)
    m, n = dy.shape
    k = w.shape[1]
    dx = torch.empty((m, k), dtype=torch_dtype(out_dtype), device=dy.device)
    ws, nb = None, 0
    if mfma and n >= 64 and (dy.dtype == torch.float32 or n % 64 == 0):
        nb = lib().modcr_linear_bwd_input_workspace(m, n, k)
        ws = _workspace("bwd_in", nb, dy.device)
    _check(lib().modcr_linear_bwd_input(_ptr(dy), n, dt_of(dy), _ptr(_contig(w)), k, _ptr(dx), k, m, n, k, dt_of(w),
                                        out_dtype, _ptr(ws), nb, _stream()), "modcr_linear_bwd_input")
    return dx


def linear_bwd_weight(dy, x, dw, db=None, accumulate=False, mfma=True):
    """dW (+)= dY^T @ X, db (+)= colsum(dY).  dY [M,N] fp32/bf16, X [M,K] fp32/bf16, dW fp32 [N,K]."""
    dy = _contig(dy)
    x = _contig(x)
    m, n = dy.shape
    k = x.shape[1]
    ws, nb = None, 0
    if mfma:
        nb = lib().modcr_linear_bwd_weight_workspace(m, n, k)
        ws = _workspace("bwd_w", nb, dy.device)
    _check(lib().modcr_linear_bwd_weight(_ptr(dy), n, dt_of(dy), _ptr(x), k, _ptr(dw), _ptr(db), m, n, k,
                                         1 if accumulate else 0, dt_of(x), _ptr(ws), nb, _stream()),
           "modcr_linear_bwd_weight")
    return dw, db


def layernorm_bwd(dy, x, gamma, eps, dgamma=None, dbeta=None, residual=None):
    dy, x = _contig(dy, torch.float32), _contig(x, torch.float32)
    residual = _contig(residual, torch.float32) if residual is not None else None
    m, h = x.shape
    dx = torch.empty_like(x)
    _check(lib().modcr_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(residual), _ptr(gamma), float(eps), _ptr(dx),
                                     _ptr(dgamma), _ptr(dbeta), m, h, _stream()), "modcr_layernorm_bwd")
    return dx


def act_bwd(dact, pre, act):
    dact, pre = _contig(dact, torch.float32), _contig(pre, torch.float32)
    out = torch.empty_like(pre)
    _check(lib().modcr_act_bwd(_ptr(dact), _ptr(pre), _ptr(out), pre.numel(), act, _stream()), "modcr_act_bwd")
    return out


def sumsq_accumulate(x, out, partials=None):
    """out (fp32 [1], zeroed by the caller) += sum(x^2).  With `partials` (fp32 [>= 1] workspace, sumsq_partials() entries are
    used at most) the sum is formed in a fixed order (bit-reproducible: the clip coefficient of data-parallel replicas)."""
    if partials is None:
        _check(lib().modcr_sumsq_f32(_ptr(x), x.numel(), _ptr(out), _stream()), "modcr_sumsq_f32")
    else:
        if partials.dtype != torch.float32 or not partials.is_contiguous():
            raise ValueError("sumsq_accumulate: partials must be a contiguous fp32 tensor")
        _check(lib().modcr_sumsq_f32_ordered(_ptr(x), x.numel(), _ptr(out), _ptr(partials), partials.numel(), _stream()),
               "modcr_sumsq_f32_ordered")


def sumsq_partials():
    return int(lib().modcr_sumsq_partials())


def adamw_step(p, g, m, v, sumsq, max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2, form="hf"):
    """clip by the global norm in `sumsq` (device scalar) + AdamW update, in place on flat fp32 buffers.
    form "hf" = transformers.AdamW (what the reference trains with), "torch" = torch.optim.AdamW."""
    fn = {"hf": lib().modcr_adamw_hf_step, "torch": lib().modcr_adamw_step}[form]
    _check(fn(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(sumsq), float(max_norm), float(lr),
              float(beta1), float(beta2), float(eps), float(weight_decay), float(bc1), float(bc2),
              _stream()), "modcr_adamw_%s_step" % form)


def qkv_attn_bwd(dctx, x, wqkv, bqkv, dwqkv, dbqkv, key_mask=None, mask_bits=None, chunk_id=None, num_heads=None,
                 accumulate=False, attn_dropout=None, d_align=None, align_t=0, dx_residual=None, ctx=None, lse=None, dump=None,
                 side_post_dropout=False):
    """Backward of qkv_attn (no prefix rows): returns dx [N,S,H] in x's dtype; dwqkv [3H,H] / dbqkv [3H] fp32 are
    written (or added into when accumulate).  attn_dropout = the (p, seed, offset) the forward ran with; d_align [N,T,R] =
    gradient of the align map the forward accumulated (align_t = T); dx_residual (fp32, x's shape) is added to dx in the
    epilogue of its GEMM.  ctx + lse: the forward's context rows and the row statistics qkv_attn(lse=) wrote -- the attention
    core then runs as the five-product kernel (csrc/attn_bwd.hip) instead of recomputing the statistics."""
    dt = dt_of(x)
    x, dctx = _contig(x), _contig(dctx)
    if (ctx is None) != (lse is None):
        raise ValueError("qkv_attn_bwd: ctx and lse come together")
    if ctx is not None and (ctx.dtype != x.dtype or ctx.shape != x.shape or not ctx.is_contiguous()):
        raise ValueError("qkv_attn_bwd: ctx must be a contiguous tensor of x's shape and dtype")

    n, s, h = x.shape
    if dump is not None and (lse is None or dump.dtype != torch.bfloat16 or dump.numel() != qkv_dump_numel(n, s, num_heads)):
        raise ValueError("qkv_attn_bwd: dump is qkv_attn's dump output of the same call shape, given with ctx and lse")
    dx = torch.empty_like(x)
    need = lib().modcr_qkv_attn_bwd_workspace(n, s, h, dt)
    ws = _workspace("attn_bwd", need, x.device)
    km = _contig(key_mask, torch.float32) if key_mask is not None else None
    chunk_t = 0 if chunk_id is None else chunk_id.shape[1]
    ap, seed, off = attn_dropout if attn_dropout is not None else (0.0, 0, 0)
    _check(lib().modcr_qkv_attn_opt_bwd(_ptr(dctx), _ptr(x), _ptr(_contig(wqkv)), _ptr(bqkv), _ptr(km), _ptr(mask_bits),
                                        _ptr(chunk_id), chunk_t,
                                        _ptr(_contig(dx_residual, torch.float32)) if dx_residual is not None else None,
                                        _ptr(dx), _ptr(dwqkv), _ptr(dbqkv), 1 if accumulate else 0,
                                        n, s, h, num_heads, float(ap), seed, off,
                                        _ptr(_contig(d_align, torch.float32)) if d_align is not None else None, int(align_t),
                                        _ptr(ctx), _ptr(lse), _ptr(dump), ATTN_SIDE_POST_DROPOUT if side_post_dropout else 0,
                                        _ptr(ws), need, dt, _stream()),
           "modcr_qkv_attn_bwd")
    return dx


def add(a, b, out_dtype=F32):
    """a (fp32) + b (fp32 or bf16) -> out_dtype"""
    a = _contig(a, torch.float32)
    b = _contig(b)
    out = torch.empty(a.shape, dtype=torch_dtype(out_dtype), device=a.device)
    _check(lib().modcr_add(_ptr(a), _ptr(b), dt_of(b), _ptr(out), out_dtype, a.numel(), _stream()), "modcr_add")
    return out


def embedding_bwd(ids, dy, dw, padding_idx=None):
    """dw[id] += sum of the rows of dy [M,H] fp32 whose id it is (ids int64, any shape with M elements); rows of padding_idx are
    skipped.  Deterministic (sorted-segment sums, one writer per table row).  Tables of at most four rows (token types) are reduced
    row by row instead: a segment of M / 2 rows is no work for one workgroup."""
    dy = _contig(dy, torch.float32)
    m, h = dy.shape
    flat = ids.reshape(-1)
    if flat.numel() != m:
        raise ValueError("embedding_bwd: %d ids for %d gradient rows" % (flat.numel(), m))
    if dw.dtype != torch.float32 or not dw.is_contiguous() or dw.shape[1] != h:
        raise ValueError("embedding_bwd: dw must be a contiguous fp32 [V,%d] tensor" % h)
    if dw.shape[0] <= 4:
        for v in range(dw.shape[0]):
            if padding_idx is not None and v == padding_idx:
                continue
            dw[v] += (dy * (flat == v).to(torch.float32)[:, None]).sum(0)
        return dw
    sid, order = torch.sort(flat, stable=True)
    _check(lib().modcr_embedding_bwd_v(_ptr(sid), _ptr(order), _ptr(dy), _ptr(dw), m, h, dw.shape[0],
                                       -1 if padding_idx is None else int(padding_idx), _stream()), "modcr_embedding_bwd")
    return dw


def _grad_out(t, shape, device, what):
    """a caller-provided fp32 gradient tensor (a view into the flat gradient buffer: written in place, no autograd add) or a fresh one"""
    if t is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    if t.dtype != torch.float32 or tuple(t.shape) != tuple(shape) or not t.is_contiguous() or t.data_ptr() % 16:
        raise ValueError("%s: the output gradient must be a contiguous, 16-byte aligned fp32 tensor of shape %s" % (what, tuple(shape)))
    return t


def linear_residual_ln_bwd(dy, pre, a, w, gamma, eps, dgamma, dbeta, dropout=None, dw_out=None, db_out=None):
    """backward of LN(dropout(a @ w.T + bias) + residual) from the saved pre-LN rows (fp32, or fp16 on the bf16 route): returns
    (d_pre fp32 [M,N] = gradient of the residual branch, da [M,K] in a's dtype, dw fp32, dbias fp32); dgamma / dbeta are accumulated.
    dy fp32 or bf16; dropout = (p, seed, offset) of the forward or None (bf16 route only)."""
    dy, pre = _contig(dy), (_contig(pre) if pre.dtype == torch.float16 else _contig(pre, torch.float32))
    a, w = _contig(a), _contig(w)
    m, n = pre.shape
    k = a.shape[1]
    dt = dt_of(a)
    d_pre = torch.empty(pre.shape, dtype=torch.float32, device=pre.device)
    da = torch.empty_like(a)
    dw = _grad_out(dw_out, (n, k), a.device, "linear_residual_ln_bwd dw")
    db = _grad_out(db_out, (n,), a.device, "linear_residual_ln_bwd db")
    need = lib().modcr_linear_residual_ln_bwd_workspace(m, n, k) if dt == BF16 else 0
    ws = _workspace("lrl_bwd", need, a.device) if need else None
    p, seed, off = dropout if dropout is not None else (0.0, 0, 0)
    _check(lib().modcr_linear_residual_ln_dropout_bwd(_ptr(dy), dt_of(dy), _ptr(pre), dt_of(pre), _ptr(a), k, _ptr(w), _ptr(gamma), float(eps),
                                                      _ptr(d_pre), _ptr(da), _ptr(dw), _ptr(db), _ptr(dgamma), _ptr(dbeta), m, n, k,
                                                      float(p), seed, off, _ptr(ws), need, dt, _stream()),
           "modcr_linear_residual_ln_dropout_bwd")
    return d_pre, da, dw, db


def ffn_up_gelu_bwd(dinter, x, w1, b1, dx_residual=None, dw_out=None, db_out=None):
    """backward of gelu(x @ w1.T + b1): returns (dx fp32 [M,H] (+ dx_residual fp32 [M,H], added in the GEMM's epilogue),
    dw1 fp32, db1 fp32)"""
    dinter, x, w1 = _contig(dinter), _contig(x), _contig(w1)
    m, h = x.shape
    i = w1.shape[0]
    dt = dt_of(x)
    dx = torch.empty((m, h), dtype=torch.float32, device=x.device)
    dw = _grad_out(dw_out, (i, h), x.device, "ffn_up_gelu_bwd dw1")
    db = _grad_out(db_out, (i,), x.device, "ffn_up_gelu_bwd db1")
    need = lib().modcr_ffn_up_gelu_bwd_workspace(m, h, i)
    ws = _workspace("ffn_up_bwd", need, x.device)
    _check(lib().modcr_ffn_up_gelu_bwd(_ptr(dinter), dt_of(dinter), _ptr(x), _ptr(w1), _ptr(b1),
                                       _ptr(_contig(dx_residual, torch.float32)) if dx_residual is not None else None, _ptr(dx), _ptr(dw), _ptr(db),
                                       m, h, i, _ptr(ws), need, dt, _stream()), "modcr_ffn_up_gelu_bwd")
    return dx, dw, db


def ffn_keep_supported(x, w1, b1=None):
    """True where the trainable FFN can keep its GELU input (modcr_ffn_keep_supported: bf16, M >= 256, M % 8 == 0, ...) and the
    operands sit on 16-byte boundaries (views into a packed buffer may not)"""
    if any(t is not None and (t.data_ptr() % 16 or not t.is_contiguous()) for t in (x, w1, b1)):
        return False
    return bool(lib().modcr_ffn_keep_supported(x.shape[0], x.shape[1], w1.shape[0], dt_of(x)))


def ffn_up_gelu_keep(x, w1, b1):
    """(gelu(x @ w1.T + b1), x @ w1.T + b1), both bf16 [M,I], from one GEMM (modcr_ffn_up_gelu_keep_fwd)"""
    x, w1 = _contig(x), _contig(w1)
    _same_dtype("ffn_up_gelu_keep", w1, x=x)
    m, h = x.shape
    i = w1.shape[0]
    out = torch.empty((m, i), dtype=x.dtype, device=x.device)
    pre_act = torch.empty((m, i), dtype=x.dtype, device=x.device)
    _check(lib().modcr_ffn_up_gelu_keep_fwd(_ptr(x), _ptr(w1), _ptr(_contig(b1, torch.float32)), _ptr(out), _ptr(pre_act), m, h, i,
                                            dt_of(x), _stream()), "modcr_ffn_up_gelu_keep_fwd")
    return out, pre_act


def ffn_down_residual_ln_gelu_bwd(dy, pre, inter, w2, gamma, eps, pre_act, dgamma, dbeta, dropout=None, want_db_u=False,
                                  dw_out=None, db_out=None, db_u_out=None):
    """backward of LN(dropout(inter @ w2.T + b2) + residual) that also crosses the GELU: returns (d_pre fp32 [M,H], d_u bf16 [M,I] =
    gradient of the GELU input, dw2 fp32, db2 fp32); dgamma / dbeta are accumulated (modcr_ffn_down_residual_ln_gelu_bwd)."""
    dy, pre = _contig(dy), (_contig(pre) if pre.dtype == torch.float16 else _contig(pre, torch.float32))
    inter, w2, pre_act = _contig(inter), _contig(w2), _contig(pre_act)
    _same_dtype("ffn_down_residual_ln_gelu_bwd", w2, inter=inter, pre_act=pre_act)
    m, h = pre.shape
    i = inter.shape[1]
    if pre_act.shape != inter.shape:
        raise ValueError("ffn_down_residual_ln_gelu_bwd: pre_act %s against inter %s" % (tuple(pre_act.shape), tuple(inter.shape)))
    dt = dt_of(inter)
    d_pre = torch.empty(pre.shape, dtype=torch.float32, device=pre.device)
    du = torch.empty_like(inter)
    db_u = _grad_out(db_u_out, (i,), inter.device, "ffn_down_residual_ln_gelu_bwd db_u") if want_db_u else None
    dw = _grad_out(dw_out, (h, i), inter.device, "ffn_down_residual_ln_gelu_bwd dw2")
    db = _grad_out(db_out, (h,), inter.device, "ffn_down_residual_ln_gelu_bwd db2")
    need = lib().modcr_ffn_down_gelu_bwd_workspace(m, h, i)
    ws = _workspace("lrl_bwd", need, inter.device)
    p, seed, off = dropout if dropout is not None else (0.0, 0, 0)
    _check(lib().modcr_ffn_down_residual_ln_gelu_bwd(_ptr(dy), dt_of(dy), _ptr(pre), dt_of(pre), _ptr(inter), _ptr(w2), _ptr(gamma), float(eps),
                                                     _ptr(pre_act), _ptr(d_pre), _ptr(du), _ptr(db_u), _ptr(dw), _ptr(db), _ptr(dgamma), _ptr(dbeta),
                                                     m, h, i, float(p), seed, off, _ptr(ws), need, dt, _stream()),
           "modcr_ffn_down_residual_ln_gelu_bwd")
    return (d_pre, du, dw, db, db_u) if want_db_u else (d_pre, du, dw, db)


def ffn_up_du_bwd(du, x, w1, dx_residual=None, db1=None, dw_out=None):
    """FFN-up backward from the GELU-input gradient: (dx [M,H] in x's dtype (+ dx_residual, summed in fp32), dw1 fp32, db1 fp32) (modcr_ffn_up_du_bwd).
    db1: the bias gradient when the caller already has it (db_u of ffn_down_residual_ln_gelu_bwd): the weight-gradient product then
    needs no transpose of du."""
    du, x, w1 = _contig(du), _contig(x), _contig(w1)
    _same_dtype("ffn_up_du_bwd", w1, du=du, x=x)
    m, h = x.shape
    i = w1.shape[0]
    dx = torch.empty((m, h), dtype=x.dtype, device=x.device)          # storage dtype (the fp32 sum with dx_residual is rounded once)
    dw = _grad_out(dw_out, (i, h), x.device, "ffn_up_du_bwd dw1")
    have_db = db1 is not None
    db = db1 if have_db else torch.empty((i,), dtype=torch.float32, device=x.device)
    need = lib().modcr_ffn_up_du_bwd_workspace(m, h, i)
    ws = _workspace("ffn_up_bwd", need, x.device)
    _check(lib().modcr_ffn_up_du_bwd(_ptr(du), _ptr(x), _ptr(w1),
                                     _ptr(_contig(dx_residual, torch.float32)) if dx_residual is not None else None, _ptr(dx), _ptr(dw),
                                     None if have_db else _ptr(db), m, h, i, _ptr(ws), need, dt_of(x), _stream()), "modcr_ffn_up_du_bwd")
    return dx, dw, db


class DropoutState(object):
    """(seed, running offset) of the counter-based dropout: every call consumes `numel` counters, so no two dropout
    sites / steps share mask bits; backward passes re-use the (seed, offset) pair their forward recorded."""

    def __init__(self, seed=0):
        self.seed, self.offset = int(seed) & (2 ** 64 - 1), 0

    def manual_seed(self, seed):
        self.seed, self.offset = int(seed) & (2 ** 64 - 1), 0

    def take(self, numel):
        off = (self.offset + 3) & ~3            # a multiple of 4: the row kernels then hash once per 16-byte piece (drop_apply4)
        self.offset = off + int(numel)
        return self.seed, off


DROPOUT = DropoutState(0)


def dropout(x, p, seed, offset, out=None):
    """x * mask / (1 - p) with the counter-based mask of (seed, offset); contiguous x, in place when out is x"""
    x = _contig(x)
    if out is None:
        out = torch.empty_like(x)
    _check(lib().modcr_dropout(_ptr(x), _ptr(out), x.numel(), dt_of(x), float(p), seed, offset, _stream()), "modcr_dropout")
    return out


def dropout_residual_ln(x, residual, gamma, beta, eps, p, seed, offset, out_dtype):
    """LN(dropout(x) + residual): x [M,H] fp32, or fp16 (the bf16 path's sublayer output)"""
    x = _contig(x) if x.dtype == torch.float16 else _contig(x, torch.float32)
    m, h = x.shape
    r2 = _contig(residual.reshape(m, h)) if residual is not None else None
    out = torch.empty((m, h), dtype=torch_dtype(out_dtype), device=x.device)
    _check(lib().modcr_dropout_residual_ln_fwd(_ptr(x), dt_of(x), _ptr(r2), dt_of(r2) if r2 is not None else 0, _ptr(gamma), _ptr(beta),
                                               float(eps), _ptr(out), out_dtype, None, F32, m, h, float(p), seed, offset, _stream()),
           "modcr_dropout_residual_ln_fwd")
    return out
