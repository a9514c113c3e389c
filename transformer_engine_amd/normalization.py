"""LayerNorm / RMSNorm with fused quantization - host mirror of ``tex.layernorm_fwd`` /
``tex.rmsnorm_fwd`` / ``tex.layernorm_bwd`` / ``tex.rmsnorm_bwd``
(``transformer_engine/pytorch/csrc/extensions/normalization.cpp:60-330``): allocate outputs through
the quantizer, run the two-call workspace protocol, return ``(out, mu, rsigma)``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from ._lib import api
from .tensor import (Float8CurrentScalingQuantizer, Float8Quantizer, MXFP8Quantizer, TensorWrapper,
                     current_stream_ptr, wrap_plain)


def _run_with_workspace(call, device) -> None:
    """call(workspace_handle): once with an empty workspace (shape query), once for real."""
    ws = TensorWrapper()
    call(ws.handle)
    bt = ws.get_param(0)
    nbytes = 1
    for i in range(bt.shape.ndim):
        nbytes *= bt.shape.data[i]
    buf = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
    ws.set_rowwise_data(buf)
    call(ws.handle)
    ws.destroy()


def _norm_fwd(kind: str, x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float,
              quantizer, out_dtype: Optional[torch.dtype], zero_centered_gamma: bool):
    if x.dim() != 2:
        raise ValueError("x must be 2D tensor.")
    x = x if x.is_contiguous() else x.contiguous()
    N, H = x.shape
    dev = x.device
    rsigma = torch.empty(N, dtype=torch.float32, device=dev)
    mu = torch.empty(N, dtype=torch.float32, device=dev) if kind == "ln" else None
    wx, wg = wrap_plain(x), wrap_plain(weight)
    wb = wrap_plain(bias) if kind == "ln" else None
    wrs = wrap_plain(rsigma)
    wmu = wrap_plain(mu) if kind == "ln" else None
    stream = current_stream_ptr()
    post_quantize = None
    if quantizer is None:
        out = torch.empty((N, H), dtype=out_dtype or x.dtype, device=dev)
        wz = wrap_plain(out)
    elif isinstance(quantizer, Float8CurrentScalingQuantizer):
        # amax is only known after the norm: high-precision output, then quantize
        # (ref normalization.cpp:118-135, "unfused" branch for current scaling)
        hp = torch.empty((N, H), dtype=x.dtype, device=dev)
        wz = wrap_plain(hp)
        post_quantize = hp
        out = None
    elif isinstance(quantizer, Float8Quantizer):
        out = quantizer.make_empty((N, H), dtype=x.dtype, device=dev)
        wz = quantizer._wrap_out(out)
        wz.set_scale(quantizer.scale).set_amax(quantizer.amax)
    elif isinstance(quantizer, MXFP8Quantizer):
        out = quantizer.make_empty((N, H), dtype=x.dtype, device=dev)
        wz = out.wrap()
    else:
        raise TypeError(f"unsupported quantizer {type(quantizer)}")

    def call(ws_handle):
        if kind == "ln":
            api.nvte_layernorm_fwd(wx.handle, wg.handle, wb.handle, float(eps), wz.handle, wmu.handle, wrs.handle,
                                   ws_handle, 0, bool(zero_centered_gamma), stream)
        else:
            api.nvte_rmsnorm_fwd(wx.handle, wg.handle, float(eps), wz.handle, wrs.handle, ws_handle, 0,
                                 bool(zero_centered_gamma), stream)

    try:
        _run_with_workspace(call, dev)
    finally:
        for w in (wx, wg, wb, wrs, wmu, wz):
            if w is not None:
                w.destroy()
    if post_quantize is not None:
        out = quantizer(post_quantize)
    elif isinstance(quantizer, Float8Quantizer):
        out._transpose_invalid = out._transpose is None
    return out, mu, rsigma


def layernorm_fwd(x, weight, bias, eps: float, quantizer=None, out_dtype=None, sm_margin: int = 0,
                  zero_centered_gamma: bool = False):
    """ref tex.layernorm_fwd -> (out, mu, rsigma)"""
    return _norm_fwd("ln", x, weight, bias, eps, quantizer, out_dtype, zero_centered_gamma)


def rmsnorm_fwd(x, weight, eps: float, quantizer=None, out_dtype=None, sm_margin: int = 0,
                zero_centered_gamma: bool = False):
    """ref tex.rmsnorm_fwd -> (out, None, rsigma)"""
    return _norm_fwd("rms", x, weight, None, eps, quantizer, out_dtype, zero_centered_gamma)


def layernorm_bwd(dz, x, mu, rsigma, gamma, sm_margin: int = 0, zero_centered_gamma: bool = False):
    """ref tex.layernorm_bwd -> (dx, dgamma, dbeta)"""
    dz = dz if dz.is_contiguous() else dz.contiguous()
    dx = torch.empty_like(x)
    dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
    ws = [wrap_plain(t) for t in (dz, x, mu, rsigma, gamma, dx, dgamma, dbeta)]
    stream = current_stream_ptr()
    try:
        _run_with_workspace(lambda h: api.nvte_layernorm_bwd(*[w.handle for w in ws], h, 0,
                                                             bool(zero_centered_gamma), stream), x.device)
    finally:
        for w in ws:
            w.destroy()
    return dx, dgamma, dbeta


def rmsnorm_bwd(dz, x, rsigma, gamma, sm_margin: int = 0, zero_centered_gamma: bool = False, add=None):
    """ref tex.rmsnorm_bwd / tex.rmsnorm_bwd_add -> (dx, dgamma)"""
    dz = dz if dz.is_contiguous() else dz.contiguous()
    dx = torch.empty_like(x)
    dgamma = torch.empty_like(gamma)
    stream = current_stream_ptr()
    if add is None:
        ws = [wrap_plain(t) for t in (dz, x, rsigma, gamma, dx, dgamma)]
        fn = api.nvte_rmsnorm_bwd
    else:
        ws = [wrap_plain(t) for t in (dz, x, add, rsigma, gamma, dx, dgamma)]
        fn = api.nvte_rmsnorm_bwd_add
    try:
        _run_with_workspace(lambda h: fn(*[w.handle for w in ws], h, 0, bool(zero_centered_gamma), stream), x.device)
    finally:
        for w in ws:
            w.destroy()
    return dx, dgamma
