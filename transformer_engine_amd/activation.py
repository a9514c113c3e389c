"""Activations fused with the FP8 cast - host mirror of ``tex.gelu`` / ``tex.dgelu`` / ``tex.swiglu`` /
``tex.dswiglu`` / ``tex.dbias_dgelu`` ... (``transformer_engine/pytorch/csrc/extensions/activation.cpp``,
``bias.cpp``): allocate the output through the quantizer, one library call.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from ._lib import api
from .tensor import (Float8CurrentScalingQuantizer, Float8Quantizer, MXFP8Quantizer, TensorWrapper,
                     current_stream_ptr, wrap_plain)

UNARY = ("gelu", "silu", "relu", "qgelu", "srelu")
GATED = ("glu", "geglu", "swiglu", "reglu", "qgeglu", "sreglu")


def _make_out(shape, like: torch.Tensor, quantizer):
    """Returns (python output object, NVTETensor wrapper, post-quantize source or None)."""
    if quantizer is None:
        out = torch.empty(shape, dtype=like.dtype, device=like.device)
        return out, wrap_plain(out), None
    if isinstance(quantizer, Float8CurrentScalingQuantizer):
        # the amax is only known afterwards: high precision first, then amax -> scale -> cast
        hp = torch.empty(shape, dtype=like.dtype, device=like.device)
        return None, wrap_plain(hp), hp
    if isinstance(quantizer, Float8Quantizer):
        out = quantizer.make_empty(shape, dtype=like.dtype, device=like.device)
        w = quantizer._wrap_out(out)
        w.set_scale(quantizer.scale).set_amax(quantizer.amax)
        return out, w, None
    if isinstance(quantizer, MXFP8Quantizer):
        # block scales are local to the tile: the activation and the MXFP8 quantize are one kernel
        out = quantizer.make_empty(shape, dtype=like.dtype, device=like.device)
        return out, out.wrap(), None
    raise NotImplementedError(f"fused activation output for {type(quantizer).__name__} is not implemented "
                              "(high precision, FP8 per-tensor scaling or MXFP8)")


def _finish(out, hp, quantizer):
    if hp is not None:
        return quantizer(hp)
    if isinstance(quantizer, Float8Quantizer):
        out._transpose_invalid = out._transpose is None
    return out


def activation(name: str, x: torch.Tensor, quantizer=None):
    """tex.<name>(input, quantizer): plain (gelu, silu, relu, qgelu, srelu) or gated (glu, geglu, swiglu,
    reglu, qgeglu, sreglu: input [..., 2H] -> [..., H])."""
    if name not in UNARY + GATED:
        raise ValueError(f"unknown activation {name}")
    x2 = x.reshape(-1, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    H = x2.shape[1] // 2 if name in GATED else x2.shape[1]
    out, wo, hp = _make_out((x2.shape[0], H), x2, quantizer)
    wi = wrap_plain(x2)
    try:
        getattr(api, "nvte_" + name)(wi.handle, wo.handle, current_stream_ptr())
    finally:
        wi.destroy(); wo.destroy()
    return _finish(out, hp, quantizer)


def dactivation(name: str, grad: torch.Tensor, x: torch.Tensor, quantizer=None):
    """tex.d<name>(grad, fwd_input, quantizer); gated: grad [..., H], input [..., 2H] -> [..., 2H]."""
    if name not in UNARY + GATED:
        raise ValueError(f"unknown activation {name}")
    x2 = x.reshape(-1, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    g2 = grad.reshape(-1, grad.shape[-1])
    g2 = g2 if g2.is_contiguous() else g2.contiguous()
    out, wo, hp = _make_out(tuple(x2.shape), x2, quantizer)
    wg, wi = wrap_plain(g2), wrap_plain(x2)
    try:
        getattr(api, "nvte_d" + name)(wg.handle, wi.handle, wo.handle, current_stream_ptr())
    finally:
        wg.destroy(); wi.destroy(); wo.destroy()
    return _finish(out, hp, quantizer)


def clamped_swiglu(x: torch.Tensor, quantizer=None, *, limit: float = 7.0, alpha: float = 1.702,
                   glu_linear_offset: float = 1.0):
    """tex.clamped_swiglu (GPT-OSS): act(a) * (clamp(g) + offset), input [..., 2H] -> [..., H]."""
    x2 = x.reshape(-1, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    out, wo, hp = _make_out((x2.shape[0], x2.shape[1] // 2), x2, quantizer)
    wi = wrap_plain(x2)
    try:
        api.nvte_clamped_swiglu_v2(wi.handle, wo.handle, C.c_float(limit), C.c_float(alpha),
                                   C.c_float(glu_linear_offset), current_stream_ptr())
    finally:
        wi.destroy(); wo.destroy()
    return _finish(out, hp, quantizer)


def clamped_dswiglu(grad: torch.Tensor, x: torch.Tensor, quantizer=None, *, limit: float = 7.0,
                    alpha: float = 1.702, glu_linear_offset: float = 1.0):
    """tex.clamped_dswiglu: grad [..., H], forward input [..., 2H] -> [..., 2H]."""
    x2 = x.reshape(-1, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    g2 = grad.reshape(-1, grad.shape[-1])
    g2 = g2 if g2.is_contiguous() else g2.contiguous()
    out, wo, hp = _make_out(tuple(x2.shape), x2, quantizer)
    wg, wi = wrap_plain(g2), wrap_plain(x2)
    try:
        api.nvte_clamped_dswiglu_v2(wg.handle, wi.handle, wo.handle, C.c_float(limit), C.c_float(alpha),
                                    C.c_float(glu_linear_offset), current_stream_ptr())
    finally:
        wg.destroy(); wi.destroy(); wo.destroy()
    return _finish(out, hp, quantizer)


SCALED = ("swiglu", "clamped_swiglu", "srelu")


def _flat2d(t: torch.Tensor) -> torch.Tensor:
    t2 = t.reshape(-1, t.shape[-1])
    return t2 if t2.is_contiguous() else t2.contiguous()


def scaled_activation(name: str, x: torch.Tensor, act_scales: torch.Tensor, *, glu_interleave_size: int = 0,
                      limit: float = 7.0, alpha: float = 1.702, glu_linear_offset: float = 1.0,
                      out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``Act(x) * act_scales[:, None]`` in one kernel (``nvte_scaled_swiglu`` / ``nvte_scaled_clamped_swiglu`` /
    ``nvte_scaled_srelu``, ref activation.h:372-406,590-600 - what ``ops/basic/swiglu.py:_ScaledGLU`` and
    ``ops/basic/activation.py:ScaledSReLU`` compute): the MoE expert activation times the router
    probability of its token.  x [..., 2H] (gated; optionally in the block-interleaved layout) or [..., H]."""
    assert name in SCALED, name
    x2 = _flat2d(x)
    gated = name != "srelu"
    h = x2.shape[1] // 2 if gated else x2.shape[1]
    out = torch.empty((x2.shape[0], h), dtype=out_dtype or x.dtype, device=x.device)
    sc = act_scales.reshape(-1).contiguous()
    wi, ws, wo = wrap_plain(x2), wrap_plain(sc), wrap_plain(out)
    st = current_stream_ptr()
    try:
        if name == "swiglu":
            api.nvte_scaled_swiglu(wi.handle, ws.handle, wo.handle, C.c_int64(glu_interleave_size), st)
        elif name == "clamped_swiglu":
            api.nvte_scaled_clamped_swiglu(wi.handle, ws.handle, wo.handle, C.c_float(limit), C.c_float(alpha),
                                           C.c_float(glu_linear_offset), C.c_int64(glu_interleave_size), st)
        else:
            api.nvte_scaled_srelu(wi.handle, ws.handle, wo.handle, st)
    finally:
        wi.destroy(); ws.destroy(); wo.destroy()
    return out.reshape(*x.shape[:-1], h)


def scaled_dactivation(name: str, grad: torch.Tensor, x: torch.Tensor, act_scales: torch.Tensor, *,
                       glu_interleave_size: int = 0, limit: float = 7.0, alpha: float = 1.702,
                       glu_linear_offset: float = 1.0, with_scale_grad: bool = True,
                       scale_grad_dtype: Optional[torch.dtype] = None):
    """Backward of :func:`scaled_activation` (``nvte_scaled_dswiglu`` / ``nvte_scaled_clamped_dswiglu`` /
    ``nvte_scaled_dsrelu``, ref activation.h:513-552,602-615).  Returns (grad_input with x's shape,
    grad_act_scales [rows] or None)."""
    assert name in SCALED, name
    x2, g2 = _flat2d(x), _flat2d(grad)
    dx = torch.empty_like(x2)
    sc = act_scales.reshape(-1).contiguous()
    dsc = torch.empty(x2.shape[0], dtype=scale_grad_dtype or x.dtype, device=x.device) if with_scale_grad else None
    wg, wi, ws, wo = wrap_plain(g2), wrap_plain(x2), wrap_plain(sc), wrap_plain(dx)
    wd = wrap_plain(dsc) if dsc is not None else None
    hd = wd.handle if wd is not None else None
    st = current_stream_ptr()
    try:
        if name == "swiglu":
            api.nvte_scaled_dswiglu(wg.handle, wi.handle, ws.handle, wo.handle, hd, C.c_int64(glu_interleave_size), st)
        elif name == "clamped_swiglu":
            api.nvte_scaled_clamped_dswiglu(wg.handle, wi.handle, ws.handle, wo.handle, hd, C.c_float(limit),
                                            C.c_float(alpha), C.c_float(glu_linear_offset),
                                            C.c_int64(glu_interleave_size), st)
        else:
            api.nvte_scaled_dsrelu(wg.handle, wi.handle, ws.handle, wo.handle, hd, st)
    finally:
        wg.destroy(); wi.destroy(); ws.destroy(); wo.destroy()
        if wd is not None:
            wd.destroy()
    return dx.reshape(x.shape), dsc


def dbias_dact(name: str, grad: torch.Tensor, act_input: torch.Tensor, quantizer: Float8Quantizer):
    """tex.dbias_d<name>(grad, act_input, quantizer) -> (dbias, quantized grad * act'(act_input))."""
    if name not in UNARY:
        raise ValueError(f"unknown activation {name}")
    x2 = act_input.reshape(-1, act_input.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    g2 = grad.reshape(-1, grad.shape[-1])
    g2 = g2 if g2.is_contiguous() else g2.contiguous()
    out, wo, hp = _make_out(tuple(x2.shape), x2, quantizer)
    if hp is not None:
        raise NotImplementedError("dbias_dact with current scaling: run dactivation, then bgrad separately")
    dbias = torch.empty(x2.shape[1], dtype=x2.dtype, device=x2.device)
    wg, wi, wd, ws = wrap_plain(g2), wrap_plain(x2), wrap_plain(dbias), TensorWrapper()
    fn = getattr(api, "nvte_quantize_dbias_d" + name)
    stream = current_stream_ptr()
    try:
        fn(wg.handle, wi.handle, wo.handle, wd.handle, ws.handle, stream)       # workspace query
        bt = ws.get_param(0)
        workspace = torch.empty([bt.shape.data[i] for i in range(bt.shape.ndim)], dtype=torch.float32, device=x2.device)
        ws.set_rowwise_data(workspace)
        fn(wg.handle, wi.handle, wo.handle, wd.handle, ws.handle, stream)
    finally:
        for w in (wg, wi, wd, ws, wo):
            w.destroy()
    return dbias, _finish(out, None, quantizer)
