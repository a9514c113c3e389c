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
