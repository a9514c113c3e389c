"""``general_gemm`` - host-side mirror of the reference's
``transformer_engine/pytorch/cpp_extensions/gemm.py:106-274`` + ``pytorch/csrc/extensions/
gemm.cpp:144-414``: layout string -> transa/transb, output shape, alpha/beta/accumulate,
bias.  Math runs in ``nvte_cublas_gemm_v2`` (hand-written MFMA kernel).

Layout semantics for a Linear with tokens M, in K, out N:
    fprop  general_gemm(W [N,K], X  [M,K], layout="TN") -> Y  [M,N] = X  W^T
    dgrad  general_gemm(W [N,K], dY [M,N], layout="NN") -> dX [M,K] = dY W
    wgrad  general_gemm(X [M,K], dY [M,N], layout="NT") -> dW [N,K] = dY^T X
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import api
from .tensor import (Float8Quantizer, Float8Tensor, MXFP8Tensor, TensorWrapper, current_stream_ptr,
                     wrap_plain)


def _wrap_operand(t) -> Tuple[TensorWrapper, Tuple[int, ...]]:
    if isinstance(t, (Float8Tensor, MXFP8Tensor)):
        return t.wrap(), tuple(t.shape)
    if isinstance(t, torch.Tensor):
        return wrap_plain(t), tuple(t.shape)
    raise TypeError(f"unsupported GEMM operand {type(t)}")


def _flat2d(shape):
    rows = 1
    for d in shape[:-1]:
        rows *= d
    return rows, shape[-1]


def gemm_output_shape(a_shape, b_shape, transa: bool, transb: bool) -> Tuple[int, ...]:
    """ref pytorch/csrc/extensions/gemm.cpp:41-82"""
    a0, a1 = _flat2d(a_shape)
    m = a0 if transa else a1
    if transb:
        return (b_shape[-1], m)
    return tuple(b_shape[:-1]) + (m,)


_CUS = {}


def _num_cus(dev) -> int:
    idx = torch.device(dev).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _CUS:
        _CUS[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    return _CUS[idx]


def general_gemm(A, B, out_dtype: Optional[torch.dtype] = None, quantization_params=None,
                 gelu: bool = False, gelu_in=None, alpha: float = 1.0, beta: Optional[float] = None,
                 accumulate: bool = False, layout: str = "TN", out: Optional[torch.Tensor] = None,
                 bias: Optional[torch.Tensor] = None, use_split_accumulator: bool = False,
                 grad: bool = False, ub=None, ub_type=None, extra_output=None,
                 bulk_overlap: bool = False):
    """GEMM supporting fp8 inputs.  Returns (out, bias_grad, gelu_input, extra_output)."""
    assert layout in ("TN", "NN", "NT"), f"GEMM layout {layout} not supported."
    if ub is not None or ub_type is not None:
        raise NotImplementedError("userbuffers objects do not exist here: use "
                                  "transformer_engine_amd.distributed for comm+GEMM overlap")
    transa, transb = layout[0] == "T", layout[1] == "T"
    if beta is None:
        beta = 1.0 if accumulate else 0.0
    if not accumulate and beta != 0.0:
        # ref csrc/extensions/gemm.cpp:195-196
        raise ValueError("Trying to use non-zero beta while not accumulating into D tensor. "
                         "Beta has nothing to be applied to.")

    A, B = _swizzle_compact_mx_scales(A, B, transa, transb)
    wa, a_shape = _wrap_operand(A)
    wb, b_shape = _wrap_operand(B)
    out_shape = gemm_output_shape(a_shape, b_shape, transa, transb)
    dev = A.device

    # Output quantization (ref csrc/extensions/gemm.cpp:215-235): fused into the GEMM epilogue only
    # for a delayed-scaling quantizer over per-tensor-scaled inputs; otherwise GEMM to high
    # precision, then quantize.
    fused_q = (isinstance(quantization_params, Float8Quantizer)
               and (isinstance(A, Float8Tensor) or isinstance(B, Float8Tensor)))
    unfused_q = quantization_params is not None and not fused_q
    q_out = None
    if fused_q:
        if accumulate:
            raise ValueError("Accumulation mode not supported for FP8 output")
        q_out = out if isinstance(out, Float8Tensor) else quantization_params.make_empty(
            out_shape, dtype=out_dtype or torch.bfloat16, device=dev)
        if q_out._data is None:
            raise ValueError("quantized GEMM output needs row-wise usage")
        wd = TensorWrapper(_lib.NVTE_DELAYED_TENSOR_SCALING)
        wd.set_rowwise_data(q_out._data, quantization_params.dtype, out_shape)
        wd.set_rowwise_scale_inv(q_out._scale_inv)
        wd.set_scale(quantization_params.scale).set_amax(quantization_params.amax)
        hp_out = None
    else:
        if out is None or unfused_q:
            if accumulate:
                raise ValueError("accumulate=True needs an `out` tensor")
            hp_out = torch.empty(out_shape, dtype=out_dtype or torch.bfloat16, device=dev)
        else:
            hp_out = out
            if tuple(out.shape) != tuple(out_shape) and out.numel() != _numel(out_shape):
                raise ValueError(f"GEMM output has invalid dims (expected {out_shape}, got {tuple(out.shape)})")
            if not out.is_contiguous():
                raise ValueError("Output tensor is not contiguous.")
        wd = wrap_plain(hp_out)

    def _finish(result):
        if fused_q:
            q_out._transpose_invalid = True
            if q_out._transpose is not None:
                q_out._create_transpose()
            return q_out
        if unfused_q:
            return quantization_params.quantize(result, out=out if not isinstance(out, torch.Tensor) else None)
        return result

    # zero-fill when an operand is empty (ref csrc/extensions/gemm.cpp:384-392)
    if _numel(a_shape) == 0 or _numel(b_shape) == 0:
        if hp_out is not None and hp_out.numel() and not accumulate:
            hp_out.zero_()
        for w in (wa, wb, wd):
            w.destroy()
        return (hp_out if hp_out is not None else q_out), None, None, extra_output

    a0, a1 = _flat2d(a_shape)
    b0, b1 = _flat2d(b_shape)
    m = a0 if transa else a1
    n = b1 if transb else b0
    hp_dtype = hp_out.dtype if hp_out is not None else (out_dtype or torch.bfloat16)
    bias_grad = gelu_input = None
    cfg = api.nvte_create_matmul_config()
    wbias = wws = waux = None
    try:
        if bias is not None and not grad:
            wbias = wrap_plain(bias.contiguous())
            p = C.c_void_p(wbias.handle)
            api.nvte_set_matmul_config_attribute(cfg, _lib.kMBiasTensor, C.byref(p), C.sizeof(p))
        elif bias is not None and grad:
            # ref csrc/extensions/gemm.cpp:241-245 allocates B.shape[-1] entries of the output dtype.
            # The entries the epilogue writes: BGRADB -> one per row of op(B) (n); DGELU_BGRAD -> one
            # per output column (m).
            bias_grad = torch.empty(m if gelu else n, dtype=hp_dtype, device=dev)
            wbias = wrap_plain(bias_grad)
            p = C.c_void_p(wbias.handle)
            api.nvte_set_matmul_config_attribute(cfg, _lib.kMDBiasTensor, C.byref(p), C.sizeof(p))
        if gelu:
            # aux dtype: the bias dtype for FP8 inputs (bf16 without a bias), ref gemm.cpp:256
            gelu_type = bias.dtype if bias is not None else torch.bfloat16
            if not grad:
                gelu_input = torch.empty(out_shape, dtype=gelu_type, device=dev)
            else:
                if gelu_in is None:
                    raise ValueError("Requested epilogue without providing aux tensor.")
                gelu_input = gelu_in.contiguous()
            waux = wrap_plain(gelu_input)
            p = C.c_void_p(waux.handle)
            api.nvte_set_matmul_config_attribute(cfg, _lib.kMEpilogueAux, C.byref(p), C.sizeof(p))
            one = C.c_uint8(1)
            api.nvte_set_matmul_config_attribute(cfg, _lib.kMWithDGELU if grad else _lib.kMWithGELU,
                                                 C.byref(one), 1)
        b8 = C.c_uint8(1 if use_split_accumulator else 0)
        api.nvte_set_matmul_config_attribute(cfg, _lib.kMUseSplitAccumulator, C.byref(b8), 1)
        # external margin for every GEMM (ref csrc/extensions/gemm.cpp:281-298): NVTE_EXT_MARGIN_SM compute units
        # stay free for the communication kernels of an overlapped step; 0 = the whole chip
        margin = os.environ.get("NVTE_EXT_MARGIN_SM")
        if margin:
            cus = _num_cus(dev)
            smc = C.c_int(max(cus - int(margin), 0))
            api.nvte_set_matmul_config_attribute(cfg, _lib.kMSMCount, C.byref(smc), C.sizeof(smc))
        al, be = C.c_float(alpha), C.c_float(beta)
        wws = wrap_plain(get_workspace(dev))
        api.nvte_cublas_gemm_v2(int(transa), int(transb), C.byref(al), wa.handle, wb.handle,
                                C.byref(be), wd.handle, wd.handle, wws.handle, cfg, current_stream_ptr())
    finally:
        _lib.raw_lib.nvte_destroy_matmul_config(cfg)
        for w in (wa, wb, wd, wbias, wws, waux):
            if w is not None:
                w.destroy()
    return _finish(hp_out), bias_grad, gelu_input, extra_output


# MXFP8 operands with COMPACT scales (a gathered operand, the output of a fused norm): the reference's GEMM wrapper
# swizzles them first (ref pytorch/csrc/extensions/gemm.cpp swizzle_scales_for_gemm); the kernel here also reads the
# compact layout directly, 8 % slower (one 128-byte line per tile row and K tile instead of 512 contiguous bytes per 128
# rows).  From this many multiply-adds on, a swizzle pass over the 1/32-size scale matrix (about 5 us) is the cheaper way.
_SWIZZLE_FIRST_MACS = 1 << 37


def _swizzled_scales_copy(t: MXFP8Tensor, rowwise: bool) -> MXFP8Tensor:
    data = t._rowwise_data if rowwise else t._columnwise_data
    sc = t._rowwise_scale_inv if rowwise else t._columnwise_scale_inv
    if data is None or sc is None:
        return t            # the library reports the missing usage itself
    def view(scales, swizzled):
        return MXFP8Tensor(shape=t._shape, dtype=t.dtype, rowwise_data=data if rowwise else None,
                           rowwise_scale_inv=scales if rowwise else None,
                           columnwise_data=None if rowwise else data,
                           columnwise_scale_inv=None if rowwise else scales, fp8_dtype=t._fp8_dtype,
                           quantizer=t._quantizer, with_gemm_swizzled_scales=swizzled,
                           columnwise_transposed=(not rowwise) and t._columnwise_transposed)
    src, dst = view(sc, False), view(torch.empty_like(sc), True)
    wi, wo = src.wrap(), dst.wrap()
    try:
        api.nvte_swizzle_scaling_factors(wi.handle, wo.handle, current_stream_ptr())
    finally:
        wi.destroy(); wo.destroy()
    return dst


def _swizzle_compact_mx_scales(A, B, transa: bool, transb: bool):
    if not (isinstance(A, MXFP8Tensor) and isinstance(B, MXFP8Tensor)):
        return A, B
    if A._with_gemm_swizzled_scales and B._with_gemm_swizzled_scales:
        return A, B
    if A.numel() == 0 or B.numel() == 0:
        return A, B
    a0, a1 = _flat2d(A.shape)
    b0, b1 = _flat2d(B.shape)
    m, k = (a0, a1) if transa else (a1, a0)
    n = b1 if transb else b0
    if m * n * k < _SWIZZLE_FIRST_MACS:
        return A, B
    if not A._with_gemm_swizzled_scales:
        A = _swizzled_scales_copy(A, rowwise=transa)
    if not B._with_gemm_swizzled_scales:
        B = _swizzled_scales_copy(B, rowwise=not transb)
    return A, B


def multi_gemm(entries, grad: bool = False):
    """A list of independent FP8 GEMMs in ONE persistent launch (``nvte_amd_multi_gemm``, amd_ext.h): each entry
    is ``(A, B, layout, out_or_dtype, accumulate)`` with the meaning of the same :func:`general_gemm` call
    (no bias, no GELU, high-precision output).  The library fuses the list when every operand is per-tensor
    scaled FP8 of matching formats and the outputs share a dtype (the dgrad + wgrad pair of a layer), and runs
    it entry by entry otherwise - the results are those of the separate calls either way."""
    if not entries:
        return []
    dev = entries[0][0].device
    n = len(entries)
    was, wbs, wds, outs = [], [], [], []
    ta, tb, acc = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
    wws = None
    try:
        for i, (A, B, layout, out, accumulate) in enumerate(entries):
            assert layout in ("TN", "NN", "NT"), f"GEMM layout {layout} not supported."
            transa, transb = layout[0] == "T", layout[1] == "T"
            wa, a_shape = _wrap_operand(A)
            wb, b_shape = _wrap_operand(B)
            was.append(wa); wbs.append(wb)
            if _numel(a_shape) == 0 or _numel(b_shape) == 0:
                raise ValueError("multi_gemm does not take empty operands: use general_gemm")
            out_shape = gemm_output_shape(a_shape, b_shape, transa, transb)
            if isinstance(out, torch.Tensor):
                if out.numel() != _numel(out_shape) or not out.is_contiguous():
                    raise ValueError(f"GEMM output has invalid dims (expected {out_shape}, got {tuple(out.shape)})")
                o = out
            else:
                if accumulate:
                    raise ValueError("accumulate=True needs an `out` tensor")
                o = torch.empty(out_shape, dtype=out or torch.bfloat16, device=dev)
            outs.append(o)
            wds.append(wrap_plain(o))
            ta[i], tb[i], acc[i] = int(transa), int(transb), int(bool(accumulate))
        ha = (C.c_void_p * n)(*[w.handle for w in was])
        hb = (C.c_void_p * n)(*[w.handle for w in wbs])
        hd = (C.c_void_p * n)(*[w.handle for w in wds])
        wws = wrap_plain(get_workspace(dev))
        api.nvte_amd_multi_gemm(ha, hb, hd, ta, tb, n, int(grad), wws.handle, acc, current_stream_ptr())
    finally:
        for w in was + wbs + wds + ([wws] if wws is not None else []):
            w.destroy()
    return outs


_WORKSPACES = {}


def get_workspace(device) -> torch.Tensor:
    """Per-device GEMM workspace of the size the reference allocates (ref pytorch/cpp_extensions/
    gemm.py:35-40: 32 MiB + 1 KiB).  This library only uses it for the FP32 column sums of the
    DGELU_BGRAD epilogue."""
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = torch.empty(32 * 1024 * 1024 + 1024, dtype=torch.uint8, device=device)
        _WORKSPACES[key] = ws
    return ws


def _numel(shape) -> int:
    n = 1
    for d in shape:
        n *= d
    return n


def general_grouped_gemm(A, B, out, out_dtype, layout: str = "TN", m_splits=None, bias=None,
                         accumulate: bool = False, grad: bool = False,
                         use_split_accumulator: bool = False, single_output: bool = False):
    """List form (ref pytorch/cpp_extensions/gemm.py:277 -> nvte_multi_tensor_gemm): independent
    GEMMs, each a persistent full-chip launch, in order on the current stream."""
    assert layout in ("TN", "NN", "NT")
    transa, transb = layout[0] == "T", layout[1] == "T"
    n = len(A)
    if single_output:
        # one contiguous output split along dim 0 according to m_splits
        assert m_splits is not None
        outs = list(torch.split(out[0], list(m_splits), dim=0))
    else:
        outs = out
    was, wbs, wds, wbias = [], [], [], []
    for i in range(n):
        wa, _ = _wrap_operand(A[i]); wb, _ = _wrap_operand(B[i])
        was.append(wa); wbs.append(wb); wds.append(wrap_plain(outs[i]))
        wbias.append(wrap_plain(bias[i]) if (bias is not None and bias[i] is not None and bias[i].numel() and not grad) else TensorWrapper())
    arr = lambda ws: (C.c_void_p * n)(*[w.handle for w in ws])
    empty = [TensorWrapper() for _ in range(n)]
    try:
        api.nvte_multi_tensor_gemm(arr(was), arr(wbs), arr(wds), arr(wbias), arr(empty), n, transa,
                                   transb, grad, arr(empty), accumulate, use_split_accumulator, 0,
                                   current_stream_ptr())
    finally:
        for w in was + wbs + wds + wbias + empty:
            w.destroy()
    return out, None, None
