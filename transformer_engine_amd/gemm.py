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
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import api
from .tensor import Float8Tensor, MXFP8Tensor, TensorWrapper, current_stream_ptr, wrap_plain


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


def general_gemm(A, B, out_dtype: Optional[torch.dtype] = None, quantization_params=None,
                 gelu: bool = False, gelu_in=None, alpha: float = 1.0, beta: Optional[float] = None,
                 accumulate: bool = False, layout: str = "TN", out: Optional[torch.Tensor] = None,
                 bias: Optional[torch.Tensor] = None, use_split_accumulator: bool = False,
                 grad: bool = False, ub=None, ub_type=None, extra_output=None,
                 bulk_overlap: bool = False):
    """GEMM supporting fp8 inputs.  Returns (out, bias_grad, gelu_input, extra_output)."""
    assert layout in ("TN", "NN", "NT"), f"GEMM layout {layout} not supported."
    if gelu or gelu_in is not None:
        raise NotImplementedError("GELU epilogues are outside the Linear hot path")
    if ub is not None or ub_type is not None:
        raise NotImplementedError("userbuffers objects do not exist here: use "
                                  "transformer_engine_amd.distributed for comm+GEMM overlap")
    if quantization_params is not None:
        raise NotImplementedError("quantized GEMM output is not implemented")
    transa, transb = layout[0] == "T", layout[1] == "T"
    if beta is None:
        beta = 1.0 if accumulate else 0.0
    if not accumulate:
        beta = 0.0

    wa, a_shape = _wrap_operand(A)
    wb, b_shape = _wrap_operand(B)
    out_shape = gemm_output_shape(a_shape, b_shape, transa, transb)
    dev = A.device
    if out is None:
        if accumulate:
            raise ValueError("accumulate=True needs an `out` tensor")
        out = torch.empty(out_shape, dtype=out_dtype or torch.bfloat16, device=dev)
    else:
        if tuple(out.shape) != tuple(out_shape) and out.numel() != _numel(out_shape):
            raise ValueError(f"GEMM output has invalid dims (expected {out_shape}, got {tuple(out.shape)})")
        if not out.is_contiguous():
            raise ValueError("GEMM output must be contiguous")
    wd = wrap_plain(out)

    # zero-fill when an operand is empty (ref csrc/extensions/gemm.cpp:384-392)
    if _numel(a_shape) == 0 or _numel(b_shape) == 0:
        if out.numel() and not accumulate:
            out.zero_()
        for w in (wa, wb, wd):
            w.destroy()
        return out, None, None, extra_output

    cfg = api.nvte_create_matmul_config()
    wbias = wws = None
    try:
        if bias is not None and not grad:
            wbias = wrap_plain(bias)
            p = C.c_void_p(wbias.handle)
            api.nvte_set_matmul_config_attribute(cfg, _lib.kMBiasTensor, C.byref(p), C.sizeof(p))
        elif bias is not None and grad:
            raise NotImplementedError("bgrad GEMM epilogue: use Quantizer + bgrad_quantize instead")
        b8 = C.c_uint8(1 if use_split_accumulator else 0)
        api.nvte_set_matmul_config_attribute(cfg, _lib.kMUseSplitAccumulator, C.byref(b8), 1)
        al, be = C.c_float(alpha), C.c_float(beta)
        wws = wrap_plain(get_workspace(dev))
        api.nvte_cublas_gemm_v2(int(transa), int(transb), C.byref(al), wa.handle, wb.handle,
                                C.byref(be), wd.handle, wd.handle, wws.handle, cfg, current_stream_ptr())
    finally:
        _lib.raw_lib.nvte_destroy_matmul_config(cfg)
        for w in (wa, wb, wd, wbias, wws):
            if w is not None:
                w.destroy()
    return out, None, None, extra_output


_WORKSPACES = {}


def get_workspace(device) -> torch.Tensor:
    """Per-device GEMM workspace of the size the reference allocates (ref pytorch/cpp_extensions/
    gemm.py:35-40: 32 MiB + 1 KiB).  This library uses it for the partial sums of split tail tiles."""
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
    GEMMs fanned over the library's compute-stream pool."""
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
