"""Grouped (multi-expert) tensors, grouped quantize and grouped GEMM with DEVICE-side member shapes.

Host-side mirror of ``transformer_engine/pytorch/tensor/storage/grouped_tensor_storage.py``
(GroupedTensorStorage), ``tex.group_quantize`` (``pytorch/csrc/extensions/cast.cpp:220-400``) and
``general_grouped_gemm_for_grouped_tensor`` (``pytorch/cpp_extensions/gemm.py:422-620``).  Member
shapes (``first_dims`` / ``tensor_offsets``) stay on the device from the router to the GEMM: nothing
in here reads them on the host, which is what makes the path HIP-graph capturable and removes the
``m_splits.tolist()`` sync of the legacy GroupedLinear path (``module/grouped_linear.py:630``).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Union

import torch

from . import _lib
from ._lib import api
from .tensor import (DType, Float8CurrentScalingQuantizer, Float8Quantizer, Float8Tensor, MXFP8Quantizer,
                     QuantizationConfig, TensorWrapper, current_stream_ptr, te_dtype, wrap_plain)

(kGRowwiseData, kGColumnwiseData, kGScale, kGAmax, kGRowwiseScaleInv, kGColumnwiseScaleInv, kGColumnwiseAmax,
 kGFirstDims, kGLastDims, kGTensorOffsets, kGWithGEMMSwizzledScales) = range(11)


class GroupedTensorWrapper:
    """RAII NVTEGroupedTensor handle (ref transformer_engine.h:1180-1320)."""

    def __init__(self, num_tensors: int, logical_shape: Sequence[int],
                 scaling_mode: int = _lib.NVTE_DELAYED_TENSOR_SCALING):
        self.handle = api.nvte_create_grouped_tensor(int(scaling_mode), int(num_tensors),
                                                     _lib.make_shape(list(logical_shape)))
        self._keep = []

    def set(self, param: int, t: Optional[torch.Tensor], dtype: Optional[DType] = None):
        if t is None:
            return self
        if not t.is_contiguous():
            raise ValueError("grouped tensor buffers must be contiguous")
        self._keep.append(t)
        bt = _lib.NVTEBasicTensor(C.c_void_p(t.data_ptr()), int(dtype if dtype is not None else te_dtype(t.dtype)),
                                  _lib.make_shape([t.numel()]))
        api.nvte_set_grouped_tensor_param(self.handle, param, C.byref(bt), C.sizeof(bt))
        return self

    def destroy(self):
        if self.handle:
            _lib.raw_lib.nvte_destroy_grouped_tensor(self.handle)
            self.handle = None
            self._keep = []

    def __del__(self):  # pragma: no cover
        try:
            self.destroy()
        except Exception:
            pass


class GroupedTensor:
    """num_tensors matrices stored back to back in flat buffers (ref grouped_tensor_storage.py).

    ``first_dims`` / ``last_dims`` are int64 DEVICE tensors (None = uniform), ``tensor_offsets`` the
    int64 device prefix sums with num_tensors + 1 entries (required when a dim varies).  For FP8
    per-tensor scaling ``columnwise_data`` holds each member's TRANSPOSE at the member's own offset
    and ``scale`` / ``scale_inv`` / ``amax`` have one entry per member.  For MXFP8
    (``scaling_mode=NVTE_MXFP8_1D_SCALING``) the scale buffers are e8m0 bytes: rowwise
    ``[pad128(sum first), pad4(last/32)]``; columnwise either the reference layout
    ``[pad4(sum first / 32), pad128(last)]`` next to an un-transposed payload, or - with
    ``columnwise_transposed`` (amd_ext.h) - per-member transposed payloads ``[last, first_i]`` with
    ``[pad128(last), first_i/32]`` scale matrices back to back.  Member first dims are multiples of 128."""

    def __init__(self, *, num_tensors: int, logical_shape, dtype: torch.dtype, rowwise_data=None,
                 columnwise_data=None, scale_inv=None, columnwise_scale_inv=None, scale=None, amax=None,
                 first_dims=None, last_dims=None, tensor_offsets=None, fp8_dtype: Optional[DType] = None,
                 scaling_mode: int = _lib.NVTE_DELAYED_TENSOR_SCALING, columnwise_transposed: bool = False,
                 with_gemm_swizzled_scales: bool = False):
        self.scaling_mode = int(scaling_mode)
        self.columnwise_transposed = bool(columnwise_transposed)
        self.with_gemm_swizzled_scales = bool(with_gemm_swizzled_scales)
        self.num_tensors = int(num_tensors)
        self.logical_shape = tuple(int(d) for d in logical_shape)
        self.dtype = dtype
        self.rowwise_data, self.columnwise_data = rowwise_data, columnwise_data
        self.scale_inv, self.columnwise_scale_inv = scale_inv, columnwise_scale_inv
        self.scale, self.amax = scale, amax
        self.first_dims, self.last_dims, self.tensor_offsets = first_dims, last_dims, tensor_offsets
        self.fp8_dtype = DType.cast(fp8_dtype) if fp8_dtype is not None else None
        for name, arr in (("first_dims", first_dims), ("last_dims", last_dims), ("tensor_offsets", tensor_offsets)):
            if arr is not None and arr.dtype != torch.int64:
                raise TypeError(f"{name} must be an int64 device tensor")

    @property
    def device(self):
        return (self.rowwise_data if self.rowwise_data is not None else self.columnwise_data).device

    def wrap(self) -> GroupedTensorWrapper:
        w = GroupedTensorWrapper(self.num_tensors, self.logical_shape, self.scaling_mode)
        pd = self.fp8_dtype  # None for high-precision payloads
        flat = lambda t: None if t is None else t.reshape(-1)
        mx = self.scaling_mode == _lib.NVTE_MXFP8_1D_SCALING
        sd = DType.kFloat8E8M0 if mx else None
        w.set(kGRowwiseData, flat(self.rowwise_data), pd)
        w.set(kGColumnwiseData, flat(self.columnwise_data), pd)
        w.set(kGRowwiseScaleInv, flat(self.scale_inv), sd)
        w.set(kGColumnwiseScaleInv, flat(self.columnwise_scale_inv), sd)
        if mx:
            if self.columnwise_transposed:
                api.nvte_amd_set_grouped_columnwise_transposed(w.handle, 1)
            if self.with_gemm_swizzled_scales:
                flag = C.c_uint8(1)
                api.nvte_set_grouped_tensor_param(w.handle, kGWithGEMMSwizzledScales, C.byref(flag), 1)
        w.set(kGScale, self.scale)
        w.set(kGAmax, self.amax)
        w.set(kGFirstDims, self.first_dims)
        w.set(kGLastDims, self.last_dims)
        w.set(kGTensorOffsets, self.tensor_offsets)
        return w


def splits_to_offsets(split_sizes: torch.Tensor, stride: int) -> torch.Tensor:
    """[0, s0*stride, (s0+s1)*stride, ...] on the device (ref tex.splits_to_offsets)."""
    assert split_sizes.dtype == torch.int64 and split_sizes.is_cuda
    out = torch.empty(split_sizes.numel() + 1, dtype=torch.int64, device=split_sizes.device)
    api.nvte_splits_to_offsets(split_sizes.data_ptr(), out.data_ptr(), split_sizes.numel(), int(stride),
                               current_stream_ptr())
    return out


def group_quantize(tensor: torch.Tensor, quantizer, num_tensors: int, first_dims: Optional[torch.Tensor] = None,
                   last_dims: Optional[torch.Tensor] = None, tensor_offsets: Optional[torch.Tensor] = None,
                   noop_flag: Optional[torch.Tensor] = None) -> GroupedTensor:
    """ref pytorch/csrc/extensions/cast.cpp:268-400 (tex.group_quantize), FP8 per-tensor scaling:
    current scaling = grouped amax -> (optional MAX all-reduce of the per-member amax vector) ->
    grouped scale/scale_inv -> grouped cast; delayed scaling quantizers carry a per-member scale
    vector (``quantizer.scale`` with num_tensors entries) and the amax vector is filled for the
    recipe step."""
    if tensor.dim() != 2:
        raise ValueError("Tensor must be 2D")
    if not tensor.is_contiguous():
        tensor = tensor.contiguous()
    R, Ccols = tensor.shape
    dev = tensor.device
    if (first_dims is not None or last_dims is not None) and tensor_offsets is None:
        if last_dims is not None:
            raise ValueError("tensor_offsets is required when last_dims varies")
        tensor_offsets = splits_to_offsets(first_dims, Ccols)
    fp8 = quantizer.dtype
    numel = R * Ccols
    if isinstance(quantizer, MXFP8Quantizer):
        return _group_quantize_mxfp8(tensor, quantizer, num_tensors, first_dims, tensor_offsets, noop_flag)
    out = GroupedTensor(
        num_tensors=num_tensors, logical_shape=(R, Ccols), dtype=tensor.dtype, fp8_dtype=fp8,
        rowwise_data=torch.empty(numel, dtype=torch.uint8, device=dev) if quantizer.rowwise_usage else None,
        columnwise_data=torch.empty(numel, dtype=torch.uint8, device=dev) if quantizer.columnwise_usage else None,
        scale=torch.empty(num_tensors, dtype=torch.float32, device=dev),
        amax=torch.empty(num_tensors, dtype=torch.float32, device=dev),
        first_dims=first_dims, last_dims=last_dims, tensor_offsets=tensor_offsets)
    sinv = torch.empty(num_tensors, dtype=torch.float32, device=dev)  # one buffer for both layouts
    out.scale_inv = sinv if quantizer.rowwise_usage else None
    out.columnwise_scale_inv = sinv if quantizer.columnwise_usage else None
    if numel == 0:
        return out
    gin = GroupedTensor(num_tensors=num_tensors, logical_shape=(R, Ccols), dtype=tensor.dtype,
                        rowwise_data=tensor, first_dims=first_dims, last_dims=last_dims,
                        tensor_offsets=tensor_offsets)
    wi, wo = gin.wrap(), out.wrap()
    wn = wrap_plain(noop_flag) if noop_flag is not None else None
    stream = current_stream_ptr()
    try:
        if isinstance(quantizer, Float8CurrentScalingQuantizer):
            cfg = QuantizationConfig(quantizer.force_pow_2_scales, quantizer.amax_epsilon, wn)
            api.nvte_group_compute_amax_with_config(wi.handle, wo.handle, cfg.handle, stream)
            if quantizer.with_amax_reduction:
                from .distributed import all_reduce_max_
                all_reduce_max_(out.amax, quantizer.amax_reduction_group)
            api.nvte_group_compute_scale_from_amax(wo.handle, cfg.handle, stream)
            api.nvte_group_quantize(wi.handle, wo.handle, cfg.handle, stream)
        elif isinstance(quantizer, Float8Quantizer):
            if quantizer.scale.numel() != num_tensors:
                raise ValueError("grouped delayed scaling needs one scale per member tensor")
            cfg = QuantizationConfig(False, 0.0, wn)
            out.scale.copy_(quantizer.scale)
            api.nvte_group_compute_amax_with_config(wi.handle, wo.handle, cfg.handle, stream)
            if quantizer.amax.numel() == num_tensors:
                torch.maximum(quantizer.amax, out.amax, out=quantizer.amax)
            sinv.copy_(1.0 / out.scale)
            api.nvte_group_quantize(wi.handle, wo.handle, cfg.handle, stream)
        else:
            raise NotImplementedError(f"group_quantize: {type(quantizer).__name__} is not implemented "
                                      "(FP8 per-tensor scaling only)")
    finally:
        wi.destroy(); wo.destroy()
        if wn is not None:
            wn.destroy()
    return out


def _pad(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _group_quantize_mxfp8(tensor, quantizer: MXFP8Quantizer, num_tensors, first_dims, tensor_offsets, noop_flag):
    """MXFP8 branch (ref cast/mxfp8/group_quantize_mxfp8.cuh through nvte_group_quantize): same-shape or
    varying-first-dim members, member rows multiples of 128, one launch, member shapes read on the device."""
    R, Ccols = tensor.shape
    dev = tensor.device
    if R % 128 or Ccols % 128:
        raise ValueError(f"grouped MXFP8 needs 128-multiple dims, got {R}x{Ccols}")
    row, col = quantizer.rowwise_usage, quantizer.columnwise_usage
    u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
    out = GroupedTensor(
        num_tensors=num_tensors, logical_shape=(R, Ccols), dtype=tensor.dtype, fp8_dtype=quantizer.dtype,
        scaling_mode=_lib.NVTE_MXFP8_1D_SCALING, columnwise_transposed=col and quantizer.columnwise_transposed,
        with_gemm_swizzled_scales=quantizer.optimize_for_gemm,
        rowwise_data=u8(R * Ccols) if row else None, columnwise_data=u8(R * Ccols) if col else None,
        scale_inv=u8(_pad(R, 128) * _pad(Ccols // 32, 4)) if row else None,
        columnwise_scale_inv=u8(_pad(R // 32, 4) * _pad(Ccols, 128)) if col else None,
        first_dims=first_dims, tensor_offsets=tensor_offsets)
    gin = GroupedTensor(num_tensors=num_tensors, logical_shape=(R, Ccols), dtype=tensor.dtype, rowwise_data=tensor,
                        first_dims=first_dims, tensor_offsets=tensor_offsets)
    wi, wo = gin.wrap(), out.wrap()
    wn = wrap_plain(noop_flag) if noop_flag is not None else None
    cfg = QuantizationConfig(False, 0.0, wn)
    try:
        api.nvte_group_quantize(wi.handle, wo.handle, cfg.handle, current_stream_ptr())
    finally:
        wi.destroy(); wo.destroy()
        if wn is not None:
            wn.destroy()
    return out


def _alloc_grouped_out(tensor, quantizer, num_tensors, first_dims, tensor_offsets):
    """Empty grouped output for `quantizer`: MXFP8, or FP8 per-tensor scaling with the quantizer's per-member
    scale vector (a delayed-scaling Float8Quantizer whose scale has num_tensors entries)."""
    R, Ccols = tensor.shape
    dev = tensor.device
    u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
    row, col = quantizer.rowwise_usage, quantizer.columnwise_usage
    if isinstance(quantizer, MXFP8Quantizer):
        if R % 128 or Ccols % 128:
            raise ValueError(f"grouped MXFP8 needs 128-multiple dims, got {R}x{Ccols}")
        return GroupedTensor(
            num_tensors=num_tensors, logical_shape=(R, Ccols), dtype=tensor.dtype, fp8_dtype=quantizer.dtype,
            scaling_mode=_lib.NVTE_MXFP8_1D_SCALING, columnwise_transposed=col and quantizer.columnwise_transposed,
            with_gemm_swizzled_scales=quantizer.optimize_for_gemm,
            rowwise_data=u8(R * Ccols) if row else None, columnwise_data=u8(R * Ccols) if col else None,
            scale_inv=u8(_pad(R, 128) * _pad(Ccols // 32, 4)) if row else None,
            columnwise_scale_inv=u8(_pad(R // 32, 4) * _pad(Ccols, 128)) if col else None,
            first_dims=first_dims, tensor_offsets=tensor_offsets)
    if isinstance(quantizer, Float8Quantizer) and not isinstance(quantizer, Float8CurrentScalingQuantizer):
        if quantizer.scale.numel() != num_tensors:
            raise ValueError("grouped delayed scaling needs one scale per member tensor")
        sinv = (1.0 / quantizer.scale).to(torch.float32)
        return GroupedTensor(
            num_tensors=num_tensors, logical_shape=(R, Ccols), dtype=tensor.dtype, fp8_dtype=quantizer.dtype,
            rowwise_data=u8(R * Ccols) if row else None, columnwise_data=u8(R * Ccols) if col else None,
            scale=quantizer.scale.to(torch.float32).contiguous(), amax=torch.zeros(num_tensors, dtype=torch.float32, device=dev),
            scale_inv=sinv if row else None, columnwise_scale_inv=sinv if col else None,
            first_dims=first_dims, tensor_offsets=tensor_offsets)
    raise NotImplementedError(f"grouped fused activation output for {type(quantizer).__name__} is not implemented "
                              "(MXFP8, or FP8 per-tensor scaling with per-member scales)")


def group_activation(name: Optional[str], x: torch.Tensor, quantizer, num_tensors: int,
                     first_dims: Optional[torch.Tensor] = None, *, grad: Optional[torch.Tensor] = None,
                     with_dbias: bool = False):
    """Grouped activation + cast in ONE launch over all members (ref activation.h:65-275, cast.h:167-410):
      grad is None, name given  -> nvte_group_<name>            out_i = cast(act(x_i))
      grad given                -> nvte_group_d<name>           out_i = cast(grad_i * act'(x_i))
      with_dbias                -> nvte_group_quantize_dbias[_d<name>]  (+ dbias [num_tensors, cols])
    Returns the grouped output, or (output, dbias)."""
    if name is not None and name not in ("gelu", "silu", "relu", "qgelu", "srelu"):
        raise ValueError(f"unknown grouped activation {name}")
    if x.dim() != 2:
        raise ValueError("Tensor must be 2D")
    x = x if x.is_contiguous() else x.contiguous()
    R, Ccols = x.shape
    offs = splits_to_offsets(first_dims, Ccols) if first_dims is not None else None
    out = _alloc_grouped_out(x, quantizer, num_tensors, first_dims, offs)
    hp = lambda t: GroupedTensor(num_tensors=num_tensors, logical_shape=(R, Ccols), dtype=t.dtype, rowwise_data=t,
                                 first_dims=first_dims, tensor_offsets=offs)
    wx, wo = hp(x).wrap(), out.wrap()
    tmp, gtmp = [], [wx, wo]
    stream = current_stream_ptr()
    dbias = None
    try:
        if grad is not None:
            grad = grad if grad.is_contiguous() else grad.contiguous()
            wg = hp(grad).wrap(); gtmp.append(wg)
        if with_dbias:
            dbias = torch.empty(num_tensors, Ccols, dtype=x.dtype, device=x.device)
            wdb = GroupedTensor(num_tensors=num_tensors, logical_shape=(num_tensors, Ccols), dtype=x.dtype,
                                rowwise_data=dbias).wrap()
            gtmp.append(wdb)
            wws = TensorWrapper(); tmp.append(wws)
            call = (lambda ws: api.nvte_group_quantize_dbias(wx.handle, wo.handle, wdb.handle, ws, None, stream)) \
                if grad is None else \
                (lambda ws: getattr(api, f"nvte_group_quantize_dbias_d{name}")(wg.handle, wx.handle, wo.handle, wdb.handle, ws, stream))
            if grad is None and name is not None:
                raise ValueError("dbias of a forward activation does not exist in the reference API")
            call(wws.handle)                       # first call: workspace shape only
            shape = wws.shape()
            work = torch.empty(shape, dtype=torch.float32, device=x.device)
            wws2 = wrap_plain(work); tmp.append(wws2)
            call(wws2.handle)
        elif grad is not None:
            getattr(api, f"nvte_group_d{name}")(wg.handle, wx.handle, wo.handle, stream)
        else:
            getattr(api, f"nvte_group_{name}")(wx.handle, wo.handle, stream)
    finally:
        for w in tmp + gtmp:
            w.destroy()
    return (out, dbias) if with_dbias else out


_SETUP_WS = {}
_ONES = {}
_ZEROS = {}


def get_grouped_gemm_setup_workspace_size(num_tensors: int) -> int:
    """ref pytorch/cpp_extensions/gemm.py:422"""
    return int(api.nvte_get_grouped_gemm_setup_workspace_size(num_tensors))


def _setup_workspace(device, num_tensors: int) -> torch.Tensor:
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream, num_tensors)
    ws = _SETUP_WS.get(key)
    if ws is None:
        ws = torch.empty(get_grouped_gemm_setup_workspace_size(num_tensors), dtype=torch.uint8, device=device)
        _SETUP_WS[key] = ws
    return ws


def _const(cache, val, n, device):
    key = (torch.device(device).index, n)
    t = cache.get(key)
    if t is None:
        t = torch.full((n,), val, dtype=torch.float32, device=device)
        cache[key] = t
    return t


def general_grouped_gemm_for_grouped_tensor(A: Union[GroupedTensor, List[Float8Tensor]], B: GroupedTensor,
                                            out: Union[GroupedTensor, List[torch.Tensor]], *, layout: str = "TN",
                                            accumulate: bool = False, use_split_accumulator: bool = False,
                                            bias: Optional[GroupedTensor] = None,
                                            bias_scale: Optional[torch.Tensor] = None, grad: bool = False,
                                            alpha: Optional[torch.Tensor] = None,
                                            beta: Optional[torch.Tensor] = None, sm_count: int = 0):
    """ref pytorch/cpp_extensions/gemm.py:463-620.  ``A`` may be a list of per-expert weights
    (discrete in), ``out`` a list of per-expert buffers (discrete out, wgrad)."""
    assert layout in ("TN", "NN", "NT"), f"GEMM layout {layout} not supported."
    if grad:
        raise NotImplementedError("grad is not supported for grouped_tensor GEMM yet.")
    transa, transb = layout[0] == "T", layout[1] == "T"
    discrete_out, discrete_in = isinstance(out, list), isinstance(A, list)
    if discrete_in and discrete_out:
        raise ValueError("Both A and out are discrete. This is not supported yet.")
    if discrete_out and bias is not None:
        raise ValueError("Bias is not supported when out is a list (discrete_out mode) yet. "
                         "Apply bias manually after the GEMM.")
    if bias_scale is not None and bias is None:
        raise ValueError("bias_scale requires bias to be provided.")
    G = B.num_tensors
    dev = B.device
    alpha = alpha if alpha is not None else _const(_ONES, 1.0, G, dev)
    beta = beta if beta is not None else (_const(_ONES, 1.0, G, dev) if accumulate else _const(_ZEROS, 0.0, G, dev))
    walpha, wbeta = wrap_plain(alpha), wrap_plain(beta)
    wws = wrap_plain(_setup_workspace(dev, G))
    cfg = api.nvte_create_grouped_matmul_config()
    b8 = C.c_uint8(1 if use_split_accumulator else 0)
    api.nvte_set_grouped_matmul_config_attribute(cfg, 4, C.byref(b8), 1)
    if sm_count:
        smc = C.c_int(int(sm_count))
        api.nvte_set_grouped_matmul_config_attribute(cfg, 3, C.byref(smc), C.sizeof(smc))
    stream = current_stream_ptr()
    wb = B.wrap()
    tmp = [walpha, wbeta, wws]
    gtmp = [wb]
    try:
        if discrete_out:
            wa = A.wrap(); gtmp.append(wa)
            wds = [wrap_plain(o) for o in out]; tmp += wds
            darr = (C.c_void_p * G)(*[w.handle for w in wds])
            api.nvte_grouped_gemm_with_discrete_out(wa.handle, int(transa), wb.handle, int(transb),
                                                    darr if accumulate else None, G if accumulate else 0, darr, G,
                                                    walpha.handle, wbeta.handle, wws.handle, None, cfg, stream)
        else:
            wd = out.wrap(); gtmp.append(wd)
            c_handle = wd.handle if accumulate else None
            if discrete_in:
                was = [a.wrap() for a in A]; tmp += was
                aarr = (C.c_void_p * G)(*[w.handle for w in was])
                api.nvte_grouped_gemm_with_discrete_inputA(aarr, G, int(transa), wb.handle, int(transb), c_handle,
                                                           wd.handle, walpha.handle, wbeta.handle, wws.handle, None,
                                                           cfg, stream)
            else:
                wa = A.wrap(); gtmp.append(wa)
                api.nvte_grouped_gemm(wa.handle, int(transa), wb.handle, int(transb), c_handle, wd.handle,
                                      walpha.handle, wbeta.handle, wws.handle, None, cfg, stream)
            if bias is not None:
                wbias = bias.wrap(); gtmp.append(wbias)
                if bias_scale is not None:
                    ws_ = wrap_plain(bias_scale); tmp.append(ws_)
                    api.nvte_grouped_scaled_bias_add(wd.handle, wbias.handle, ws_.handle, stream)
                else:
                    api.nvte_grouped_bias_add(wd.handle, wbias.handle, stream)
    finally:
        _lib.raw_lib.nvte_destroy_grouped_matmul_config(cfg)
        for w in tmp + gtmp:
            w.destroy()
    return out
