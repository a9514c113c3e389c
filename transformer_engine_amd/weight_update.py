"""FP8 weight-update path: sharded master weights -> FP8 primary weights.

Host-side mirror of the reference's ``transformer_engine/pytorch/tensor/utils.py:74-296,797-933``
(``quantize_master_weights`` for delayed scaling, current scaling and MXFP8) and of the pybind
entry points it calls (``tex.multi_tensor_*``, ``tex.mxfp8_scaling_*``,
``pytorch/csrc/extensions/multi_tensor/*.cpp``, ``optimizers/multi_tensor_apply.py``).  Every
kernel runs through the C ABI of libtransformer_engine.so; there is no CPU fallback.

Each data-parallel rank holds a 1-D shard ``[start_offset, start_offset + n)`` of every master
weight (possibly none) and a full copy of the FP8 model weight.  The exchange step is one MAX
all-reduce of the packed amaxes per recipe; FP8 block scaling and NVFP4 are other recipes and are
not provided.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from ._lib import api
from .tensor import (DType, Float8CurrentScalingQuantizer, Float8Quantizer, Float8Tensor,
                     MXFP8Quantizer, MXFP8Tensor, TensorWrapper, current_stream_ptr, fp8_max,
                     non_tn_fp8_gemm_supported, wrap_plain)

__all__ = ["multi_tensor_applier", "multi_tensor_scale", "multi_tensor_compute_scale_and_scale_inv",
           "multi_tensor_compute_scale_inv_e8m0", "multi_tensor_adam_fp8",
           "mxfp8_scaling_compute_partial_amax", "mxfp8_scaling_partial_cast",
           "quantize_master_weights", "cast_master_weights_to_fp8", "post_all_gather_processing"]


# =====================================================================================
# multi-tensor entry points (ref pytorch/csrc/extensions/multi_tensor/*.cpp)
# =====================================================================================
class _Lists:
    """NVTETensor** for ``tensor_lists`` (keeps the handles alive for the call)."""

    def __init__(self, tensor_lists: Sequence[Sequence[torch.Tensor]], dtypes=None):
        n = len(tensor_lists[0]) if tensor_lists else 0
        for lst in tensor_lists:
            if len(lst) != n:
                raise ValueError("tensor lists must have the same length")
        self.wrappers: List[List[TensorWrapper]] = []
        for i, lst in enumerate(tensor_lists):
            dt = dtypes[i] if dtypes else None
            self.wrappers.append([self._wrap(t, dt) for t in lst])
        row_t = C.c_void_p * n
        self.rows = [row_t(*[w.handle for w in ws]) for ws in self.wrappers]
        self.ptr = (C.POINTER(C.c_void_p) * len(self.rows))(
            *[C.cast(r, C.POINTER(C.c_void_p)) for r in self.rows])
        self.num_lists, self.per_list = len(self.rows), n

    @staticmethod
    def _wrap(t: Optional[torch.Tensor], dtype) -> TensorWrapper:
        if t is None:
            # an absent FP8 meta tensor (ref adam.cu:94-97 tests the pointer); only the optional
            # scale / amax / scale_inv lists may hold one - the library rejects it anywhere else
            return TensorWrapper().set_shape_only(0, DType.kFloat32, (1,))
        if not t.is_cuda:
            raise RuntimeError("multi-tensor kernels run on the GPU only (no CPU fallback in this library)")
        return TensorWrapper().set_rowwise_data(t, dtype)

    def destroy(self):
        for ws in self.wrappers:
            for w in ws:
                w.destroy()


def multi_tensor_scale(chunk_size: int, noop_flag: torch.Tensor, tensor_lists, scale: float):
    """out = in * scale over lists [in, out]; sets noop_flag (int32) on Inf/NaN inputs
    (ref multi_tensor/scale.cu:150)."""
    ls, f = _Lists(tensor_lists), wrap_plain(noop_flag)
    api.nvte_multi_tensor_scale_cuda(int(chunk_size), f.handle, ls.ptr, ls.num_lists, ls.per_list,
                                     C.c_float(scale), current_stream_ptr())
    ls.destroy(); f.destroy()


def multi_tensor_compute_scale_and_scale_inv(chunk_size: int, noop_flag: torch.Tensor, tensor_lists,
                                             max_fp8: float, force_pow_2_scales: bool, epsilon: float):
    """lists [amax, scale, scale_inv] (ref multi_tensor/compute_scale.cu:107)."""
    ls, f = _Lists(tensor_lists), wrap_plain(noop_flag)
    api.nvte_multi_tensor_compute_scale_and_scale_inv_cuda(
        int(chunk_size), f.handle, ls.ptr, ls.num_lists, ls.per_list, C.c_float(max_fp8),
        int(bool(force_pow_2_scales)), C.c_float(epsilon), current_stream_ptr())
    ls.destroy(); f.destroy()


def multi_tensor_compute_scale_inv_e8m0(chunk_size: int, noop_flag, tensor_lists):
    """lists [amax (bf16), scale_inv (uint8)] (ref multi_tensor/compute_scale.cu:122); the flag is
    unused, as in the reference binding."""
    ls = _Lists(tensor_lists)
    api.nvte_multi_tensor_compute_scale_inv_e8m0_cuda(int(chunk_size), ls.ptr, ls.num_lists, ls.per_list,
                                                      current_stream_ptr())
    ls.destroy()


def multi_tensor_adam_fp8(chunk_size: int, noop_flag: torch.Tensor, tensor_lists, lr: float, beta1: float,
                          beta2: float, epsilon: float, step: int, mode: int, bias_correction: int,
                          weight_decay: float, fp8_dtype):
    """lists [g, p_fp8, m, v, p_master, scale, amax, scale_inv] (ref multi_tensor/adam.cu:1005)."""
    fp8_dtype = DType.cast(fp8_dtype)
    ls = _Lists(tensor_lists, dtypes=[None, fp8_dtype, None, None, None, None, None, None])
    f = wrap_plain(noop_flag)
    api.nvte_multi_tensor_adam_fp8_cuda(
        int(chunk_size), f.handle, ls.ptr, ls.num_lists, ls.per_list, C.c_float(lr), C.c_float(beta1),
        C.c_float(beta2), C.c_float(epsilon), int(step), int(mode), int(bias_correction),
        C.c_float(weight_decay), int(fp8_dtype), current_stream_ptr())
    ls.destroy(); f.destroy()


class MultiTensorApply:
    """ref pytorch/optimizers/multi_tensor_apply.py: fixes the chunk size, forwards the rest."""

    def __init__(self, chunk_size: int):
        self.chunk_size = chunk_size

    def __call__(self, op, noop_flag_buffer, tensor_lists, *args):
        return op(self.chunk_size, noop_flag_buffer, tensor_lists, *args)


multi_tensor_applier = MultiTensorApply(2048 * 32)


# =====================================================================================
# shard (partial) amax / cast for MXFP8 (ref pytorch/csrc/extensions/recipe.cpp)
# =====================================================================================
def mxfp8_scaling_compute_partial_amax(inp: torch.Tensor, amax_rowwise: torch.Tensor,
                                       amax_colwise: torch.Tensor, rows: int, cols: int, start_offset: int):
    wi = wrap_plain(inp.view(-1))
    wr, wc = wrap_plain(amax_rowwise), wrap_plain(amax_colwise)
    api.nvte_mxfp8_scaling_compute_partial_amax(wi.handle, wr.handle, wc.handle, int(rows), int(cols),
                                                int(start_offset), current_stream_ptr())
    wi.destroy(); wr.destroy(); wc.destroy()


def mxfp8_scaling_partial_cast(inp: torch.Tensor, output_rowwise: torch.Tensor, output_colwise: torch.Tensor,
                               scale_inv_rowwise: torch.Tensor, scale_inv_colwise: torch.Tensor, rows: int,
                               cols: int, start_offset: int):
    wi = wrap_plain(inp.view(-1))
    wor, woc = wrap_plain(output_rowwise.view(-1)), wrap_plain(output_colwise.view(-1))
    wsr, wsc = wrap_plain(scale_inv_rowwise), wrap_plain(scale_inv_colwise)
    api.nvte_mxfp8_scaling_partial_cast(wi.handle, wor.handle, woc.handle, wsr.handle, wsc.handle, int(rows),
                                        int(cols), int(start_offset), current_stream_ptr())
    for w in (wi, wor, woc, wsr, wsc):
        w.destroy()


# =====================================================================================
# quantize_master_weights (ref pytorch/tensor/utils.py:74-181)
# =====================================================================================
def _all_reduce_max(t: torch.Tensor, group):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return  # single process: the local amax is the global one
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()  # CPU rehearsal of the N > 1 path (tests): gloo reduces on the host
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)


def _check_shard(model_weight, master_weight, start_offset):
    if start_offset is None:
        raise ValueError("start_offset must not be None when master_weight is provided")
    if start_offset < 0:
        raise ValueError(f"start_offset must be non-negative, got {start_offset}")
    end_offset = start_offset + master_weight.numel()
    if end_offset > model_weight.numel():
        raise ValueError(f"end_offset ({end_offset}) exceeds model_weight numel ({model_weight.numel()}), "
                         f"start_offset={start_offset}, master_weight numel={master_weight.numel()}")
    return end_offset


def _invalidate_transpose(w: Float8Tensor):
    # ref Float8Tensor._reset_caches: the transposed copy no longer matches the bytes
    w._transpose = None
    w._transpose_invalid = True


def quantize_master_weights(model_weights, master_weights, start_offsets, group=None,
                            fsdp_shard_model_weights=None, manual_post_all_gather_processing=False):
    """Cast each rank's master-weight shards into the FP8 model weights, with amaxes reduced over
    ``group`` so that every rank derives the same scales (ref tensor/utils.py:74-181)."""
    delayed, current, mx = [], [], []
    use_fsdp = fsdp_shard_model_weights is not None
    if not use_fsdp:
        fsdp_shard_model_weights = [None] * len(model_weights)
    for mw, master, off, frag in zip(model_weights, master_weights, start_offsets, fsdp_shard_model_weights):
        if master is not None:
            master = master.to(mw.dtype)  # ref :139-144, numerically as the non-primary-FP8 path
        q = mw._quantizer
        if isinstance(q, Float8Quantizer):
            delayed.append((mw, master, off, frag))
        elif isinstance(q, Float8CurrentScalingQuantizer):
            current.append((mw, master, off, frag))
        elif isinstance(q, MXFP8Quantizer):
            mx.append((mw, master, off, frag))
        else:
            raise ValueError(f"quantize_master_weights for {type(q)} is not supported yet")
    extra = (group, use_fsdp, manual_post_all_gather_processing)
    if delayed:
        _cast_delayed(delayed, *extra)
    if current:
        _cast_current(current, *extra)
    if mx:
        _cast_mxfp8(mx, *extra)


cast_master_weights_to_fp8 = quantize_master_weights  # deprecated name (ref :184-205)


def _flat_fp8(data: torch.Tensor, like: Float8Tensor, scale_inv: torch.Tensor) -> Float8Tensor:
    """A [1, n] rowwise-only FP8 view on raw bytes (ref Float8Quantizer.create_tensor_from_data)."""
    return Float8Tensor(shape=(1, data.numel()), dtype=like.dtype, data=data.view(1, -1),
                        fp8_scale_inv=scale_inv, fp8_dtype=like._fp8_dtype)


def _cast_delayed(params, group, use_fsdp, manual_post):
    """ref tensor/utils.py:208-296"""
    amaxes, scales, scale_invs = [], [], []
    for mw, master, off, frag in params:
        if not manual_post:
            _invalidate_transpose(mw)
        q = mw._quantizer
        amaxes.append(q.amax.view(1))
        scales.append(q.scale.view(1))
        scale_invs.append(mw._scale_inv.view(1))
        if master is None:
            continue
        end = _check_shard(mw, master, off)
        raw = frag if use_fsdp else mw._data.view(-1)[off:end]
        shard = _flat_fp8(raw, mw, torch.empty(1, dtype=torch.float32, device=raw.device))
        q.update_quantized(master.reshape(1, -1), shard)
    if not amaxes:
        return
    dev = amaxes[0].device
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    packed = torch.empty(len(amaxes), dtype=torch.float32, device=dev)
    views = [packed[i:i + 1] for i in range(len(amaxes))]
    multi_tensor_applier(multi_tensor_scale, flag, [amaxes, views], 1.0)
    _all_reduce_max(packed, group)
    multi_tensor_applier(multi_tensor_scale, flag, [views, amaxes], 1.0)
    # scale_inv of the model weights = 1 / the scale the cast used
    packed_s = torch.empty(len(scales), dtype=torch.float32, device=dev)
    sviews = [packed_s[i:i + 1] for i in range(len(scales))]
    multi_tensor_applier(multi_tensor_scale, flag, [scales, sviews], 1.0)
    torch.reciprocal(packed_s, out=packed_s)
    multi_tensor_applier(multi_tensor_scale, flag, [sviews, scale_invs], 1.0)


def _cast_current(params, group, use_fsdp, manual_post):
    """ref tensor/utils.py:299-419"""
    q0 = params[0][0]._quantizer
    dev = params[0][0].device
    for mw, _, _, _ in params:
        q = mw._quantizer
        if q.dtype != q0.dtype:
            raise ValueError(f"All model weights must have the same fp8 dtype, expected {q0.dtype} but got {q.dtype}")
        if q.force_pow_2_scales != q0.force_pow_2_scales:
            raise ValueError("All model weights must have the same force_pow_2_scales")
        if q.amax_epsilon != q0.amax_epsilon:
            raise ValueError("All model weights must have the same amax_epsilon")
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    packed = torch.zeros(len(params), dtype=torch.float32, device=dev)
    amaxes = [packed[i:i + 1] for i in range(len(params))]
    scales = [torch.empty(1, dtype=torch.float32, device=dev) for _ in params]
    scale_invs = [mw._scale_inv.view(1) for mw, _, _, _ in params]
    stream = current_stream_ptr()
    # 1. local amax of every shard this rank holds (0 for the others)
    for (mw, master, _, _), amax in zip(params, amaxes):
        if master is not None and master.numel() > 0:
            wi = wrap_plain(master.contiguous().view(-1))
            wo = TensorWrapper().set_amax(amax)  # output descriptor that only carries the amax slot
            api.nvte_compute_amax(wi.handle, wo.handle, stream)
            wi.destroy(); wo.destroy()
    # 2. global amax
    _all_reduce_max(packed, group)
    # 3. scales and the model weights' scale_inv
    multi_tensor_applier(multi_tensor_compute_scale_and_scale_inv, flag, [amaxes, scales, scale_invs],
                         fp8_max(q0.dtype), q0.force_pow_2_scales, q0.amax_epsilon)
    # 4. cast the shards with the global scale
    for (mw, master, off, frag), scale in zip(params, scales):
        if not manual_post:
            _invalidate_transpose(mw)
        if master is None:
            continue
        end = _check_shard(mw, master, off)
        raw = frag if use_fsdp else mw._data.view(-1)[off:end]
        if isinstance(raw, Float8Tensor):
            raw = raw._data
        shard = _flat_fp8(raw.view(-1), mw, torch.empty(1, dtype=torch.float32, device=dev))
        Float8Quantizer(scale, None, mw._fp8_dtype, columnwise=False).update_quantized(master.reshape(1, -1), shard)


def _cast_mxfp8(params, group, use_fsdp, manual_post):  # pylint: disable=unused-argument
    """ref tensor/utils.py:797-933"""
    dev = params[0][0].device
    master_dtype = next((m.dtype for _, m, _, _ in params if m is not None), torch.bfloat16)
    cu_r, cu_c = [0], [0]
    for mw, _, _, _ in params:
        rs, cs = mw._rowwise_scale_inv.shape, mw._columnwise_scale_inv.shape
        if len(rs) != 2:
            raise ValueError(f"rowwise_scale_inv must be 2D, got {len(rs)}D shape {tuple(rs)}")
        if len(cs) != 2:
            raise ValueError(f"columnwise_scale_inv must be 2D, got {len(cs)}D shape {tuple(cs)}")
        if mw._columnwise_transposed:
            raise ValueError("quantize_master_weights needs the reference MXFP8 layout "
                             "(columnwise_transposed=False)")
        cu_r.append(cu_r[-1] + rs[0] * rs[1])
        cu_c.append(cu_c[-1] + cs[0] * cs[1])
    # one buffer for every amax so that a single all-reduce covers the step
    packed = torch.zeros(cu_r[-1] + cu_c[-1], dtype=master_dtype, device=dev)
    am_r, am_c, si_r, si_c = [], [], [], []
    for i, (mw, master, off, _) in enumerate(params):
        rs, cs = mw._rowwise_scale_inv.shape, mw._columnwise_scale_inv.shape
        ar = packed[cu_r[i]:cu_r[i + 1]].view(rs)
        ac = packed[cu_r[-1] + cu_c[i]:cu_r[-1] + cu_c[i + 1]].view(cs)
        am_r.append(ar); am_c.append(ac)
        si_r.append(mw._rowwise_scale_inv); si_c.append(mw._columnwise_scale_inv)
        if master is not None:
            if len(mw.shape) != 2:
                raise ValueError(f"model_weight must be 2D for MXFP8 scaling, got {len(mw.shape)}D shape {mw.shape}")
            _check_shard(mw, master, off)
            h, w = mw.shape
            mxfp8_scaling_compute_partial_amax(master.contiguous(), ar, ac, h, w, off)
    _all_reduce_max(packed, group)
    multi_tensor_applier(multi_tensor_compute_scale_inv_e8m0, None, [am_r + am_c, si_r + si_c])
    for (mw, master, off, frag), sr, sc in zip(params, si_r, si_c):
        if master is None:
            continue
        end = off + master.numel()
        if use_fsdp:
            row_frag, col_frag = frag[0], frag[1]
        else:
            row_frag = mw._rowwise_data.view(-1)[off:end]
            col_frag = mw._columnwise_data.view(-1)[off:end]
        h, w = mw.shape
        mxfp8_scaling_partial_cast(master.contiguous(), row_frag, col_frag, sr, sc, h, w, off)


def post_all_gather_processing(model_weights):
    """After the all-gather of the FP8 bytes: rebuild what the GEMM reads besides them
    (ref tensor/utils.py:936-971).  Per-tensor FP8: the transposed copy (this library reports
    nvte_is_non_tn_fp8_gemm_supported() == 0); MXFP8: nothing."""
    if not isinstance(model_weights, (list, tuple)):
        model_weights = [model_weights]
    for mw in model_weights:
        if isinstance(mw, Float8Tensor):
            if not non_tn_fp8_gemm_supported():
                mw._create_transpose()
        elif isinstance(mw, MXFP8Tensor) or isinstance(mw, torch.Tensor):
            pass
        else:
            raise ValueError(f"post_processing for {type(mw)} is not supported")
