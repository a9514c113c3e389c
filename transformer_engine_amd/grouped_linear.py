"""FP8 GroupedLinear (MoE experts) forward + backward on the grouped path, member shapes on the device.

Mirrors the call sequence of ``transformer_engine/pytorch/module/grouped_linear.py`` (fused grouped
path, ``NVTE_GROUPED_LINEAR_USE_FUSED_GROUPED_GEMM=1``, :136): one grouped quantize per operand, one
grouped GEMM per pass, ``m_splits`` never leaves the GPU.

  forward   X [sum m_i, K] --group_quantize(row+col)--> X8 ;  W [G*N, K] --group_quantize (cached)--> W8
            Y_i  = X8_i W8_i^T                      grouped GEMM "TN"
  backward  dY --group_quantize(row+col, E5M2 under HYBRID)--> dY8
            dX_i = dY8_i W8_i                       grouped GEMM "NN"
            dW_i = dY8_i^T X8_i (+ main_grad_i)     grouped GEMM "NT", discrete outputs
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .grouped import (GroupedTensor, general_grouped_gemm_for_grouped_tensor, group_quantize,
                      splits_to_offsets)
from .recipe import Float8CurrentScaling, MXFP8BlockScaling, get_fp8_te_dtype
from .tensor import Float8CurrentScalingQuantizer, MXFP8Quantizer


class GroupedLinearFP8State:
    def __init__(self, recipe, num_gemms: int, device=None):
        self.recipe, self.num_gemms = recipe, num_gemms
        self.device = device or torch.device("cuda")
        f, b = get_fp8_te_dtype(recipe.fp8_format, True), get_fp8_te_dtype(recipe.fp8_format, False)
        if isinstance(recipe, Float8CurrentScaling):
            mk = lambda dt: Float8CurrentScalingQuantizer(dt, self.device, force_pow_2_scales=recipe.use_power_2_scales,
                                                          amax_epsilon=recipe.amax_epsilon)
        elif isinstance(recipe, MXFP8BlockScaling):
            # member rows (m_splits) must be multiples of 128; columnwise copies per member, transposed (amd_ext.h)
            # scales in the layout the grouped GEMM reads (ref module/grouped_linear.py:349, :864, :1905)
            def mk(dt):
                q = MXFP8Quantizer(dt, columnwise_transposed=True)
                q.optimize_for_gemm = True
                return q
        else:
            raise NotImplementedError("the grouped path implements FP8 current scaling and MXFP8")
        self.q_input, self.q_weight, self.q_grad = mk(f), mk(f), mk(b)
        self.weight_cache: Optional[GroupedTensor] = None


def _hp_out(first_dims: torch.Tensor, rows: int, cols: int, G: int, dtype) -> GroupedTensor:
    return GroupedTensor(num_tensors=G, logical_shape=(rows, cols), dtype=dtype,
                         rowwise_data=torch.empty(rows * cols, dtype=dtype, device=first_dims.device),
                         first_dims=first_dims, tensor_offsets=splits_to_offsets(first_dims, cols))


def grouped_linear_forward(x: torch.Tensor, weight: torch.Tensor, first_dims: torch.Tensor,
                           st: GroupedLinearFP8State, *, is_first_microbatch: Optional[bool] = None,
                           needs_backward: bool = True):
    """x [sum m_i, K] (rows already sorted by expert), weight [G*N, K] (experts stacked),
    first_dims int64 device tensor [G].  Returns (y [sum m_i, N], saved)."""
    G = st.num_gemms
    K = x.shape[-1]
    N = weight.shape[0] // G
    st.q_input.set_usage(rowwise=True, columnwise=needs_backward)
    x8 = group_quantize(x, st.q_input, G, first_dims=first_dims)
    if st.weight_cache is None or is_first_microbatch is None or is_first_microbatch:
        st.q_weight.set_usage(rowwise=True, columnwise=needs_backward)
        w8 = group_quantize(weight, st.q_weight, G)
        if is_first_microbatch is not None:
            st.weight_cache = w8
    else:
        w8 = st.weight_cache
    out = _hp_out(first_dims, x.shape[0], N, G, x.dtype)
    general_grouped_gemm_for_grouped_tensor(w8, x8, out, layout="TN")
    return out.rowwise_data.view(x.shape[0], N), (x8, w8, K, N)


def grouped_linear_backward(dy: torch.Tensor, saved, first_dims: torch.Tensor, st: GroupedLinearFP8State, *,
                            wgrad_out: Optional[List[torch.Tensor]] = None, accumulate_wgrad: bool = False):
    """Returns (dx [sum m_i, K], list of dW_i [N, K])."""
    x8, w8, K, N = saved
    G = st.num_gemms
    dy = dy if dy.is_contiguous() else dy.contiguous()
    st.q_grad.set_usage(rowwise=True, columnwise=True)
    dy8 = group_quantize(dy, st.q_grad, G, first_dims=first_dims)
    dx = _hp_out(first_dims, dy.shape[0], K, G, dy.dtype)
    general_grouped_gemm_for_grouped_tensor(w8, dy8, dx, layout="NN")
    if wgrad_out is None:
        wgrad_out = [torch.empty(N, K, dtype=dy.dtype, device=dy.device) for _ in range(G)]
        accumulate_wgrad = False
    general_grouped_gemm_for_grouped_tensor(x8, dy8, wgrad_out, layout="NT", accumulate=accumulate_wgrad)
    return dx.rowwise_data.view(dy.shape[0], K), wgrad_out


class _GroupedLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, first_dims, module):
        y, saved = grouped_linear_forward(x, weight, first_dims, module.fp8_state,
                                          is_first_microbatch=module._is_first_microbatch,
                                          needs_backward=module._needs_backward)
        ctx.saved_q, ctx.first_dims, ctx.module = saved, first_dims, module
        return y

    @staticmethod
    def backward(ctx, dy):
        m = ctx.module
        dx, dws = grouped_linear_backward(dy, ctx.saved_q, ctx.first_dims, m.fp8_state)
        return dx, torch.cat(dws, dim=0), None, None


class GroupedLinear(torch.nn.Module):
    """``te.GroupedLinear`` look-alike (ref pytorch/module/grouped_linear.py): ``num_gemms`` experts of
    [out_features, in_features] stacked in ONE parameter; ``forward(x, m_splits)`` takes the rows of x
    sorted by expert and the split sizes as an int64 DEVICE tensor (they are never read on the host)."""

    def __init__(self, num_gemms: int, in_features: int, out_features: int,
                 params_dtype: torch.dtype = torch.bfloat16, device="cuda", recipe=None):
        super().__init__()
        self.num_gemms, self.in_features, self.out_features = num_gemms, in_features, out_features
        self.weight = torch.nn.Parameter(torch.empty(num_gemms * out_features, in_features, dtype=params_dtype,
                                                     device=device))
        torch.nn.init.normal_(self.weight, std=0.02)
        self.fp8_state = GroupedLinearFP8State(recipe or Float8CurrentScaling(), num_gemms, device=torch.device(device))
        self._is_first_microbatch = None

    def forward(self, x: torch.Tensor, m_splits: torch.Tensor, is_first_microbatch: Optional[bool] = None):
        if not (isinstance(m_splits, torch.Tensor) and m_splits.is_cuda and m_splits.dtype == torch.int64):
            raise TypeError("m_splits must be an int64 tensor on the GPU (the grouped path never syncs with the host)")
        self._is_first_microbatch = is_first_microbatch
        self._needs_backward = torch.is_grad_enabled()
        return _GroupedLinearFn.apply(x, self.weight, m_splits, self)
