"""FP8 GroupedLinear (MoE experts) forward + backward on the grouped path, member shapes on the device.

Mirrors the call sequence of ``transformer_engine/pytorch/module/grouped_linear.py`` (fused grouped
path, ``NVTE_GROUPED_LINEAR_USE_FUSED_GROUPED_GEMM=1``, :136): one grouped quantize per operand, one
grouped GEMM per pass, ``m_splits`` never leaves the GPU.

  forward   X [sum m_i, K] --group_quantize(row+col)--> X8 ;  W [G*N, K] --group_quantize (cached)--> W8
            Y_i  = X8_i W8_i^T                      grouped GEMM "TN"
  backward  dY --group_quantize(row+col, E5M2 under HYBRID)--> dY8
            dX_i = dY8_i W8_i                       grouped GEMM "NN"
            dW_i = dY8_i^T X8_i (+ main_grad_i)     grouped GEMM "NT", discrete outputs

Recipes: current scaling, MXFP8 and delayed scaling (one scale / amax history per expert and operand).  The
reference's legacy path (host ``m_splits``, ``tex.split_quantize``, list GEMM; :630-697) is mirrored by
``grouped_linear_forward_list`` / ``grouped_linear_backward_list`` over the SAME recipe state.
"""
from __future__ import annotations

from typing import List, Optional

import os

import torch

from .grouped import (GroupedTensor, general_grouped_gemm_for_grouped_tensor, group_quantize,
                      splits_to_offsets)
from .recipe import (DelayedScaling, DelayedScalingRecipeState, Float8CurrentScaling, MXFP8BlockScaling,
                     get_fp8_te_dtype, reduce_and_update_fp8_tensors)
from .tensor import Float8CurrentScalingQuantizer, Float8Quantizer, MXFP8Quantizer


class GroupedLinearFP8State:
    def __init__(self, recipe, num_gemms: int, device=None, fp8_transposes: Optional[str] = None,
                 mx_columnwise: Optional[str] = None):
        """fp8_transposes (per-tensor FP8 recipes, the grouped-tensor path): "all" = transposed copies of X, W and dY per
        member, "none" = row-wise payloads only, the grouped dgrad / wgrad read them K-major (as LinearFP8State;
        default NVTE_AMD_FP8_TRANSPOSES, else linear.FP8_TRANSPOSES_DEFAULT).  The legacy list path keeps all
        transposes.  mx_columnwise (MXFP8): "reference" = the members' columnwise copies keep the members' own shape
        (what the reference's grouped quantize writes; the grouped GEMM reads them K-major), "transposed" = per-member
        transposed copies (amd_ext.h); default NVTE_AMD_MX_COLUMNWISE, else linear.MX_COLUMNWISE_DEFAULT."""
        from .linear import FP8_TRANSPOSES_DEFAULT, MX_COLUMNWISE_DEFAULT
        self.recipe, self.num_gemms = recipe, num_gemms
        self.device = device or torch.device("cuda")
        mode = fp8_transposes or os.environ.get("NVTE_AMD_FP8_TRANSPOSES") or FP8_TRANSPOSES_DEFAULT
        if mode not in ("all", "grad_output", "input", "none"):
            raise ValueError("fp8_transposes must be 'all' or 'none'")
        self.keep_transposes = mode != "none" or isinstance(recipe, MXFP8BlockScaling)
        f, b = get_fp8_te_dtype(recipe.fp8_format, True), get_fp8_te_dtype(recipe.fp8_format, False)
        if isinstance(recipe, Float8CurrentScaling):
            mk = lambda dt: Float8CurrentScalingQuantizer(dt, self.device, force_pow_2_scales=recipe.use_power_2_scales,
                                                          amax_epsilon=recipe.amax_epsilon)
        elif isinstance(recipe, MXFP8BlockScaling):
            # member rows (m_splits) must be multiples of 128; columnwise copies per member;
            # scales in the layout the grouped GEMM reads (ref module/grouped_linear.py:349, :864, :1905)
            layout = mx_columnwise or os.environ.get("NVTE_AMD_MX_COLUMNWISE") or MX_COLUMNWISE_DEFAULT
            if layout not in ("reference", "transposed", "mixed"):
                raise ValueError(f"mx_columnwise must be 'reference' or 'transposed', got {layout!r}")

            def mk(dt):
                q = MXFP8Quantizer(dt, columnwise_transposed=layout == "transposed")
                q.optimize_for_gemm = True
                return q
        elif isinstance(recipe, DelayedScaling):
            # One scale / amax-history column PER EXPERT and operand, as the reference's GroupedLinear keeps them
            # (ref module/grouped_linear.py: num_gemms x {input, weight, output} forward quantizers, num_gemms x
            # {grad_output, grad_input} backward ones).  Laid out [inputs G | weights G] so that a grouped quantizer
            # sees its G scales / amaxes as one contiguous vector (nvte_group_quantize reads scale[i] per member).
            G = num_gemms
            self.fwd_state = DelayedScalingRecipeState(recipe, mode="forward", num_quantizers=2 * G, device=self.device)
            self.bwd_state = DelayedScalingRecipeState(recipe, mode="backward", num_quantizers=G, device=self.device)
            fs, bs = self.fwd_state, self.bwd_state
            self.q_input = Float8Quantizer(fs.scale[:G], fs.amax_history[0][:G], f)
            self.q_weight = Float8Quantizer(fs.scale[G:], fs.amax_history[0][G:], f)
            self.q_grad = Float8Quantizer(bs.scale, bs.amax_history[0], b)
            self.weight_cache = None
            return
        else:
            raise NotImplementedError(f"GroupedLinear: recipe {type(recipe).__name__} is not implemented")
        self.fwd_state = self.bwd_state = None
        self.q_input, self.q_weight, self.q_grad = mk(f), mk(f), mk(b)
        self.weight_cache: Optional[GroupedTensor] = None

    def member_quantizers(self, which: str):
        """Per-expert quantizers over the same recipe state, for the list path (split_quantize): expert i gets views
        of scale[i] / amax[i] under delayed scaling, a quantizer of its own otherwise."""
        q = {"input": self.q_input, "weight": self.q_weight, "grad": self.q_grad}[which]
        if isinstance(q, Float8Quantizer):
            return [Float8Quantizer(q.scale[i:i + 1], q.amax[i:i + 1], q.dtype) for i in range(self.num_gemms)]
        return [q.copy() for _ in range(self.num_gemms)]

    def update_fwd(self, group=None):
        if self.fwd_state is not None:
            reduce_and_update_fp8_tensors([self.fwd_state], group)

    def update_bwd(self, group=None):
        if self.bwd_state is not None:
            reduce_and_update_fp8_tensors([self.bwd_state], group)


def _hp_out(first_dims: torch.Tensor, rows: int, cols: int, G: int, dtype, offsets=None) -> GroupedTensor:
    return GroupedTensor(num_tensors=G, logical_shape=(rows, cols), dtype=dtype,
                         rowwise_data=torch.empty(rows * cols, dtype=dtype, device=first_dims.device),
                         first_dims=first_dims,
                         tensor_offsets=offsets if offsets is not None else splits_to_offsets(first_dims, cols))


def grouped_linear_forward(x: torch.Tensor, weight: torch.Tensor, first_dims: torch.Tensor,
                           st: GroupedLinearFP8State, *, is_first_microbatch: Optional[bool] = None,
                           needs_backward: bool = True):
    """x [sum m_i, K] (rows already sorted by expert), weight [G*N, K] (experts stacked),
    first_dims int64 device tensor [G].  Returns (y [sum m_i, N], saved)."""
    G = st.num_gemms
    K = x.shape[-1]
    N = weight.shape[0] // G
    st.q_input.set_usage(rowwise=True, columnwise=needs_backward and st.keep_transposes)
    # the member offsets of the [*, K] and [*, N] tensors: one scan each per step (X and dX, Y and dY share them)
    offs_k, offs_n = splits_to_offsets(first_dims, K), splits_to_offsets(first_dims, N)
    x8 = group_quantize(x, st.q_input, G, first_dims=first_dims, tensor_offsets=offs_k)
    if st.weight_cache is None or is_first_microbatch is None or is_first_microbatch:
        st.q_weight.set_usage(rowwise=True, columnwise=needs_backward and st.keep_transposes)
        w8 = group_quantize(weight, st.q_weight, G)
        if is_first_microbatch is not None:
            st.weight_cache = w8
    else:
        w8 = st.weight_cache
    out = _hp_out(first_dims, x.shape[0], N, G, x.dtype, offs_n)
    general_grouped_gemm_for_grouped_tensor(w8, x8, out, layout="TN")
    return out.rowwise_data.view(x.shape[0], N), (x8, w8, K, N, offs_k, offs_n)


def grouped_linear_backward(dy: torch.Tensor, saved, first_dims: torch.Tensor, st: GroupedLinearFP8State, *,
                            wgrad_out: Optional[List[torch.Tensor]] = None, accumulate_wgrad: bool = False):
    """Returns (dx [sum m_i, K], list of dW_i [N, K])."""
    x8, w8, K, N, offs_k, offs_n = saved
    G = st.num_gemms
    dy = dy if dy.is_contiguous() else dy.contiguous()
    st.q_grad.set_usage(rowwise=True, columnwise=st.keep_transposes)
    dy8 = group_quantize(dy, st.q_grad, G, first_dims=first_dims, tensor_offsets=offs_n)
    dx = _hp_out(first_dims, dy.shape[0], K, G, dy.dtype, offs_k)
    general_grouped_gemm_for_grouped_tensor(w8, dy8, dx, layout="NN")
    if wgrad_out is None:
        wgrad_out = [torch.empty(N, K, dtype=dy.dtype, device=dy.device) for _ in range(G)]
        accumulate_wgrad = False
    general_grouped_gemm_for_grouped_tensor(x8, dy8, wgrad_out, layout="NT", accumulate=accumulate_wgrad)
    return dx.rowwise_data.view(dy.shape[0], K), wgrad_out


# ---------------------------------------------------------------------------------------
# legacy list path (ref module/grouped_linear.py:630-697, the reference's default when
# NVTE_GROUPED_LINEAR_USE_FUSED_GROUPED_GEMM is unset): host-side m_splits, tex.split_quantize into a
# list of per-expert FP8 tensors, nvte_multi_tensor_gemm over the lists
# ---------------------------------------------------------------------------------------
def split_quantize(tensor: torch.Tensor, m_splits: List[int], quantizers) -> list:
    """ref pytorch/csrc/extensions/cast.cpp (tex.split_quantize): rows [sum m_i, K] cut along dim 0, piece i
    quantized with quantizers[i] (an empty piece gives an empty FP8 tensor)."""
    if len(m_splits) != len(quantizers):
        raise ValueError("split_quantize needs one quantizer per split")
    if sum(m_splits) != tensor.shape[0]:
        raise ValueError(f"splits sum to {sum(m_splits)}, the tensor has {tensor.shape[0]} rows")
    outs, r0 = [], 0
    for m, q in zip(m_splits, quantizers):
        piece = tensor[r0:r0 + m]
        outs.append(q(piece) if m else q.make_empty(piece.shape, dtype=tensor.dtype, device=tensor.device))
        r0 += m
    return outs


def grouped_linear_forward_list(x: torch.Tensor, weights: List[torch.Tensor], m_splits: List[int],
                                st: GroupedLinearFP8State, *, is_first_microbatch: Optional[bool] = None,
                                needs_backward: bool = True):
    """Legacy path forward: returns (y [sum m_i, N], saved)."""
    from .gemm import general_grouped_gemm
    G = st.num_gemms
    qi, qw = st.member_quantizers("input"), st.member_quantizers("weight")
    for q in qi + qw:
        q.set_usage(rowwise=True, columnwise=needs_backward)
    x8 = split_quantize(x.reshape(-1, x.shape[-1]), m_splits, qi)
    cache = getattr(st, "weight_list_cache", None)
    if cache is None or is_first_microbatch is None or is_first_microbatch:
        w8 = [qw[i](weights[i]) for i in range(G)]
        if is_first_microbatch is not None:
            st.weight_list_cache = w8
    else:
        w8 = cache
    N = weights[0].shape[0]
    y = torch.empty(sum(m_splits), N, dtype=x.dtype, device=x.device)
    live = [i for i in range(G) if m_splits[i] > 0]
    if live:
        outs = list(torch.split(y, list(m_splits), dim=0))
        general_grouped_gemm([w8[i] for i in live], [x8[i] for i in live], [outs[i] for i in live], x.dtype, layout="TN")
    return y, (x8, w8, list(m_splits))


def grouped_linear_backward_list(dy: torch.Tensor, saved, st: GroupedLinearFP8State):
    """Legacy path backward: returns (dx [sum m_i, K], list of dW_i [N, K]; an empty expert's dW is zero)."""
    from .gemm import general_grouped_gemm
    x8, w8, m_splits = saved
    G = st.num_gemms
    qg = st.member_quantizers("grad")
    for q in qg:
        q.set_usage(rowwise=True, columnwise=True)
    dy8 = split_quantize(dy.reshape(-1, dy.shape[-1]), m_splits, qg)
    N, K = w8[0].shape
    dx = torch.empty(sum(m_splits), K, dtype=dy.dtype, device=dy.device)
    dws = [torch.zeros(N, K, dtype=dy.dtype, device=dy.device) if m_splits[i] == 0 else
           torch.empty(N, K, dtype=dy.dtype, device=dy.device) for i in range(G)]
    live = [i for i in range(G) if m_splits[i] > 0]
    if live:
        dxs = list(torch.split(dx, list(m_splits), dim=0))
        general_grouped_gemm([w8[i] for i in live], [dy8[i] for i in live], [dxs[i] for i in live], dy.dtype,
                             layout="NN", grad=True)
        general_grouped_gemm([x8[i] for i in live], [dy8[i] for i in live], [dws[i] for i in live], dy.dtype,
                             layout="NT", grad=True)
    return dx, dws


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
