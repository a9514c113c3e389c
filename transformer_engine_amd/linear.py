"""FP8 Linear forward + backward: the call sequence of the reference's ``te.Linear``.

Mirrors ``transformer_engine/pytorch/module/linear.py``: ``_linear_forward_impl`` (:262),
``_linear_backward`` (:760), ``grad_output_preprocess`` (``module/base.py:1622``) and the
weight-cache rule (``is_first_microbatch``, :450-460):

  forward   X  --quantize(row [+col])--> X8 ;  W --quantize(row+col, cached)--> W8
            Y  = general_gemm(W8, X8, "TN")            (+bias)
  backward  dY --quantize(row+col, E5M2 under HYBRID) [+ dbias] --> dY8
            dX = general_gemm(W8, dY8, "NN", grad=True)
            dW = general_gemm(X8, dY8, "NT", grad=True, accumulate=fuse_wgrad_accumulation)
  recipe    delayed scaling: amax history roll + scale update once per fwd and once per bwd

Tensor/sequence parallel variants of the same sequence live in ``distributed.py``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch

from ._lib import api
from .gemm import general_gemm, multi_gemm
from .recipe import (DelayedScaling, DelayedScalingRecipeState, Float8CurrentScaling,
                     MXFP8BlockScaling, get_fp8_te_dtype, reduce_and_update_fp8_tensors)
from .tensor import (DType, Float8CurrentScalingQuantizer, Float8Quantizer, Float8Tensor,
                     MXFP8Quantizer, MXFP8Tensor, TensorWrapper, current_stream_ptr, wrap_plain)

# quantizer slots, as in the reference (module/base.py:1073: 3 per GEMM fwd, 2 per GEMM bwd)
FWD_INPUT, FWD_WEIGHT, FWD_OUTPUT = 0, 1, 2
BWD_GRAD_OUTPUT, BWD_GRAD_INPUT = 0, 1


def bgrad_quantize(dy: torch.Tensor, quantizer):
    """dbias = sum over tokens of dY, fused with the quantize of dY
    (ref tex.bgrad_quantize, module/base.py:1716 -> nvte_quantize_dbias)."""
    out = quantizer.make_empty(dy.shape, dtype=dy.dtype, device=dy.device)
    dbias = torch.empty(dy.shape[-1], dtype=dy.dtype, device=dy.device)
    stream = current_stream_ptr()
    wi = wrap_plain(dy)
    wd = wrap_plain(dbias)
    ws = TensorWrapper()
    if isinstance(quantizer, MXFP8Quantizer):
        wo = out.wrap()
    else:
        wo = quantizer._wrap_out(out)
        if isinstance(quantizer, Float8Quantizer):
            wo.set_scale(quantizer.scale).set_amax(quantizer.amax)
        else:
            raise NotImplementedError("bgrad_quantize with current scaling: quantize then reduce "
                                      "(the reference also runs these unfused)")
    api.nvte_quantize_dbias(wi.handle, wo.handle, wd.handle, ws.handle, stream)   # workspace query
    bt = ws.get_param(0)
    shape = [bt.shape.data[i] for i in range(bt.shape.ndim)]
    workspace = torch.empty(shape, dtype=torch.float32, device=dy.device)
    ws.set_rowwise_data(workspace)
    api.nvte_quantize_dbias(wi.handle, wo.handle, wd.handle, ws.handle, stream)
    if isinstance(out, Float8Tensor):
        out._transpose_invalid = out._transpose is None
    for w in (wi, wd, ws, wo):
        w.destroy()
    return dbias, out


# which MXFP8 columnwise layout the module states ask the quantize kernels for (measured on the configs[2] step,
# DESIGN 6c); both run through the same GEMM entry point
MX_COLUMNWISE_DEFAULT = "reference"
# which per-tensor FP8 tensors of a Linear keep a transposed copy for the backward GEMMs: "none" - the GEMM reads
# row-wise payloads K-major at 95-98 % of the K-contiguous rate, the casts move 3 instead of 4 bytes per element and
# the configs[1] step is 7 % faster; "all" is the reference's data flow on hardware without non-TN FP8
# GEMMs.  Outputs and recipe state are bit-equal either way.
FP8_TRANSPOSES_DEFAULT = "none"


class LinearFP8State:
    """Per-module quantization state: recipe states + quantizers for one Linear."""

    def __init__(self, recipe, device=None, compact_mx_scales=(), mx_columnwise: Optional[str] = None,
                 fp8_transposes: Optional[str] = None):
        """compact_mx_scales: which of "input", "weight", "grad_output" keep their MXFP8 scales in the compact
        layout (the others are written GEMM-swizzled, see below).
        mx_columnwise: layout of the MXFP8 columnwise copies that dgrad / wgrad read - "reference" (the payload keeps
        the tensor's own shape, what TE's MXFP8Tensor holds; the GEMM reads it K-major), "transposed" (the
        amd_ext.h extension, K-contiguous) or "mixed" (reference for input and weight, transposed for grad_output);
        default: NVTE_AMD_MX_COLUMNWISE, else MX_COLUMNWISE_DEFAULT."""
        # per-tensor FP8: which tensors keep a transposed copy for the backward GEMMs.  "all" = the reference's
        # behaviour where non-TN FP8 GEMMs are unsupported (input, weight, grad_output).  "grad_output": only dY - dgrad
        # and wgrad then read the ROW-WISE payloads of W and X K-major (one K-major operand per GEMM, the rate of the
        # K-contiguous kernel), X is cast at 3 instead of 4 bytes per element and neither X^T nor W^T is ever stored.
        # "none": dY is cast row-wise only as well; wgrad then reads BOTH operands K-major (3-4 % slower than the
        # K-contiguous wgrad) and no FP8 transpose exists anywhere in the layer.
        # Default: NVTE_AMD_FP8_TRANSPOSES, else FP8_TRANSPOSES_DEFAULT.
        fp8_transposes = fp8_transposes or os.environ.get("NVTE_AMD_FP8_TRANSPOSES") or FP8_TRANSPOSES_DEFAULT
        # "input": only X^T (the smallest of the three) - wgrad then has ONE K-major operand (dY) like dgrad (W).
        if fp8_transposes not in ("all", "grad_output", "input", "none"):
            raise ValueError("fp8_transposes must be 'all', 'grad_output', 'input' or 'none'")
        self.fp8_transposes = fp8_transposes
        self.recipe = recipe
        self.device = device or torch.device("cuda")
        self.fwd_state: Optional[DelayedScalingRecipeState] = None
        self.bwd_state: Optional[DelayedScalingRecipeState] = None
        if isinstance(recipe, DelayedScaling):
            self.fwd_state = DelayedScalingRecipeState(recipe, mode="forward", num_quantizers=3, device=self.device)
            self.bwd_state = DelayedScalingRecipeState(recipe, mode="backward", num_quantizers=2, device=self.device)
            self.fwd_q = self.fwd_state.make_quantizers()
            self.bwd_q = self.bwd_state.make_quantizers()
        elif isinstance(recipe, Float8CurrentScaling):
            f, b = get_fp8_te_dtype(recipe.fp8_format, True), get_fp8_te_dtype(recipe.fp8_format, False)
            mk = lambda dt: Float8CurrentScalingQuantizer(dt, self.device,
                                                          force_pow_2_scales=recipe.use_power_2_scales,
                                                          amax_epsilon=recipe.amax_epsilon)
            self.fwd_q = [mk(f) for _ in range(3)]
            self.bwd_q = [mk(b) for _ in range(2)]
        elif isinstance(recipe, MXFP8BlockScaling):
            f, b = get_fp8_te_dtype(recipe.fp8_format, True), get_fp8_te_dtype(recipe.fp8_format, False)
            layout = mx_columnwise or os.environ.get("NVTE_AMD_MX_COLUMNWISE") or MX_COLUMNWISE_DEFAULT
            if layout not in ("reference", "transposed", "mixed"):
                raise ValueError(f"mx_columnwise must be 'reference', 'transposed' or 'mixed', got {layout!r}")
            # "mixed": input and weight in the reference layout (each is the ONE K-major operand of its backward GEMM,
            # which reads it at the K-contiguous rate), grad_output transposed (wgrad would otherwise read two K-major
            # operands: twice the LDS read instructions, ~20 % slower)
            tr_in = layout == "transposed"
            tr_g = layout in ("transposed", "mixed")
            self.fwd_q = [MXFP8Quantizer(f, columnwise_transposed=tr_in) for _ in range(3)]
            self.bwd_q = [MXFP8Quantizer(b, columnwise_transposed=tr_g) for _ in range(2)]
            # Scales are written in the layout the GEMM reads (ref module/linear.py:2067-2078 for the input and
            # grad_output quantizers, module/base.py:1211-1229 for the weight), except where another consumer needs
            # the compact one: an all-gather of the operand (column+SP input, row+SP grad_output, same rule as
            # the reference) or the norm kernel that writes the MXFP8 input itself (rmsnorm_api.cpp:32 / ln_api.cpp:36 check).
            for name, q in (("input", self.fwd_q[FWD_INPUT]), ("weight", self.fwd_q[FWD_WEIGHT]),
                            ("grad_output", self.bwd_q[BWD_GRAD_OUTPUT])):
                q.optimize_for_gemm = name not in compact_mx_scales
        else:
            raise TypeError(f"unsupported recipe {type(recipe)}")
        self.weight_cache = None  # quantized weight kept across microbatches

    # recipe step (delayed scaling only; the other recipes are stateless)
    def update_fwd(self, group=None):
        if self.fwd_state is not None:
            reduce_and_update_fp8_tensors([self.fwd_state], group)

    def update_bwd(self, group=None):
        if self.bwd_state is not None:
            reduce_and_update_fp8_tensors([self.bwd_state], group)

    def update_fwd_bwd(self, group=None):
        """Both directions' recipe step in ONE launch (and one packed amax all-reduce), behind the backward
        pass.  The reference steps the forward states when the outermost autocast exits and the backward
        states after the backward pass (ref pytorch/quantization.py:646-701); the two states of a layer are
        independent and the forward scales are first read by the NEXT forward pass, so the values are those
        of the two separate calls (tests/test_linear_gpu.py)."""
        states = [s for s in (self.fwd_state, self.bwd_state) if s is not None]
        if states:
            reduce_and_update_fp8_tensors(states, group)


def linear_forward(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                   st: LinearFP8State, *, is_first_microbatch: Optional[bool] = None,
                   needs_backward: bool = True, out_dtype: Optional[torch.dtype] = None):
    """Returns (y, saved) where saved = (x8, w8) for the backward pass."""
    qi, qw = st.fwd_q[FWD_INPUT], st.fwd_q[FWD_WEIGHT]
    # per-tensor FP8 with fp8_transposes="grad_output": no transposed copies of X and W (read K-major in backward)
    per_tensor = not isinstance(qi, MXFP8Quantizer)
    col = needs_backward and not (st.fp8_transposes != "all" and per_tensor)
    col_x = needs_backward and not (st.fp8_transposes in ("grad_output", "none") and per_tensor)
    qi.set_usage(rowwise=True, columnwise=col_x)                   # linear.py:399-406
    x2d = x.reshape(-1, x.shape[-1])
    x8 = qi(x2d)                                                   # linear.py:407
    # weight: quantize on the first microbatch (or always when no caching was requested)
    if st.weight_cache is None or is_first_microbatch is None or is_first_microbatch:
        qw.set_usage(rowwise=True, columnwise=col)
        w8 = qw(weight)                                            # linear.py:451-461
        if is_first_microbatch is not None:
            st.weight_cache = w8
    else:
        w8 = st.weight_cache
    y, *_ = general_gemm(w8, x8, out_dtype=out_dtype or x.dtype, layout="TN", bias=bias)  # linear.py:507
    y = y.reshape(*x.shape[:-1], weight.shape[0])
    return y, (x8, w8)


def linear_backward(dy: torch.Tensor, saved, st: LinearFP8State, *, use_bias: bool = False,
                    wgrad_out: Optional[torch.Tensor] = None, accumulate_wgrad: bool = False,
                    grad_dtype: Optional[torch.dtype] = None, requires_dgrad: bool = True,
                    requires_wgrad: bool = True, fuse_bwd_gemms: bool = False):
    """Returns (dx, dw, dbias)."""
    x8, w8 = saved
    qg = st.bwd_q[BWD_GRAD_OUTPUT]
    # linear.py:874-876 (per-tensor FP8 with fp8_transposes="none": no dY^T either, wgrad reads dY K-major)
    qg.set_usage(rowwise=True, columnwise=not (st.fp8_transposes in ("none", "input") and not isinstance(qg, MXFP8Quantizer)))
    dy2d = dy.reshape(-1, dy.shape[-1])
    dy2d = dy2d if dy2d.is_contiguous() else dy2d.contiguous()
    dbias = None
    if use_bias and not isinstance(qg, Float8CurrentScalingQuantizer):
        dbias, dy8 = bgrad_quantize(dy2d, qg)                      # base.py:1716
    else:
        dy8 = qg(dy2d)                                             # base.py:1718
        if use_bias:
            dbias = dy2d.sum(dim=0)
    gdt = grad_dtype or dy.dtype
    dx = dw = None
    if (fuse_bwd_gemms and requires_dgrad and requires_wgrad and isinstance(dy8, Float8Tensor)
            and isinstance(w8, Float8Tensor) and isinstance(x8, Float8Tensor)
            and (wgrad_out is None or wgrad_out.dtype == gdt)):
        # dgrad (linear.py:1062) and wgrad (linear.py:1225-1261) are independent: one persistent launch for
        # both (amd_ext.h nvte_amd_multi_gemm), bit-equal to the two calls.  Off by default: it saves a launch
        # and the gap behind it in eager mode (-32 us on configs[1]) but measured 0.7 % slower inside a replayed
        # HIP graph, where wgrad's half-empty last round costs little (the idle CUs leave the others more clock).
        dx, dw = multi_gemm([(w8, dy8, "NN", gdt, False),
                             (x8, dy8, "NT", wgrad_out if wgrad_out is not None else gdt, accumulate_wgrad)], grad=True)
        return dx, dw, dbias
    if requires_dgrad:
        dx, *_ = general_gemm(w8, dy8, out_dtype=gdt, layout="NN", grad=True)          # linear.py:1062
    if requires_wgrad:
        if wgrad_out is not None:
            dw, *_ = general_gemm(x8, dy8, layout="NT", grad=True, out=wgrad_out,
                                  accumulate=accumulate_wgrad)                           # linear.py:1225-1261
        else:
            dw, *_ = general_gemm(x8, dy8, out_dtype=gdt, layout="NT", grad=True)
    return dx, dw, dbias


def layernorm_linear_forward(x: torch.Tensor, ln_weight: torch.Tensor, ln_bias: Optional[torch.Tensor],
                             weight: torch.Tensor, bias: Optional[torch.Tensor], st: LinearFP8State, *,
                             eps: float = 1e-5, normalization: str = "LayerNorm", zero_centered_gamma: bool = False,
                             is_first_microbatch: Optional[bool] = None, needs_backward: bool = True):
    """``LayerNormLinear`` forward (ref pytorch/module/layernorm_linear.py:180-420): the norm writes the
    GEMM's FP8 input directly (delayed scaling / MXFP8: fused in the norm kernel; current scaling:
    norm to high precision, then amax -> scale -> cast), so the normalized activation never makes a
    bf16 round trip through HBM.  Returns (y, saved)."""
    from .normalization import layernorm_fwd, rmsnorm_fwd
    qi, qw = st.fwd_q[FWD_INPUT], st.fwd_q[FWD_WEIGHT]
    # per-tensor FP8: the norm kernel writes a transposed copy of its output only where the state keeps one
    # (fp8_transposes, as linear_forward); the backward GEMMs read the row-wise bytes K-major otherwise
    per_tensor = not isinstance(qi, MXFP8Quantizer)
    col_w = needs_backward and not (st.fp8_transposes != "all" and per_tensor)
    col_x = needs_backward and not (st.fp8_transposes in ("grad_output", "none") and per_tensor)
    qi.set_usage(rowwise=True, columnwise=col_x)
    if isinstance(qi, MXFP8Quantizer):
        qi.optimize_for_gemm = False       # the norm kernels write compact scales only (ref ln_api.cpp:36)
    x2d = x.reshape(-1, x.shape[-1])
    if normalization == "LayerNorm":
        ln8, mu, rsigma = layernorm_fwd(x2d, ln_weight, ln_bias, eps, quantizer=qi,
                                        zero_centered_gamma=zero_centered_gamma)
    elif normalization == "RMSNorm":
        ln8, mu, rsigma = rmsnorm_fwd(x2d, ln_weight, eps, quantizer=qi, zero_centered_gamma=zero_centered_gamma)
    else:
        raise ValueError(f"unsupported normalization {normalization}")
    if st.weight_cache is None or is_first_microbatch is None or is_first_microbatch:
        qw.set_usage(rowwise=True, columnwise=col_w)
        w8 = qw(weight)
        if is_first_microbatch is not None:
            st.weight_cache = w8
    else:
        w8 = st.weight_cache
    y, *_ = general_gemm(w8, ln8, out_dtype=x.dtype, layout="TN", bias=bias)
    return y.reshape(*x.shape[:-1], weight.shape[0]), (ln8, w8, x2d, mu, rsigma)


def layernorm_linear_backward(dy: torch.Tensor, saved, ln_weight: torch.Tensor, st: LinearFP8State, *,
                              normalization: str = "LayerNorm", zero_centered_gamma: bool = False,
                              use_bias: bool = False):
    """Returns (dx, dgamma, dbeta or None, dw, dbias): Linear backward, then the norm backward on dgrad
    (ref layernorm_linear.py:560-1010)."""
    from .normalization import layernorm_bwd, rmsnorm_bwd
    ln8, w8, x2d, mu, rsigma = saved
    dln, dw, dbias = linear_backward(dy, (ln8, w8), st, use_bias=use_bias, grad_dtype=ln_weight.dtype)
    dln = dln.reshape(x2d.shape)
    if normalization == "LayerNorm":
        dx, dgamma, dbeta = layernorm_bwd(dln, x2d, mu, rsigma, ln_weight, zero_centered_gamma=zero_centered_gamma)
    else:
        dx, dgamma = rmsnorm_bwd(dln, x2d, rsigma, ln_weight, zero_centered_gamma=zero_centered_gamma)
        dbeta = None
    return dx, dgamma, dbeta, dw, dbias


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, module):
        st = module.fp8_state
        y, saved = linear_forward(x, weight, bias, st, is_first_microbatch=module._is_first_microbatch,
                                  needs_backward=module._needs_backward)   # grad mode is off inside forward()
        ctx.saved_q = saved
        ctx.module = module
        ctx.use_bias = bias is not None
        ctx.x_shape = x.shape
        ctx.needs = (x.requires_grad, weight.requires_grad)
        return y

    @staticmethod
    def backward(ctx, dy):
        m = ctx.module
        wgrad_out, acc = None, False
        if m.fuse_wgrad_accumulation and hasattr(m.weight, "main_grad"):
            wgrad_out, acc = m.weight.main_grad, True
        dx, dw, dbias = linear_backward(dy, ctx.saved_q, m.fp8_state, use_bias=ctx.use_bias,
                                        wgrad_out=wgrad_out, accumulate_wgrad=acc,
                                        requires_dgrad=ctx.needs[0], requires_wgrad=ctx.needs[1])
        m.fp8_state.update_bwd(m.amax_group)
        if dx is not None:
            dx = dx.reshape(ctx.x_shape)
        if acc:
            dw = None
        return dx, dw, dbias, None


class Linear(torch.nn.Module):
    """``te.Linear`` look-alike for one GPU (same constructor names for the arguments it has)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True,
                 params_dtype: torch.dtype = torch.bfloat16, device="cuda",
                 fuse_wgrad_accumulation: bool = False, recipe=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = torch.nn.Parameter(torch.empty(out_features, in_features, dtype=params_dtype, device=device))
        torch.nn.init.normal_(self.weight, std=0.02)
        self.bias = torch.nn.Parameter(torch.zeros(out_features, dtype=params_dtype, device=device)) if bias else None
        self.fuse_wgrad_accumulation = fuse_wgrad_accumulation
        self.fp8_state = LinearFP8State(recipe or DelayedScaling(), device=torch.device(device))
        self._is_first_microbatch = None
        self.amax_group = None

    def forward(self, x: torch.Tensor, is_first_microbatch: Optional[bool] = None) -> torch.Tensor:
        self._is_first_microbatch = is_first_microbatch
        self._needs_backward = torch.is_grad_enabled()
        y = _LinearFn.apply(x, self.weight, self.bias, self)
        # the reference runs the forward recipe step when the autocast region exits
        # (quantization.py:762); one module == one region here
        self.fp8_state.update_fwd(self.amax_group)
        return y


class _LayerNormLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln_weight, ln_bias, weight, bias, module):
        y, saved = layernorm_linear_forward(
            x, ln_weight, ln_bias, weight, bias, module.fp8_state, eps=module.eps,
            normalization=module.normalization, zero_centered_gamma=module.zero_centered_gamma,
            is_first_microbatch=module._is_first_microbatch, needs_backward=module._needs_backward)
        ctx.saved_q = saved
        ctx.module = module
        ctx.ln_weight = ln_weight
        ctx.use_bias = bias is not None
        ctx.x_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        m = ctx.module
        dx, dgamma, dbeta, dw, dbias = layernorm_linear_backward(
            dy, ctx.saved_q, ctx.ln_weight, m.fp8_state, normalization=m.normalization,
            zero_centered_gamma=m.zero_centered_gamma, use_bias=ctx.use_bias)
        m.fp8_state.update_bwd(m.amax_group)
        return dx.reshape(ctx.x_shape), dgamma, dbeta, dw, dbias, None


class LayerNormLinear(torch.nn.Module):
    """``te.LayerNormLinear`` look-alike for one GPU (ref pytorch/module/layernorm_linear.py): LayerNorm /
    RMSNorm whose output is written as the GEMM's FP8 operand, then the FP8 Linear."""

    def __init__(self, in_features: int, out_features: int, eps: float = 1e-5, bias: bool = True,
                 normalization: str = "LayerNorm", zero_centered_gamma: bool = False,
                 params_dtype: torch.dtype = torch.bfloat16, device="cuda", recipe=None):
        super().__init__()
        if normalization not in ("LayerNorm", "RMSNorm"):
            raise ValueError(f"unsupported normalization {normalization}")
        self.in_features, self.out_features, self.eps = in_features, out_features, eps
        self.normalization, self.zero_centered_gamma = normalization, zero_centered_gamma
        g0 = torch.zeros if zero_centered_gamma else torch.ones
        self.layer_norm_weight = torch.nn.Parameter(g0(in_features, dtype=params_dtype, device=device))
        self.layer_norm_bias = (torch.nn.Parameter(torch.zeros(in_features, dtype=params_dtype, device=device))
                                if normalization == "LayerNorm" else None)
        self.weight = torch.nn.Parameter(torch.empty(out_features, in_features, dtype=params_dtype, device=device))
        torch.nn.init.normal_(self.weight, std=0.02)
        self.bias = torch.nn.Parameter(torch.zeros(out_features, dtype=params_dtype, device=device)) if bias else None
        self.fp8_state = LinearFP8State(recipe or DelayedScaling(), device=torch.device(device),
                                        compact_mx_scales=("input",))   # the norm kernel writes the MXFP8 input
        self._is_first_microbatch = None
        self.amax_group = None

    def forward(self, x: torch.Tensor, is_first_microbatch: Optional[bool] = None) -> torch.Tensor:
        self._is_first_microbatch = is_first_microbatch
        self._needs_backward = torch.is_grad_enabled()
        y = _LayerNormLinearFn.apply(x, self.layer_norm_weight, self.layer_norm_bias, self.weight, self.bias, self)
        self.fp8_state.update_fwd(self.amax_group)
        return y
