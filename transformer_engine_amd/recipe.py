"""Recipes and the delayed-scaling recipe state.

Mirrors ``transformer_engine/common/recipe/__init__.py:253-367`` (recipe dataclasses) and the
parts of ``transformer_engine/pytorch/quantization.py`` that sit on the hot path:
``DelayedScalingRecipeState`` (:1387-1441) and ``FP8GlobalStateManager.
reduce_and_update_fp8_tensors`` (:646-701).
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _dist_flags as _flags
from . import _lib
from ._lib import api
from .tensor import (DType, Float8CurrentScalingQuantizer, Float8Quantizer, MXFP8Quantizer,
                     TensorWrapper, current_stream_ptr, wrap_plain)


class Format(enum.Enum):
    """ref common/recipe/__init__.py: (fwd dtype, bwd dtype)"""
    E4M3 = (DType.kFloat8E4M3, DType.kFloat8E4M3)
    E5M2 = (DType.kFloat8E5M2, DType.kFloat8E5M2)
    HYBRID = (DType.kFloat8E4M3, DType.kFloat8E5M2)


def get_fp8_te_dtype(fmt: Format, fprop_tensor: bool = True) -> DType:
    """ref pytorch/quantization.py:271-302"""
    return fmt.value[0] if fprop_tensor else fmt.value[1]


@dataclass
class DelayedScaling:
    margin: int = 0
    fp8_format: Format = Format.HYBRID
    amax_history_len: int = 1024
    amax_compute_algo: str = "max"
    reduce_amax: bool = True


@dataclass
class MMParams:
    use_split_accumulator: bool = True


@dataclass
class Float8CurrentScaling:
    fp8_format: Format = Format.HYBRID
    use_power_2_scales: bool = False
    amax_epsilon: float = 0.0
    fp8_gemm_fprop: MMParams = field(default_factory=lambda: MMParams(False))
    fp8_gemm_dgrad: MMParams = field(default_factory=lambda: MMParams(True))
    fp8_gemm_wgrad: MMParams = field(default_factory=lambda: MMParams(True))


@dataclass
class MXFP8BlockScaling:
    fp8_format: Format = Format.E4M3


class DelayedScalingRecipeState:
    """scale [S] and amax_history [H, S] for the S quantizers of one module direction
    (ref pytorch/quantization.py:1387-1441).  Quantizer i gets VIEWS scale[i] and
    amax_history[0][i]."""

    def __init__(self, recipe: DelayedScaling, *, mode: str, num_quantizers: int = 1,
                 device: Optional[torch.device] = None):
        assert mode in ("forward", "backward")
        self.recipe = recipe
        self.mode = mode
        self.num_quantizers = num_quantizers
        self.dtype = get_fp8_te_dtype(recipe.fp8_format, mode == "forward")
        device = device or torch.device("cuda")
        self.scale = torch.ones(num_quantizers, dtype=torch.float32, device=device)
        self.amax_history = torch.zeros(recipe.amax_history_len, num_quantizers,
                                        dtype=torch.float32, device=device)

    def make_quantizers(self) -> List[Float8Quantizer]:
        return [Float8Quantizer(self.scale[i], self.amax_history[0][i].reshape((1,)), self.dtype)
                for i in range(self.num_quantizers)]


def amax_and_scale_update(state: DelayedScalingRecipeState) -> None:
    """Single-tensor recipe step, in place (nvte_delayed_scaling_recipe_amax_and_scale_update)."""
    wh = wrap_plain(state.amax_history)
    ws = wrap_plain(state.scale)
    api.nvte_delayed_scaling_recipe_amax_and_scale_update(
        wh.handle, ws.handle, wh.handle, ws.handle, state.recipe.amax_compute_algo.encode(),
        int(state.dtype), float(state.recipe.margin), current_stream_ptr())
    wh.destroy(); ws.destroy()


def fused_amax_and_scale_update_after_reduction(amax_reduction_buffer: torch.Tensor,
                                                amax_histories: List[torch.Tensor],
                                                scales: List[torch.Tensor],
                                                amax_compute_algo: str, fp8_dtype,
                                                margin: float) -> None:
    """ref pytorch/csrc/extensions/recipe.cpp:32 (tex.fused_amax_and_scale_update_after_reduction)"""
    n = len(amax_histories)
    # None = nothing was reduced: the kernel then takes each history's own row 0 as the new amax
    wb = wrap_plain(amax_reduction_buffer) if amax_reduction_buffer is not None else TensorWrapper()
    whs = [wrap_plain(h) for h in amax_histories]
    wss = [wrap_plain(s) for s in scales]
    H = (C.c_void_p * n)(*[w.handle for w in whs])
    S = (C.c_void_p * n)(*[w.handle for w in wss])
    try:
        if isinstance(fp8_dtype, (list, tuple)):
            # one format per tensor (amd_ext.h): forward and backward states of a step in one launch
            D = (C.c_int * n)(*[int(d) for d in fp8_dtype])
            api.nvte_amd_delayed_scaling_after_reduction_multi(wb.handle, H, S, D, n, amax_compute_algo.encode(),
                                                               float(margin), current_stream_ptr())
        else:
            api.nvte_amd_delayed_scaling_after_reduction(wb.handle, H, S, n, amax_compute_algo.encode(),
                                                         int(fp8_dtype), float(margin),
                                                         current_stream_ptr())
    finally:
        for w in [wb] + whs + wss:
            w.destroy()


def reduce_and_update_fp8_tensors(states: List[DelayedScalingRecipeState], group=None) -> None:
    """One fwd (or bwd) recipe step for a list of module states that share a recipe
    (ref pytorch/quantization.py:646-701): concat amaxes -> ONE all_reduce(MAX) -> bulk update."""
    if not states:
        return
    recipe = states[0].recipe
    import torch.distributed as dist
    # group=None means the default (world) group, as in torch.distributed
    reduce = recipe.reduce_amax and _flags.collectives_active(group)
    # single process: no packed buffer (and no copy kernel) at all
    contiguous_amax = torch.cat([s.amax_history[0] for s in states]) if reduce else None
    if reduce:
        if contiguous_amax.is_cuda and dist.get_backend(group) == "gloo":
            # CPU rehearsal backend: reduce on the host (gloo's own CUDA staging was seen to
            # stall when two ranks share one GPU); RCCL reduces the device buffer in place
            host = contiguous_amax.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
            contiguous_amax.copy_(host)
        else:
            dist.all_reduce(contiguous_amax, op=dist.ReduceOp.MAX, group=group)
    dtypes = [s.dtype for s in states]
    fused_amax_and_scale_update_after_reduction(
        contiguous_amax, [s.amax_history for s in states], [s.scale for s in states],
        recipe.amax_compute_algo, dtypes[0] if all(d == dtypes[0] for d in dtypes) else dtypes, recipe.margin)


def make_quantizers(recipe, *, mode: str, num_quantizers: int, device=None):
    """Quantizer factory per recipe (ref pytorch/quantization.py RecipeState.create)."""
    if isinstance(recipe, DelayedScaling):
        st = DelayedScalingRecipeState(recipe, mode=mode, num_quantizers=num_quantizers, device=device)
        return st, st.make_quantizers()
    dtype = get_fp8_te_dtype(recipe.fp8_format, mode == "forward")
    if isinstance(recipe, Float8CurrentScaling):
        return None, [Float8CurrentScalingQuantizer(dtype, device,
                                                    force_pow_2_scales=recipe.use_power_2_scales,
                                                    amax_epsilon=recipe.amax_epsilon)
                      for _ in range(num_quantizers)]
    if isinstance(recipe, MXFP8BlockScaling):
        return None, [MXFP8Quantizer(dtype) for _ in range(num_quantizers)]
    raise TypeError(f"unsupported recipe {type(recipe)}")
