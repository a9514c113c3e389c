"""MI355X-native FP8 linear hot path behind Transformer Engine's ``nvte_*`` C ABI.

Package layout (only what the path needs):
  csrc/      HIP kernels + the C ABI (built into lib/libtransformer_engine.so)
  _lib.py    ctypes binding of the C ABI
  tensor.py  Float8Quantizer / Float8CurrentScalingQuantizer / MXFP8Quantizer, Float8Tensor, MXFP8Tensor
  gemm.py    general_gemm / general_grouped_gemm
  recipe.py  recipes, DelayedScalingRecipeState, recipe step
  linear.py  Linear (fwd + bwd) driver mirroring te.Linear's call sequence
  distributed.py  TP/SP collectives on RCCL + comm/GEMM overlap
  grouped.py / grouped_linear.py  grouped (MoE) tensors, quantize and GEMM with device-side splits
  normalization.py  LayerNorm / RMSNorm with fused FP8 output (LayerNormLinear lives in linear.py)
  activation.py  plain / gated activations fused with the FP8 cast, transpose and dbias
  weight_update.py  sharded master weights -> FP8 primary weights (multi-tensor kernels, MXFP8 shard cast)
"""
from ._lib import NVTEError, api, version  # noqa: F401
from .tensor import (DType, Float8CurrentScalingQuantizer, Float8Quantizer, Float8Tensor,  # noqa: F401
                     MXFP8Quantizer, MXFP8Tensor, TensorWrapper)
from .gemm import general_gemm, general_grouped_gemm  # noqa: F401
from .recipe import (DelayedScaling, DelayedScalingRecipeState, Float8CurrentScaling, Format,  # noqa: F401
                     MXFP8BlockScaling)
