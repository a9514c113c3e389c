"""CPU-only tests of the host-side mirror (shapes, layout rules, recipe dataclasses)."""
import pytest
import torch

from transformer_engine_amd.gemm import gemm_output_shape
from transformer_engine_amd.recipe import (DelayedScaling, Float8CurrentScaling, Format, MXFP8BlockScaling,
                                           get_fp8_te_dtype)
from transformer_engine_amd.tensor import DType, MXFP8Quantizer, fp8_max, te_dtype


def test_gemm_output_shapes_follow_the_reference_table():
    # fprop TN, dgrad NN, wgrad NT for tokens M=10, in K=32, out N=48
    M, K, N = 10, 32, 48
    assert gemm_output_shape((N, K), (M, K), True, False) == (M, N)
    assert gemm_output_shape((N, K), (M, N), False, False) == (M, K)
    assert gemm_output_shape((M, K), (M, N), False, True) == (N, K)
    # leading dims of B are kept for non-transposed B (ref csrc/extensions/gemm.cpp:41-82)
    assert gemm_output_shape((N, K), (2, 5, K), True, False) == (2, 5, N)


def test_recipe_defaults_match_the_reference():
    r = DelayedScaling()
    assert (r.margin, r.fp8_format, r.amax_history_len, r.amax_compute_algo, r.reduce_amax) == \
        (0, Format.HYBRID, 1024, "max", True)                      # ref common/recipe/__init__.py:253-258
    assert get_fp8_te_dtype(Format.HYBRID, True) == DType.kFloat8E4M3
    assert get_fp8_te_dtype(Format.HYBRID, False) == DType.kFloat8E5M2
    c = Float8CurrentScaling()
    assert c.fp8_gemm_fprop.use_split_accumulator is False and c.fp8_gemm_wgrad.use_split_accumulator is True
    assert MXFP8BlockScaling().fp8_format == Format.E4M3
    assert fp8_max(DType.kFloat8E4M3) == 448.0 and fp8_max(DType.kFloat8E5M2) == 57344.0


def test_mxfp8_scale_shapes():
    # ref pytorch/tensor/mxfp8_tensor.py:132-139 / common/transformer_engine.cpp:93-115
    assert MXFP8Quantizer.scale_shapes((8192, 8192)) == ((8192, 256), (256, 8192))
    assert MXFP8Quantizer.scale_shapes((993, 512)) == ((1024, 16), (32, 512))
    assert MXFP8Quantizer.scale_shapes((16, 48)) == ((128, 4), (4, 128))
    assert MXFP8Quantizer.scale_shapes((8, 32, 1024)) == ((256, 32), (8, 1024))


def test_dtype_mapping():
    assert te_dtype(torch.bfloat16) == DType.kBFloat16 and te_dtype(torch.float32) == DType.kFloat32
    with pytest.raises(TypeError):
        te_dtype(torch.complex64)


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing on the host."""
    from transformer_engine_amd.tensor import Float8Quantizer
    q = Float8Quantizer(torch.ones(1), torch.zeros(1), DType.kFloat8E4M3)
    with pytest.raises(RuntimeError, match="GPU only"):
        q(torch.zeros(4, 4))
