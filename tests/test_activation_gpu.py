"""GPU parity of the fused activation kernels (plain / gated, forward / backward, + cast + transpose + dbias).

Oracle: C restatement of ref common/util/math.h and of the gated layout of util/vectorized_pointwise.h.
Floating point: device expf/tanhf vs glibc's differ in the last bits, so high-precision outputs are
compared with the tolerance of the reference's activation tests (tests/cpp/operator/test_act.cu uses
getTolerances(otype): fp32 1e-6/5e-6 ... bf16 1e-5/1e-2), FP8 bytes must match the oracle's codes except
at fp32 rounding boundaries (<= one code step, on < 0.5 % of the elements); amax / scale_inv / dbias with
fp32 tolerances; transposes are byte-exact transposes of the row-wise payload.
"""
import numpy as np
import pytest
import torch

import oracle
from helpers import FMT, randn_input, to_oracle_input, u8
from transformer_engine_amd.activation import (GATED, UNARY, activation, clamped_dswiglu, clamped_swiglu, dactivation,
                                               dbias_dact)
from transformer_engine_amd.tensor import DType, Float8CurrentScalingQuantizer, Float8Quantizer

pytestmark = pytest.mark.gpu
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2
ORACLE_ACT = {"gelu": "gelu", "silu": "silu", "relu": "relu", "qgelu": "qgelu", "srelu": "srelu",
              "glu": "sigmoid", "geglu": "gelu", "swiglu": "silu", "reglu": "relu", "qgeglu": "qgelu", "sreglu": "srelu"}


def _ref(name, x, grad=None):
    gated = name in GATED
    mode = ("dgated" if gated else "dact") if grad is not None else ("gated" if gated else "unary")
    return oracle.activation(mode, ORACLE_ACT[name], to_oracle_input(x), to_oracle_input(grad) if grad is not None else None)


def _check_fp8(out, ref, scale, fmt, amax):
    got = oracle.fp8_to_float(u8(out._data), FMT[fmt])
    want = oracle.fp8_to_float(oracle.float_to_fp8(ref * np.float32(scale), FMT[fmt]), FMT[fmt])
    step = 2.0 ** -3 if fmt == E4 else 2.0 ** -2
    sub = 2.0 ** -9 if fmt == E4 else 2.0 ** -16
    diff = np.abs(got - want)
    assert (diff <= step * np.abs(want) + sub).all()
    assert (diff != 0).mean() < 5e-3
    assert np.isclose(amax, np.abs(ref).max(), rtol=1e-5)
    assert out._scale_inv.item() == np.float32(1.0) / np.float32(scale)
    if out._transpose is not None:
        assert np.array_equal(u8(out._transpose), u8(out._data).T)


@pytest.mark.parametrize("name", UNARY + GATED)
@pytest.mark.parametrize("N,H", [(256, 512), (300, 264), (37, 100)])      # tiles / guarded tiles / element path
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_activation_forward_high_precision(name, N, H, dtype):
    x = randn_input((N, 2 * H if name in GATED else H), dtype, 3) * 2
    y = activation(name, x)
    torch.cuda.synchronize()
    ref = _ref(name, x)
    tol = dict(rtol=5e-6, atol=2e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-5)
    assert y.shape == ref.shape
    assert np.allclose(y.float().cpu().numpy(), ref, **tol)


@pytest.mark.parametrize("name", UNARY + GATED)
@pytest.mark.parametrize("N,H", [(256, 512), (304, 264), (37, 100)])
def test_activation_backward_high_precision(name, N, H):
    x = randn_input((N, 2 * H if name in GATED else H), torch.bfloat16, 4) * 2
    g = randn_input((N, H), torch.bfloat16, 5)
    dx = dactivation(name, g, x)
    torch.cuda.synchronize()
    ref = _ref(name, x, g)
    assert dx.shape == ref.shape
    assert np.allclose(dx.float().cpu().numpy(), ref, rtol=1e-2, atol=1e-5)


@pytest.mark.parametrize("name", ["gelu", "swiglu", "srelu", "geglu"])
@pytest.mark.parametrize("fmt", [E4, E5])
@pytest.mark.parametrize("N,H", [(256, 512), (304, 264), (37, 100)])
def test_activation_forward_fp8_cast_transpose(name, fmt, N, H):
    x = randn_input((N, 2 * H if name in GATED else H), torch.bfloat16, 6) * 2
    scale, amax = torch.full((1,), 5.5, device="cuda"), torch.zeros(1, device="cuda")
    out = activation(name, x, Float8Quantizer(scale, amax, fmt))
    torch.cuda.synchronize()
    _check_fp8(out, _ref(name, x), 5.5, fmt, amax.item())


@pytest.mark.parametrize("name", ["gelu", "silu", "swiglu", "reglu", "qgeglu"])
@pytest.mark.parametrize("N,H", [(256, 512), (304, 264), (37, 100)])
def test_activation_backward_fp8_cast_transpose(name, N, H):
    x = randn_input((N, 2 * H if name in GATED else H), torch.bfloat16, 7) * 2
    g = randn_input((N, H), torch.bfloat16, 8)
    scale, amax = torch.full((1,), 3.0, device="cuda"), torch.zeros(1, device="cuda")
    out = dactivation(name, g, x, Float8Quantizer(scale, amax, E5))
    torch.cuda.synchronize()
    _check_fp8(out, _ref(name, x, g), 3.0, E5, amax.item())


@pytest.mark.parametrize("name", UNARY)
@pytest.mark.parametrize("N,H", [(512, 768), (304, 264), (37, 100), (1000, 128)])
def test_dbias_dact_fused(name, N, H):
    x = randn_input((N, H), torch.bfloat16, 9) * 2
    g = randn_input((N, H), torch.bfloat16, 10)
    scale, amax = torch.full((1,), 2.0, device="cuda"), torch.zeros(1, device="cuda")
    dbias, out = dbias_dact(name, g, x, Float8Quantizer(scale, amax, E5))
    torch.cuda.synchronize()
    ref = _ref(name, x, g)
    _check_fp8(out, ref, 2.0, E5, amax.item())
    want = ref.astype(np.float64).sum(axis=0)
    # fp32 partial sums per 128-row slab, one bf16 rounding of the result
    assert np.allclose(dbias.float().cpu().numpy(), want, rtol=2 ** -7, atol=2 ** -7 * np.abs(ref).sum(axis=0).max())


def test_quantize_dbias_uses_the_fused_pass_and_matches_oracle_bytes():
    """nvte_quantize_dbias for FP8 per-tensor scaling: bytes bit-exact as nvte_quantize, dbias = column sums."""
    from transformer_engine_amd.linear import bgrad_quantize
    for (N, H) in [(512, 768), (304, 264), (37, 100)]:
        dy = randn_input((N, H), torch.bfloat16, 11)
        scale, amax = torch.full((1,), 7.0, device="cuda"), torch.zeros(1, device="cuda")
        dbias, out = bgrad_quantize(dy, Float8Quantizer(scale, amax, E5))
        torch.cuda.synchronize()
        rc, rt, ra, rs = oracle.cast_transpose(to_oracle_input(dy), 7.0, oracle.E5M2)
        assert np.array_equal(u8(out._data), rc) and np.array_equal(u8(out._transpose), rt)
        assert amax.item() == ra and out._scale_inv.item() == rs
        ref = oracle.colsum(to_oracle_input(dy))
        assert np.allclose(dbias.float().cpu().numpy(), ref, rtol=2 ** -7, atol=2 ** -7 * np.abs(dy.float().cpu().numpy()).sum(axis=0).max())


def test_activation_current_scaling_unfuses():
    x = randn_input((128, 256), torch.bfloat16, 12)
    q = Float8CurrentScalingQuantizer(E4, torch.device("cuda"))
    out = activation("gelu", x, q)
    ref = q(activation("gelu", x))
    torch.cuda.synchronize()
    assert torch.equal(out._data, ref._data) and torch.equal(out._scale_inv, ref._scale_inv)


# ---------------------------------------------------------------- MXFP8 outputs (ref cast/mxfp8/gated_mxfp8.cuh,
# quantize_mxfp8.cuh IS_ACT / IS_DACT; tests/cpp/operator/test_cast_mxfp8_gated_swiglu.cu, test_cast_mxfp8.cu)
def _mx_quantizer(layout, fmt=E4):
    from transformer_engine_amd.tensor import MXFP8Quantizer
    q = MXFP8Quantizer(fmt, columnwise_transposed=(layout == "tcol"))
    q.optimize_for_gemm = layout == "swizzled"
    return q


def _assert_same_mxfp8(a, b, rows, cols, layout):
    """Payload and the scales of the valid region (compact layouts) / the whole arrays (swizzled: both
    come from zero-initialised arrays of the same shape) must be identical."""
    assert torch.equal(a._rowwise_data, b._rowwise_data)
    assert torch.equal(a._columnwise_data, b._columnwise_data)
    if layout == "swizzled":
        assert torch.equal(a._rowwise_scale_inv, b._rowwise_scale_inv)
        assert torch.equal(a._columnwise_scale_inv, b._columnwise_scale_inv)
    else:
        assert torch.equal(a._rowwise_scale_inv[:rows, :cols // 32], b._rowwise_scale_inv[:rows, :cols // 32])
        if layout == "tcol":
            assert torch.equal(a._columnwise_scale_inv[:cols, :rows // 32], b._columnwise_scale_inv[:cols, :rows // 32])
        else:
            assert torch.equal(a._columnwise_scale_inv[:rows // 32, :cols], b._columnwise_scale_inv[:rows // 32, :cols])


@pytest.mark.parametrize("name", ["gelu", "srelu", "swiglu", "geglu", "glu"])
@pytest.mark.parametrize("layout", ["compact", "tcol", "swizzled"])
@pytest.mark.parametrize("N,H", [(256, 512), (384, 128), (96, 160)])      # whole tiles x2 / 32x32-tile path
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_activation_forward_mxfp8_equals_unfused(name, layout, N, H, dtype):
    """The fused activation -> MXFP8 kernel must produce exactly what the unfused sequence does:
    activation in the input dtype, then nvte_quantize (the reference rounds to IType before the block
    quantize for that reason, gated_mxfp8.cuh:277-283)."""
    if layout != "compact" and (N % 128 or H % 128):
        pytest.skip("transposed / swizzled MXFP8 layouts need 128-multiples (as the plain quantize does)")
    x = randn_input((N, 2 * H if name in GATED else H), dtype, 31) * 2
    q = _mx_quantizer(layout)
    out = activation(name, x, q)
    ref = q(activation(name, x))
    torch.cuda.synchronize()
    _assert_same_mxfp8(out, ref, N, H, layout)
    # and the unfused high-precision activation is the oracle-checked one (tests above)


@pytest.mark.parametrize("name", ["gelu", "silu", "swiglu", "reglu", "qgeglu"])
@pytest.mark.parametrize("layout", ["compact", "tcol"])
@pytest.mark.parametrize("N,H", [(256, 512), (96, 160)])
@pytest.mark.parametrize("fmt", [E4, E5])
def test_activation_backward_mxfp8_equals_unfused(name, layout, N, H, fmt):
    if layout != "compact" and (N % 128 or H % 128):
        pytest.skip("transposed MXFP8 layout needs 128-multiples")
    gated = name in GATED
    x = randn_input((N, 2 * H if gated else H), torch.bfloat16, 32) * 2
    g = randn_input((N, H), torch.bfloat16, 33)
    q = _mx_quantizer(layout, fmt)
    out = dactivation(name, g, x, q)
    ref = q(dactivation(name, g, x))
    torch.cuda.synchronize()
    _assert_same_mxfp8(out, ref, N, 2 * H if gated else H, layout)


@pytest.mark.parametrize("name", ["gelu", "relu"])
@pytest.mark.parametrize("N,H", [(512, 768), (96, 160)])
def test_dbias_dact_mxfp8(name, N, H):
    x = randn_input((N, H), torch.bfloat16, 34) * 2
    g = randn_input((N, H), torch.bfloat16, 35)
    q = _mx_quantizer("compact")
    dbias, out = dbias_dact(name, g, x, q)
    ref_q = q(dactivation(name, g, x))
    torch.cuda.synchronize()
    _assert_same_mxfp8(out, ref_q, N, H, "compact")
    ref = _ref(name, x, g)
    want = ref.astype(np.float64).sum(axis=0)
    assert np.allclose(dbias.float().cpu().numpy(), want, rtol=2 ** -7, atol=2 ** -7 * np.abs(ref).sum(axis=0).max())


def test_quantize_dbias_mxfp8_fused_pass_equals_plain_quantize():
    from transformer_engine_amd.linear import bgrad_quantize
    for (N, H), layout in [((512, 768), "compact"), ((256, 384), "tcol"), ((96, 160), "compact")]:
        dy = randn_input((N, H), torch.bfloat16, 36)
        q = _mx_quantizer(layout)
        dbias, out = bgrad_quantize(dy, q)
        ref_q = q(dy)
        torch.cuda.synchronize()
        _assert_same_mxfp8(out, ref_q, N, H, layout)
        ref = oracle.colsum(to_oracle_input(dy))
        assert np.allclose(dbias.float().cpu().numpy(), ref, rtol=2 ** -7,
                           atol=2 ** -7 * np.abs(dy.float().cpu().numpy()).sum(axis=0).max())


# ---------------------------------------------------------------- clamped SwiGLU (GPT-OSS), ref activation.h:346-370,487-512
CLAMP = dict(limit=1.5, alpha=1.702)      # a limit inside the data range, so that every clamp branch is taken


@pytest.mark.parametrize("offset", [1.0, 0.25])
@pytest.mark.parametrize("N,H", [(256, 512), (304, 264), (37, 100)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_clamped_swiglu_forward_backward_high_precision(offset, N, H, dtype):
    x = randn_input((N, 2 * H), dtype, 41) * 2
    g = randn_input((N, H), dtype, 42)
    y = clamped_swiglu(x, glu_linear_offset=offset, **CLAMP)
    dx = clamped_dswiglu(g, x, glu_linear_offset=offset, **CLAMP)
    torch.cuda.synchronize()
    ref_y = oracle.clamped_swiglu(to_oracle_input(x), offset=offset, **CLAMP)
    ref_dx = oracle.clamped_swiglu(to_oracle_input(x), to_oracle_input(g), offset=offset, **CLAMP)
    # ref test_act.cu getTolerances: fp32 atol 1e-6 rtol 5e-6 (widened for device exp2/rcp: 1e-5 / 2e-5), bf16 1e-5 / 1e-2
    rtol, atol = (1e-2, 1e-5) if dtype == torch.bfloat16 else (2e-5, 1e-5)
    assert np.allclose(y.float().cpu().numpy(), ref_y, rtol=rtol, atol=atol * max(1.0, np.abs(ref_y).max()))
    assert np.allclose(dx.float().cpu().numpy(), ref_dx, rtol=rtol, atol=atol * max(1.0, np.abs(ref_dx).max()))
    # the clamp branches exist in this data: saturated activations, clamped gates with zero gate gradient
    xf = x.float().cpu().numpy()
    assert (xf[:, :H] > CLAMP["limit"]).any() and (np.abs(xf[:, H:]) > CLAMP["limit"]).any()
    assert (ref_dx[:, H:][np.abs(xf[:, H:]) > CLAMP["limit"]] == 0).all()
    assert (dx.float().cpu().numpy()[:, H:][np.abs(xf[:, H:]) > CLAMP["limit"]] == 0).all()


@pytest.mark.parametrize("N,H", [(256, 512), (96, 160)])
def test_clamped_swiglu_quantized_outputs_equal_unfused(N, H):
    x = randn_input((N, 2 * H), torch.bfloat16, 43) * 2
    g = randn_input((N, H), torch.bfloat16, 44)
    # FP8 per-tensor (delayed) scaling: codes against the oracle as for the other gated activations
    scale, amax = torch.full((1,), 3.0, device="cuda"), torch.zeros(1, device="cuda")
    out = clamped_swiglu(x, Float8Quantizer(scale, amax, E4), **CLAMP)
    torch.cuda.synchronize()
    _check_fp8(out, oracle.clamped_swiglu(to_oracle_input(x), **CLAMP), 3.0, E4, amax.item())
    amax.zero_()
    out = clamped_dswiglu(g, x, Float8Quantizer(scale, amax, E5), **CLAMP)
    torch.cuda.synchronize()
    _check_fp8(out, oracle.clamped_swiglu(to_oracle_input(x), to_oracle_input(g), **CLAMP), 3.0, E5, amax.item())
    # MXFP8: fused == activation, then nvte_quantize
    q = _mx_quantizer("compact")
    _assert_same_mxfp8(clamped_swiglu(x, q, **CLAMP), q(clamped_swiglu(x, **CLAMP)), N, H, "compact")
    _assert_same_mxfp8(clamped_dswiglu(g, x, q, **CLAMP), q(clamped_dswiglu(g, x, **CLAMP)), N, 2 * H, "compact")
