"""GPU parity: quantize / cast_transpose / current scaling / MXFP8 / transpose / dequantize /
recipe step through the C ABI, bit-exact against the CPU oracle.

Shapes are the reference's own test shapes (ref tests/cpp/operator/test_cast_transpose.cu:81-90,
test_cast.cu:78-95, test_cast_mxfp8.cu:524-533)."""
import numpy as np
import pytest
import torch

import oracle
from helpers import FMT, randn_input, to_oracle_input, u8, uniform_input
from transformer_engine_amd import recipe as te_recipe
from transformer_engine_amd._lib import NVTEError, api
from transformer_engine_amd.tensor import (DType, Float8CurrentScalingQuantizer, Float8Quantizer,
                                           MXFP8Quantizer, TensorWrapper, current_stream_ptr,
                                           wrap_plain)

pytestmark = pytest.mark.gpu

CT_SHAPES = [(2048, 12288), (768, 1024), (256, 65536), (65536, 128), (256, 256), (120, 2080),
             (8, 8), (1, 3221), (2333, 1), (1481, 677)]
IN_DTYPES = [torch.bfloat16, torch.float16, torch.float32]
FP8 = [DType.kFloat8E4M3, DType.kFloat8E5M2]


def _delayed_quantizer(scale_value, fp8_dtype, **kw):
    scale = torch.full((1,), scale_value, dtype=torch.float32, device="cuda")
    amax = torch.zeros(1, dtype=torch.float32, device="cuda")
    return Float8Quantizer(scale, amax, fp8_dtype, **kw), scale, amax


@pytest.mark.parametrize("shape", CT_SHAPES)
@pytest.mark.parametrize("in_dtype", IN_DTYPES)
@pytest.mark.parametrize("fp8", FP8)
def test_cast_transpose_delayed(shape, in_dtype, fp8):
    if in_dtype != torch.bfloat16 and shape[0] * shape[1] > 2048 * 12288 // 2:
        pytest.skip("large shapes are covered with bf16")
    x = uniform_input(shape, in_dtype, seed=hash((shape, str(in_dtype))) & 0xFFFF)
    scale_value = 0.7310791015625  # arbitrary non-power-of-two scale (ref setRandomScale)
    q, scale, amax = _delayed_quantizer(scale_value, fp8)
    out = q(x)
    torch.cuda.synchronize()
    ref_c, ref_t, ref_amax, ref_sinv = oracle.cast_transpose(to_oracle_input(x), scale_value, FMT[fp8])
    assert np.array_equal(u8(out._data).reshape(ref_c.shape), ref_c)
    assert np.array_equal(u8(out._transpose), ref_t)
    assert amax.item() == ref_amax
    assert out._scale_inv.item() == ref_sinv


@pytest.mark.parametrize("shape", [(16,), (16000,), (65536, 160), (16384, 1616), (1, 1296),
                                   (5, 4, 3, 160), (217, 256)])
@pytest.mark.parametrize("in_dtype", IN_DTYPES)
def test_cast_rowwise_only(shape, in_dtype):
    x = uniform_input(shape, in_dtype, seed=7)
    q, scale, amax = _delayed_quantizer(1.5, DType.kFloat8E4M3, columnwise=False)
    out = q(x)
    torch.cuda.synchronize()
    ref_c, _, ref_amax, ref_sinv = oracle.cast_transpose(to_oracle_input(x), 1.5, oracle.E4M3, transpose=False)
    assert out._transpose is None
    assert np.array_equal(u8(out._data).reshape(-1), ref_c.reshape(-1))
    assert amax.item() == ref_amax
    assert out._scale_inv.item() == ref_sinv


def test_cast_saturation_and_specials():
    """Values beyond the FP8 range saturate to +-max (CUDA satfinite semantics)."""
    vals = [0.0, -0.0, 1.0, 447.0, 448.0, 449.0, 464.0, 480.0, 1e6, -1e6, float("inf"), -float("inf"),
            2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -10, 2.0 ** -6, 0.0156, 17.0, 18.0, 19.0, 20.0, 57344.0,
            60000.0, 1e30, 2.0 ** -16, 2.0 ** -17, 3.0 * 2.0 ** -18]
    x = torch.tensor(vals * 16, dtype=torch.float32, device="cuda")
    for fp8 in FP8:
        q, _, amax = _delayed_quantizer(1.0, fp8, columnwise=False)
        out = q(x)
        torch.cuda.synchronize()
        ref = oracle.float_to_fp8(x.cpu().numpy(), FMT[fp8])
        assert np.array_equal(u8(out._data), ref)


def test_nan_codes():
    """NaN in -> NaN code out, sign kept (code points unpinned in the reference, see oracle header)."""
    bits = np.array([0x7FC00000, 0xFFC00000, 0x7F800001, 0xFF800001, 0x7FFFFFFF, 0xFFFFFFFF] * 8 + [0] * 16, np.uint32)
    f = bits.view(np.float32)
    x = torch.from_numpy(f).cuda()
    for fp8 in FP8:
        q, _, amax = _delayed_quantizer(1.0, fp8, columnwise=False)
        out = q(x)
        torch.cuda.synchronize()
        assert np.array_equal(u8(out._data), oracle.float_to_fp8(f, FMT[fp8]))
        assert amax.item() == 0.0   # NaN never enters amax (maxNum semantics)


def test_fp32_to_fp8_exhaustive_stride():
    """Hardware conversion vs oracle over a dense sweep of fp32 bit patterns (every 251st
    pattern, all exponents, both signs; NaN payloads excluded: that code point is unpinned)."""
    bits = np.arange(0, 2 ** 32, 251, dtype=np.uint64).astype(np.uint32)
    f = bits.view(np.float32)
    f = f[~np.isnan(f)]
    pad = (-len(f)) % 16
    f = np.concatenate([f, np.zeros(pad, np.float32)])
    x = torch.from_numpy(f).cuda()
    for fp8 in FP8:
        q, _, _ = _delayed_quantizer(1.0, fp8, columnwise=False)
        out = q(x)
        torch.cuda.synchronize()
        ref = oracle.float_to_fp8(f, FMT[fp8])
        got = u8(out._data)
        bad = np.nonzero(got != ref)[0]
        assert bad.size == 0, f"{bad.size} mismatches, first: x={f[bad[0]]!r} got={got[bad[0]]:#x} ref={ref[bad[0]]:#x}"


@pytest.mark.parametrize("shape", [(256, 256), (768, 1024), (2048, 768), (1481, 677), (8, 8)])
@pytest.mark.parametrize("fp8", FP8)
@pytest.mark.parametrize("pow2", [False, True])
def test_current_scaling(shape, fp8, pow2):
    x = uniform_input(shape, torch.bfloat16, seed=11)
    q = Float8CurrentScalingQuantizer(fp8, torch.device("cuda"), force_pow_2_scales=pow2)
    out = q(x)
    torch.cuda.synchronize()
    xin = to_oracle_input(x)
    ref_amax = oracle.amax(xin)
    ref_scale = oracle.scale_from_amax(ref_amax, FMT[fp8], pow2, 0.0)
    ref_c, ref_t, _, ref_sinv = oracle.cast_transpose(xin, float(ref_scale), FMT[fp8])
    assert out._amax.item() == ref_amax
    assert out._scale.item() == ref_scale
    assert out._scale_inv.item() == ref_sinv
    assert np.array_equal(u8(out._data), ref_c)
    assert np.array_equal(u8(out._transpose), ref_t)


def test_current_scaling_zero_and_eps():
    x = torch.zeros(128, 128, dtype=torch.bfloat16, device="cuda")
    out = Float8CurrentScalingQuantizer(DType.kFloat8E4M3, None)(x)
    torch.cuda.synchronize()
    assert out._amax.item() == 0.0 and out._scale.item() == 1.0 and out._scale_inv.item() == 1.0
    out = Float8CurrentScalingQuantizer(DType.kFloat8E4M3, None, amax_epsilon=0.5)(x)
    torch.cuda.synchronize()
    assert out._scale.item() == oracle.scale_from_amax(0.0, oracle.E4M3, False, 0.5)


def test_noop_flag_skips_quantize():
    x = uniform_input((256, 256), torch.bfloat16, seed=3)
    q, scale, amax = _delayed_quantizer(2.0, DType.kFloat8E4M3)
    out = q.make_empty(x.shape, dtype=x.dtype, device=x.device)
    out._data.fill_(0xAB); out._transpose.fill_(0xCD); out._scale_inv.fill_(-1.0)
    noop = torch.ones(1, dtype=torch.float32, device="cuda")
    q.update_quantized(x, out, noop_flag=noop)
    torch.cuda.synchronize()
    assert (out._data == 0xAB).all() and (out._transpose == 0xCD).all()
    assert out._scale_inv.item() == -1.0 and amax.item() == 0.0
    noop.zero_()
    q.update_quantized(x, out, noop_flag=noop)
    torch.cuda.synchronize()
    ref_c, ref_t, ref_amax, _ = oracle.cast_transpose(to_oracle_input(x), 2.0, oracle.E4M3)
    assert np.array_equal(u8(out._data), ref_c) and amax.item() == ref_amax


MX_SHAPES = [(128, 128), (993, 512), (1024,), (8, 32, 1024), (16, 8, 4, 512), (256, 4096), (64, 96), (16, 48)]


@pytest.mark.parametrize("shape", MX_SHAPES)
@pytest.mark.parametrize("direction", ["row", "col", "both"])
@pytest.mark.parametrize("fp8", FP8)
@pytest.mark.parametrize("in_dtype", [torch.bfloat16, torch.float32])
def test_mxfp8_quantize(shape, direction, fp8, in_dtype):
    if len(shape) < 2 and direction != "row":
        pytest.skip("columnwise needs 2 dims (ref test_cast_mxfp8.cu:191-193)")
    x = uniform_input(shape, in_dtype, seed=5)
    q = MXFP8Quantizer(fp8, rowwise=direction in ("row", "both"), columnwise=direction in ("col", "both"))
    out = q(x)
    torch.cuda.synchronize()
    rows = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    cols = shape[-1]
    ref = oracle.mxfp8(to_oracle_input(x).reshape(rows, cols), FMT[fp8], q.rowwise_usage, q.columnwise_usage)
    if q.rowwise_usage:
        assert np.array_equal(u8(out._rowwise_data).reshape(rows, cols), ref["out_row"])
        nb = (cols + 31) // 32
        assert np.array_equal(u8(out._rowwise_scale_inv)[:rows, :nb], ref["sc_row"][:rows, :nb])
    if q.columnwise_usage:
        assert np.array_equal(u8(out._columnwise_data).reshape(rows, cols), ref["out_col"])
        nb = (rows + 31) // 32
        assert np.array_equal(u8(out._columnwise_scale_inv)[:nb, :cols], ref["sc_col"][:nb, :cols])


@pytest.mark.parametrize("fp8", FP8)
def test_mxfp8_special_magnitudes(fp8):
    """Blocks that are all zero, tiny (subnormal scale) and huge."""
    x = torch.zeros(128, 128, dtype=torch.float32, device="cuda")
    x[0:32, 32:64] = 1e-38
    x[32:64, :] = torch.linspace(-1.0, 1.0, 128, device="cuda") * 3e38
    x[64:96, 0:32] = 2.0 ** -130
    x[96:, 64:] = torch.randn(32, 64, device="cuda") * 1e-3
    q = MXFP8Quantizer(fp8)
    out = q(x)
    torch.cuda.synchronize()
    ref = oracle.mxfp8(x.cpu().numpy(), FMT[fp8])
    assert np.array_equal(u8(out._rowwise_scale_inv)[:128, :4], ref["sc_row"][:128, :4])
    assert np.array_equal(u8(out._columnwise_scale_inv)[:4, :128], ref["sc_col"][:4, :128])
    assert np.array_equal(u8(out._rowwise_data), ref["out_row"])
    assert np.array_equal(u8(out._columnwise_data), ref["out_col"])


def test_mxfp8_swizzled_scales_and_swizzle_api():
    x = uniform_input((256, 512), torch.bfloat16, seed=9)
    q = MXFP8Quantizer(DType.kFloat8E4M3)
    compact = q(x)
    q2 = q.copy(); q2.optimize_for_gemm = True
    direct = q2(x)
    torch.cuda.synchronize()
    ref_r = oracle.swizzle_scales(u8(compact._rowwise_scale_inv))
    ref_c = oracle.swizzle_col_scales(u8(compact._columnwise_scale_inv))
    assert np.array_equal(u8(direct._rowwise_scale_inv).reshape(-1), ref_r)
    assert np.array_equal(u8(direct._columnwise_scale_inv).reshape(-1), ref_c)
    # nvte_swizzle_scaling_factors on the compact tensor gives the same bytes
    swz = q.make_empty(x.shape, dtype=x.dtype, device=x.device)
    swz._with_gemm_swizzled_scales = True
    wi, wo = compact.wrap(), swz.wrap()
    api.nvte_swizzle_scaling_factors(wi.handle, wo.handle, current_stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(u8(swz._rowwise_scale_inv).reshape(-1), ref_r)
    assert np.array_equal(u8(swz._columnwise_scale_inv).reshape(-1), ref_c)
    # and back
    back = q.make_empty(x.shape, dtype=x.dtype, device=x.device)
    wb = back.wrap()
    api.nvte_unswizzle_scaling_factors(wo.handle, wb.handle, current_stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(u8(back._rowwise_scale_inv), u8(compact._rowwise_scale_inv))
    assert np.array_equal(u8(back._columnwise_scale_inv), u8(compact._columnwise_scale_inv))
    for w in (wi, wo, wb):
        w.destroy()


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_dequantize_roundtrip(out_dtype):
    x = uniform_input((300, 260), torch.bfloat16, seed=21)
    q, _, _ = _delayed_quantizer(3.25, DType.kFloat8E4M3)
    t = q(x)
    dq = t.dequantize(out_dtype)
    torch.cuda.synchronize()
    ref = oracle.dequant(u8(t._data), oracle.E4M3, t._scale_inv.item())
    ref_t = torch.from_numpy(ref).to(out_dtype)
    assert torch.equal(dq.cpu(), ref_t)
    # MXFP8
    xm = uniform_input((256, 384), torch.bfloat16, seed=22)
    tm = MXFP8Quantizer(DType.kFloat8E4M3)(xm)
    dqm = tm.dequantize(out_dtype)
    torch.cuda.synchronize()
    refm = oracle.dequant_mxfp8(u8(tm._rowwise_data), oracle.E4M3, u8(tm._rowwise_scale_inv), True)
    assert torch.equal(dqm.cpu(), torch.from_numpy(refm).to(out_dtype))


@pytest.mark.parametrize("shape", [(256, 384), (1024, 8192), (100, 37), (1, 513), (640, 1)])
@pytest.mark.parametrize("dtype", [torch.uint8, torch.bfloat16, torch.float32])
def test_transpose(shape, dtype):
    g = torch.Generator().manual_seed(1)
    if dtype == torch.uint8:
        x = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8).cuda()
    else:
        x = torch.randn(shape, generator=g).to(dtype).cuda()
    out = torch.empty((shape[1], shape[0]), dtype=dtype, device="cuda")
    wi, wo = wrap_plain(x), wrap_plain(out)
    api.nvte_transpose(wi.handle, wo.handle, current_stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out, x.t().contiguous())
    wi.destroy(); wo.destroy()


@pytest.mark.parametrize("algo", ["max", "most_recent"])
@pytest.mark.parametrize("H,S", [(1024, 3), (5, 2), (1, 4), (300, 7)])
def test_delayed_scaling_update(algo, H, S):
    g = torch.Generator().manual_seed(4)
    hist = torch.rand(H, S, generator=g) * 10
    hist[0, 0] = 0.0          # amax 0 for "most_recent" keeps the old scale
    if S > 1:
        hist[:, 1] = 0.0      # all-zero column keeps the old scale for "max"
    scale = torch.rand(S, generator=g) + 0.5
    for fp8, margin in ((DType.kFloat8E4M3, 0.0), (DType.kFloat8E5M2, 2.0)):
        rec = te_recipe.DelayedScaling(margin=int(margin), amax_history_len=H, amax_compute_algo=algo)
        st = te_recipe.DelayedScalingRecipeState(rec, mode="forward", num_quantizers=S)
        st.dtype = fp8
        st.amax_history.copy_(hist); st.scale.copy_(scale)
        te_recipe.amax_and_scale_update(st)
        torch.cuda.synchronize()
        rh, rs = oracle.delayed_scaling(hist.numpy(), scale.numpy(), algo, FMT[fp8], margin)
        assert np.array_equal(st.amax_history.cpu().numpy(), rh)
        assert np.array_equal(st.scale.cpu().numpy(), rs)


def test_delayed_scaling_bulk_after_reduction():
    g = torch.Generator().manual_seed(8)
    rec = te_recipe.DelayedScaling(amax_history_len=16)
    states, hists, scales = [], [], []
    for S in (3, 2, 3, 5):
        st = te_recipe.DelayedScalingRecipeState(rec, mode="forward", num_quantizers=S)
        h = torch.rand(16, S, generator=g); h[0, 0] = 0.0
        s = torch.rand(S, generator=g) + 0.5
        st.amax_history.copy_(h); st.scale.copy_(s)
        states.append(st); hists.append(h.numpy().copy()); scales.append(s.numpy().copy())
    te_recipe.reduce_and_update_fp8_tensors(states)
    torch.cuda.synchronize()
    for st, h, s in zip(states, hists, scales):
        rh, rs = oracle.delayed_scaling_bulk(h[0].copy(), h, s, "max", oracle.E4M3, 0.0)
        assert np.array_equal(st.amax_history.cpu().numpy(), rh)
        assert np.array_equal(st.scale.cpu().numpy(), rs)


def test_errors_are_reported_not_fatal():
    from transformer_engine_amd import NVTEError
    x = torch.zeros(16, 16, dtype=torch.bfloat16, device="cuda")
    wi = wrap_plain(x)
    wo = TensorWrapper()  # nothing allocated
    with pytest.raises(NVTEError, match="Either rowwise or columnwise output data need to be allocated"):
        api.nvte_quantize(wi.handle, wo.handle, current_stream_ptr())
    wi.destroy(); wo.destroy()


@pytest.mark.parametrize("fp8", FP8)
@pytest.mark.parametrize("rowwise", [True, False])
@pytest.mark.parametrize("swizzled", [False, True])
@pytest.mark.parametrize("shape", [(128, 128), (256, 384), (1024, 640)])
def test_mxfp8_transposed_columnwise_extension(shape, swizzled, rowwise, fp8):
    """amd_ext.h extension: the columnwise MXFP8 payload stored transposed with (column, block)
    scales.  Expected bytes = the oracle's (reference-layout) columnwise result, transposed."""
    rows, cols = shape
    x = uniform_input(shape, torch.bfloat16, seed=33)
    q = MXFP8Quantizer(fp8, rowwise=rowwise, columnwise=True, columnwise_transposed=True)
    q.optimize_for_gemm = swizzled
    out = q(x)
    torch.cuda.synchronize()
    ref = oracle.mxfp8(to_oracle_input(x), FMT[fp8], rowwise, True)
    assert tuple(out._columnwise_data.shape) == (cols, rows)
    assert np.array_equal(u8(out._columnwise_data), ref["out_col"].T)
    nb = rows // 32
    if swizzled:
        # identical to the reference's swizzled columnwise scales
        assert np.array_equal(u8(out._columnwise_scale_inv).reshape(-1), oracle.swizzle_col_scales(ref["sc_col"]))
    else:
        assert np.array_equal(u8(out._columnwise_scale_inv)[:cols, :nb], ref["sc_col"][:nb, :cols].T)
    if rowwise:
        assert np.array_equal(u8(out._rowwise_data), ref["out_row"])
    # the flag round-trips through the C API and unaligned shapes are rejected loudly
    w = out.wrap()
    assert api.nvte_amd_get_columnwise_transposed(w.handle) == 1
    w.destroy()
    with pytest.raises(NVTEError):
        q(uniform_input((96, 128), torch.bfloat16, seed=1))
