"""GPU parity of the scaled FP8 / MXFP8 GEMM through nvte_cublas_gemm_v2.

Oracle: matmul of the dequantized operands (double accumulation), the contract the reference
itself tests against (ref tests/pytorch/utils.py:97-104; SURVEY 8c: bit-level GEMM parity is
unpinned in the reference).  Tolerances are written at each assert.  Layout/indexing bugs
are caught by the exact small-integer tests, where every product and sum is exact in fp32.
"""
import numpy as np
import pytest
import torch

import oracle
from helpers import FMT, randn_input, u8
from transformer_engine_amd import general_gemm
from transformer_engine_amd.tensor import (DType, Float8CurrentScalingQuantizer, Float8Tensor,
                                           MXFP8Quantizer, MXFP8Tensor)

pytestmark = pytest.mark.gpu

E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2


def _int_fp8_tensor(shape, fp8, seed, lo=-4, hi=5, scale_inv=1.0, columnwise=True):
    """Float8Tensor whose bytes encode small integers exactly (no quantize kernel involved)."""
    g = np.random.default_rng(seed)
    vals = g.integers(lo, hi, size=shape).astype(np.float32)
    codes = oracle.float_to_fp8(vals, FMT[fp8])
    data = torch.from_numpy(codes).cuda()
    tr = torch.from_numpy(np.ascontiguousarray(codes.T)).cuda() if columnwise else None
    sinv = torch.full((1,), scale_inv, dtype=torch.float32, device="cuda")
    t = Float8Tensor(shape=shape, dtype=torch.float32, data=data, fp8_scale_inv=sinv, fp8_dtype=fp8,
                     data_transpose=tr)
    return t, vals


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (256, 512, 256), (512, 256, 1024), (32, 32, 64),
                                   (300, 200, 160), (1, 16, 16), (257, 255, 144), (1024, 768, 4096)])
@pytest.mark.parametrize("fa,fb", [(E4, E4), (E4, E5), (E5, E4)])
def test_gemm_exact_small_integers_TN(M, N, K, fa, fb):
    W, w = _int_fp8_tensor((N, K), fa, 1)       # A = weight [N, K]
    X, x = _int_fp8_tensor((M, K), fb, 2)       # B = input  [M, K]
    y, *_ = general_gemm(W, X, out_dtype=torch.float32, layout="TN")
    torch.cuda.synchronize()
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    assert y.shape == (M, N)
    assert np.array_equal(y.cpu().numpy().astype(np.float64), ref)   # exact


def test_gemm_exact_all_layouts_and_scales():
    M, N, K = 384, 256, 320
    W, w = _int_fp8_tensor((N, K), E4, 3, scale_inv=0.5)
    X, x = _int_fp8_tensor((M, K), E4, 4, scale_inv=0.25)
    dY, dy = _int_fp8_tensor((M, N), E5, 5, scale_inv=2.0)
    y, *_ = general_gemm(W, X, out_dtype=torch.float32, layout="TN")
    dx, *_ = general_gemm(W, dY, out_dtype=torch.float32, layout="NN")
    dw, *_ = general_gemm(X, dY, out_dtype=torch.float32, layout="NT")
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), (x @ w.T * 0.125).astype(np.float32))
    assert np.array_equal(dx.cpu().numpy(), (dy @ w * 1.0).astype(np.float32))
    assert np.array_equal(dw.cpu().numpy(), (dy.T @ x * 0.5).astype(np.float32))


def test_gemm_accumulate_alpha_beta_bias():
    M, N, K = 256, 384, 256
    W, w = _int_fp8_tensor((N, K), E4, 6)
    X, x = _int_fp8_tensor((M, K), E4, 7)
    base = torch.arange(M * N, dtype=torch.float32, device="cuda").reshape(M, N) * 0.5
    out = base.clone()
    general_gemm(W, X, layout="TN", out=out, accumulate=True)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), (x @ w.T + base.cpu().numpy()).astype(np.float32))
    out = base.clone()
    general_gemm(W, X, layout="TN", out=out, accumulate=True, alpha=2.0, beta=0.5)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), (2.0 * (x @ w.T) + 0.5 * base.cpu().numpy()).astype(np.float32))
    bias = torch.arange(N, dtype=torch.bfloat16, device="cuda")
    y, *_ = general_gemm(W, X, out_dtype=torch.float32, layout="TN", bias=bias)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), (x @ w.T + bias.float().cpu().numpy()[None, :]).astype(np.float32))


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(128, 96, 256), (2048, 3072, 768), (784, 1040, 528)])
def test_gemm_quantized_random_vs_dequantized_matmul(out_dtype, M, N, K):
    x = randn_input((M, K), torch.bfloat16, 11)
    w = randn_input((N, K), torch.bfloat16, 12, scale=0.05)
    dy = randn_input((M, N), torch.bfloat16, 13, scale=0.01)
    qx = Float8CurrentScalingQuantizer(E4, None)(x)
    qw = Float8CurrentScalingQuantizer(E4, None)(w)
    qdy = Float8CurrentScalingQuantizer(E5, None)(dy)
    y, *_ = general_gemm(qw, qx, out_dtype=out_dtype, layout="TN")
    dx, *_ = general_gemm(qw, qdy, out_dtype=out_dtype, layout="NN")
    dw, *_ = general_gemm(qx, qdy, out_dtype=out_dtype, layout="NT")
    torch.cuda.synchronize()
    fx = oracle.dequant(u8(qx._data), oracle.E4M3, qx._scale_inv.item())
    fw = oracle.dequant(u8(qw._data), oracle.E4M3, qw._scale_inv.item())
    fdy = oracle.dequant(u8(qdy._data), oracle.E5M2, qdy._scale_inv.item())
    refs = {"y": oracle.gemm_tn(fx, fw), "dx": oracle.gemm_tn(fdy, np.ascontiguousarray(fw.T)),
            "dw": oracle.gemm_tn(np.ascontiguousarray(fdy.T), np.ascontiguousarray(fx.T))}
    # Tolerance (stated): bf16/fp16 outputs must match the oracle rounded to the output dtype
    # within rtol 1.6e-2 / atol 1e-3-of-absmax (ref tests/pytorch/utils.py:97-98: bf16 tolerance);
    # fp32 outputs within rtol 1e-3, atol 1e-3 * absmax: the gfx950 FP8 MFMA sums the 64 products
    # of one instruction with reduced internal precision (measured 2.3e-5 * absmax at K=256 vs
    # 2.5e-7 for an fp32 matmul), the analogue of cuBLAS "fast accumulation".
    for name, got in (("y", y), ("dx", dx), ("dw", dw)):
        ref = refs[name]
        amax = float(np.abs(ref).max())
        g = got.float().cpu().numpy()
        if out_dtype == torch.float32:
            assert np.allclose(g, ref, rtol=1e-3, atol=1e-3 * amax), name
        else:
            r = torch.from_numpy(ref).to(out_dtype).float().numpy()
            assert np.allclose(g, r, rtol=1.6e-2, atol=1e-3 * amax), name


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (256, 256, 512), (384, 640, 1024), (130, 70, 96), (1024, 1024, 2048)])
@pytest.mark.parametrize("swizzled", [False, True])
def test_mxfp8_gemm_TN(M, N, K, swizzled):
    x = randn_input((M, K), torch.bfloat16, 21)
    w = randn_input((N, K), torch.bfloat16, 22, scale=0.1)
    # make block magnitudes very different so that a wrong scale/byte pairing is obvious
    x[:, : K // 2] *= 64.0
    w[: N // 2] *= 1e-3
    q = MXFP8Quantizer(E4, columnwise=False)
    q.optimize_for_gemm = swizzled
    qx, qw = q(x), q(w)
    y, *_ = general_gemm(qw, qx, out_dtype=torch.float32, layout="TN")
    torch.cuda.synchronize()
    if swizzled:
        q2 = MXFP8Quantizer(E4, columnwise=False)
        cx, cw = q2(x), q2(w)
        torch.cuda.synchronize()
        assert torch.equal(cx._rowwise_data, qx._rowwise_data)
        sx, sw = u8(cx._rowwise_scale_inv), u8(cw._rowwise_scale_inv)
    else:
        sx, sw = u8(qx._rowwise_scale_inv), u8(qw._rowwise_scale_inv)
    fx = oracle.dequant_mxfp8(u8(qx._rowwise_data), oracle.E4M3, sx, True)
    fw = oracle.dequant_mxfp8(u8(qw._rowwise_data), oracle.E4M3, sw, True)
    ref = oracle.gemm_tn(fx, fw)
    amax = float(np.abs(ref).max())
    # Tolerance (stated): fp32 output, rtol 1e-3 / atol 1e-3 * absmax (MFMA internal precision, as above).
    assert np.allclose(y.cpu().numpy(), ref, rtol=1e-3, atol=1e-3 * amax)


def test_gemm_empty_operand_zero_fills():
    W, _ = _int_fp8_tensor((64, 32), E4, 1)
    X = Float8Tensor(shape=(0, 32), dtype=torch.float32,
                     data=torch.empty((0, 32), dtype=torch.uint8, device="cuda"),
                     fp8_scale_inv=torch.ones(1, device="cuda"), fp8_dtype=E4)
    y, *_ = general_gemm(W, X, out_dtype=torch.float32, layout="TN")
    assert y.shape == (0, 64)


def test_gemm_rejects_bad_k():
    from transformer_engine_amd import NVTEError
    W, _ = _int_fp8_tensor((64, 40), E4, 1)
    X, _ = _int_fp8_tensor((64, 40), E4, 2)
    with pytest.raises(NVTEError, match="Leading dimension requirement"):
        general_gemm(W, X, out_dtype=torch.float32, layout="TN")


@pytest.mark.parametrize("M,N,K", [
    (4352, 4352, 256),    # 17 x 17 = 289 tiles > 256 CUs: some workgroups walk two tiles; 2 K tiles
    (4400, 4200, 144),    # ragged edge tiles on both sides, 306 tiles, K tail inside the 2nd K tile
    (8192, 4608, 384),    # 576 tiles: 2-3 tiles per workgroup, odd number (3) of K tiles
    (5000, 5000, 128),    # single K tile: one workgroup per tile path, 400 tiles
])
def test_gemm_exact_persistent_multi_tile(M, N, K):
    """The persistent kernel pipelines K tiles across output-tile boundaries; small-integer data
    makes every product and sum exact, so any slip of a slot, tile or fragment shows as a
    mismatch.  Reference computed on the GPU in fp64 (exact for these magnitudes)."""
    W, w = _int_fp8_tensor((N, K), E4, 11, lo=-3, hi=4)
    X, x = _int_fp8_tensor((M, K), E5, 12, lo=-3, hi=4)
    y, *_ = general_gemm(W, X, out_dtype=torch.float32, layout="TN")
    bias = torch.arange(N, device="cuda", dtype=torch.float32) % 7
    yb, *_ = general_gemm(W, X, out_dtype=torch.bfloat16, layout="TN", bias=bias.to(torch.bfloat16))
    torch.cuda.synchronize()
    ref = torch.from_numpy(x).cuda().double() @ torch.from_numpy(w).cuda().double().T
    assert torch.equal(y.double(), ref)
    # bf16 output: |values| <= 9 * K + 6 are integers; exact in bf16 only below 256, so compare rounded
    assert torch.equal(yb.float(), (ref + bias.double()).float().to(torch.bfloat16).float())


def test_gemm_full_size_configs1_against_gpu_matmul():
    """BASELINE configs[1] sizes (the bench workload): fprop / dgrad / wgrad of the 4096 -> 14336
    Linear on 8192 tokens, checked against torch.matmul of the dequantized operands in fp32 on the
    GPU (independent code path; the CPU oracle would take minutes at this size).
    Tolerance: rtol 1.6e-2 (bf16 output rounding 2^-8 plus FP8-MFMA accumulation) and
    atol 1e-3 * max|ref| (measured MFMA accumulation error ~2e-5 * absmax per 256 of K)."""
    tok, hid, ffn = 8192, 4096, 14336
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn(tok, hid, device="cuda", dtype=torch.bfloat16, generator=g)
    w = (torch.randn(ffn, hid, device="cuda", dtype=torch.float32, generator=g) * 0.02).to(torch.bfloat16)
    dy = (torch.randn(tok, ffn, device="cuda", dtype=torch.float32, generator=g) * 0.01).to(torch.bfloat16)
    qx, qw, qg = (Float8CurrentScalingQuantizer(E4, device="cuda"), Float8CurrentScalingQuantizer(E4, device="cuda"),
                  Float8CurrentScalingQuantizer(E5, device="cuda"))
    x8, w8, dy8 = qx(x), qw(w), qg(dy)
    # dequantize with torch's own FP8 decode, not with this library
    deq = lambda t, td: t._data.view(td).to(torch.float32) * t._scale_inv
    xf, wf, dyf = deq(x8, torch.float8_e4m3fn), deq(w8, torch.float8_e4m3fn), deq(dy8, torch.float8_e5m2)
    old = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        for layout, a, b, ref in (("TN", w8, x8, xf @ wf.T), ("NN", w8, dy8, dyf @ wf), ("NT", x8, dy8, dyf.T @ xf)):
            out, *_ = general_gemm(a, b, out_dtype=torch.bfloat16, layout=layout)
            torch.cuda.synchronize()
            err = (out.float() - ref).abs()
            bound = 1.6e-2 * ref.abs() + 1e-3 * ref.abs().max()
            assert bool((err <= bound).all()), (layout, float(err.max()), float(ref.abs().max()))
            del out, ref, err, bound
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old


@pytest.mark.parametrize("swizzled", [False, True])
def test_mxfp8_gemm_dgrad_wgrad_layouts(swizzled):
    """MXFP8 NN (dgrad) and NT (wgrad): contraction over the stored rows, served by the
    transposed-columnwise extension.  Oracle: matmul of the dequantized COLUMNWISE-quantized
    operands (reference layout from the oracle), fp32 output tolerance rtol 1e-3 / atol 1e-3 absmax."""
    M, N, K = 384, 256, 512     # tokens, out features, in features
    g = np.random.default_rng(7)
    x = torch.from_numpy(g.standard_normal((M, K), dtype=np.float32)).cuda().to(torch.bfloat16)
    w = torch.from_numpy(g.standard_normal((N, K), dtype=np.float32) * 0.05).cuda().to(torch.bfloat16)
    dy = torch.from_numpy(g.standard_normal((M, N), dtype=np.float32) * 0.01).cuda().to(torch.bfloat16)
    def mk(dt):
        q = MXFP8Quantizer(dt, rowwise=True, columnwise=True, columnwise_transposed=True)
        q.optimize_for_gemm = swizzled
        return q
    x8, w8, dy8 = mk(E4)(x), mk(E4)(w), mk(E5)(dy)
    dx, *_ = general_gemm(w8, dy8, out_dtype=torch.float32, layout="NN", grad=True)
    dw, *_ = general_gemm(x8, dy8, out_dtype=torch.float32, layout="NT", grad=True)
    torch.cuda.synchronize()
    bits = lambda t: oracle.float_to_bf16_bits(t.float().cpu().numpy())
    rx, rw, rdy = oracle.mxfp8(bits(x), oracle.E4M3), oracle.mxfp8(bits(w), oracle.E4M3), oracle.mxfp8(bits(dy), oracle.E5M2)
    # dgrad: dX = dY (rowwise blocks along N) @ W (columnwise blocks along N)
    fdy_r = oracle.dequant_mxfp8(rdy["out_row"], oracle.E5M2, rdy["sc_row"], True)
    fw_c = oracle.dequant_mxfp8(rw["out_col"], oracle.E4M3, rw["sc_col"], False)
    ref_dx = fdy_r.astype(np.float64) @ fw_c.astype(np.float64)
    # wgrad: dW = dY^T (columnwise blocks along M) @ X (columnwise blocks along M)
    fdy_c = oracle.dequant_mxfp8(rdy["out_col"], oracle.E5M2, rdy["sc_col"], False)
    fx_c = oracle.dequant_mxfp8(rx["out_col"], oracle.E4M3, rx["sc_col"], False)
    ref_dw = fdy_c.astype(np.float64).T @ fx_c.astype(np.float64)
    for got, ref in ((dx, ref_dx), (dw, ref_dw)):
        got = got.cpu().numpy().astype(np.float64)
        assert got.shape == ref.shape
        assert np.all(np.abs(got - ref) <= 1e-3 * np.abs(ref) + 1e-3 * np.abs(ref).max())


@pytest.mark.parametrize("swizzled", [False, True])
@pytest.mark.parametrize("M,N,K", [(384, 256, 512), (256, 256, 128), (1024, 768, 1536), (2048, 4352, 640), (288, 416, 96)])
def test_mxfp8_gemm_reference_columnwise_layout_dgrad_wgrad(M, N, K, swizzled):
    """MXFP8 NN (dgrad) and NT (wgrad) on the REFERENCE's columnwise layout: the payload keeps the tensor's own
    [rows, cols] shape, quantized along the rows, scales [pad4(rows/32), pad128(cols)] (ref cublaslt_gemm.cu:179-193,
    :268-278 hands it to cuBLASLt as a non-TN operand).  The kernel reads it K-major through ds_read_b64_tr_b8.
      * oracle: matmul of the dequantized columnwise-quantized operands (oracle.mxfp8), fp32 output tolerance
        rtol 1e-3 / atol 1e-3 of absmax (FP8 MFMA internal precision);
      * bit-equal to the same GEMM fed the transposed-columnwise extension layout (same tiles, same K order).
    Shapes: one tile, several K tiles, multi-tile with ragged edges (288 x 416 x 96: partial tiles on every side, K tails)."""
    g = np.random.default_rng(7)
    x = torch.from_numpy(g.standard_normal((M, K), dtype=np.float32)).cuda().to(torch.bfloat16)
    w = torch.from_numpy(g.standard_normal((N, K), dtype=np.float32) * 0.05).cuda().to(torch.bfloat16)
    dy = torch.from_numpy(g.standard_normal((M, N), dtype=np.float32) * 0.01).cuda().to(torch.bfloat16)
    # per-row / per-column magnitudes differ, so a scale taken from the wrong row or block shows
    dy *= (1.0 + torch.arange(M, device="cuda").remainder(7).to(torch.bfloat16))[:, None]
    x *= (1.0 + torch.arange(K, device="cuda").remainder(5).to(torch.bfloat16))[None, :]

    def mk(dt, transposed):
        q = MXFP8Quantizer(dt, rowwise=True, columnwise=True, columnwise_transposed=transposed)
        q.optimize_for_gemm = swizzled
        return q
    whole = M % 128 == 0 and N % 128 == 0 and K % 128 == 0
    x8, w8, dy8 = mk(E4, False)(x), mk(E4, False)(w), mk(E5, False)(dy)
    assert tuple(w8._columnwise_data.shape) == (N, K) and not w8._columnwise_transposed      # the reference layout
    dx, *_ = general_gemm(w8, dy8, out_dtype=torch.float32, layout="NN", grad=True)
    dw, *_ = general_gemm(x8, dy8, out_dtype=torch.float32, layout="NT", grad=True)
    torch.cuda.synchronize()
    bits = lambda t: oracle.float_to_bf16_bits(t.float().cpu().numpy())
    rx, rw, rdy = oracle.mxfp8(bits(x), oracle.E4M3), oracle.mxfp8(bits(w), oracle.E4M3), oracle.mxfp8(bits(dy), oracle.E5M2)
    fdy_r = oracle.dequant_mxfp8(rdy["out_row"], oracle.E5M2, rdy["sc_row"], True)
    fw_c = oracle.dequant_mxfp8(rw["out_col"], oracle.E4M3, rw["sc_col"], False)
    ref_dx = fdy_r.astype(np.float64) @ fw_c.astype(np.float64)
    fdy_c = oracle.dequant_mxfp8(rdy["out_col"], oracle.E5M2, rdy["sc_col"], False)
    fx_c = oracle.dequant_mxfp8(rx["out_col"], oracle.E4M3, rx["sc_col"], False)
    ref_dw = fdy_c.astype(np.float64).T @ fx_c.astype(np.float64)
    for name, got, ref in (("dgrad", dx, ref_dx), ("wgrad", dw, ref_dw)):
        got = got.cpu().numpy().astype(np.float64)
        assert got.shape == ref.shape
        bad = np.abs(got - ref) > 1e-3 * np.abs(ref) + 1e-3 * np.abs(ref).max()
        assert not bad.any(), (name, int(bad.sum()), np.argwhere(bad)[:4].tolist())
    if whole:      # the transposed-columnwise extension needs whole 128 x 128 tiles
        x8t, w8t, dy8t = mk(E4, True)(x), mk(E4, True)(w), mk(E5, True)(dy)
        dxt, *_ = general_gemm(w8t, dy8t, out_dtype=torch.float32, layout="NN", grad=True)
        dwt, *_ = general_gemm(x8t, dy8t, out_dtype=torch.float32, layout="NT", grad=True)
        torch.cuda.synchronize()
        assert torch.equal(dx, dxt) and torch.equal(dw, dwt)


def test_mxfp8_gemm_reference_columnwise_layout_exact_integers():
    """Small-integer payloads with power-of-two block scales: every product and sum is exact, so a fragment byte
    taken from the wrong k or row of the K-major image shows as a mismatch.  Multi-tile (9 x 5 tiles), 5 K tiles."""
    M, N, K = 1152, 2304, 640
    g = np.random.default_rng(3)
    xv = g.integers(-3, 4, size=(M, K)).astype(np.float32)
    wv = g.integers(-3, 4, size=(N, K)).astype(np.float32)
    dyv = g.integers(-3, 4, size=(M, N)).astype(np.float32)
    # scale every 32-row block of the columnwise direction by a different power of two (exact in MXFP8)
    xv *= (2.0 ** (np.arange(M) // 32 % 3))[:, None]
    wv *= (2.0 ** (np.arange(N) // 32 % 4))[:, None]
    dyv *= (2.0 ** (np.arange(M) // 32 % 2))[:, None]
    q4, q5 = MXFP8Quantizer(E4), MXFP8Quantizer(E5)
    x8, w8, dy8 = (q(torch.from_numpy(v).cuda().to(torch.bfloat16)) for q, v in ((q4, xv), (q4, wv), (q5, dyv)))
    dx, *_ = general_gemm(w8, dy8, out_dtype=torch.float32, layout="NN", grad=True)
    dw, *_ = general_gemm(x8, dy8, out_dtype=torch.float32, layout="NT", grad=True)
    torch.cuda.synchronize()
    ref_dx = torch.from_numpy(dyv).cuda().double() @ torch.from_numpy(wv).cuda().double()
    ref_dw = torch.from_numpy(dyv).cuda().double().T @ torch.from_numpy(xv).cuda().double()
    assert torch.equal(dx.double(), ref_dx)
    assert torch.equal(dw.double(), ref_dw)


@pytest.mark.parametrize("tcol", [False, True])
def test_mxfp8_gemm_compact_scales_swizzled_first_equals_direct(tcol, monkeypatch):
    """general_gemm swizzles compact MXFP8 scales before a LARGE GEMM (the reference's wrapper always does, ref
    pytorch/csrc/extensions/gemm.cpp) and lets the kernel read them directly otherwise: the same scales either way, so
    the results are bit-equal - TN, NN, NT, both columnwise layouts, the reference one at sizes off the tile grid."""
    from transformer_engine_amd import gemm as G
    M, N, K = (384, 640, 512) if tcol else (416, 640, 352)    # (the transposed layout needs 128-multiples)
    q = MXFP8Quantizer(E4, columnwise_transposed=tcol)          # compact scales
    qg = MXFP8Quantizer(E5, columnwise_transposed=tcol)
    x, w, dy = q(randn_input((M, K), torch.bfloat16, 71)), q(randn_input((N, K), torch.bfloat16, 72, scale=0.05)), \
        qg(randn_input((M, N), torch.bfloat16, 73, scale=0.01))
    outs = {}
    for first in (0, 1 << 62):
        monkeypatch.setattr(G, "_SWIZZLE_FIRST_MACS", first)
        y, *_ = general_gemm(w, x, out_dtype=torch.float32, layout="TN")
        dx, *_ = general_gemm(w, dy, out_dtype=torch.float32, layout="NN", grad=True)
        dw, *_ = general_gemm(x, dy, out_dtype=torch.float32, layout="NT", grad=True)
        torch.cuda.synchronize()
        outs[first] = (y, dx, dw)
    for a, b in zip(outs[0], outs[1 << 62]):
        assert torch.equal(a, b) and not torch.isnan(a).any()
    assert not x._with_gemm_swizzled_scales          # the caller's tensors are left as they were


def test_mxfp8_gemm_k_major_argument_errors():
    """What the K-major path does not take is refused loudly, never computed wrongly."""
    from transformer_engine_amd._lib import NVTEError
    q, qt = MXFP8Quantizer(E4), MXFP8Quantizer(E4, columnwise_transposed=True)
    a = q(torch.randn(256, 256, device="cuda", dtype=torch.bfloat16))
    bt = qt(torch.randn(256, 256, device="cuda", dtype=torch.bfloat16))
    with pytest.raises(NVTEError, match="K-major B operand"):
        general_gemm(bt, a, out_dtype=torch.float32, layout="NT")     # A transposed layout, B reference layout
    with pytest.raises(NVTEError, match="epilogues read K-contiguous"):
        general_gemm(a, a, out_dtype=torch.float32, layout="NN", gelu=True)


@pytest.mark.parametrize("M,N,K", [
    (1024, 1024, 2048),    # 16 tiles on 256 CUs
    (8192, 9216, 1152),    # 1152 tiles = 4 full rounds + 128, odd K-tile count (9)
    (1000, 1100, 1040),    # ragged tiles, K tail inside the last K tile
])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_gemm_exact_partial_rounds(M, N, K, out_dtype):
    """Tile counts that leave the last round of the persistent walk partly empty (16 tiles, 4.5
    rounds, ragged edges), long K.  Small-integer data: exact.  Run twice (state-free launches)."""
    W, w = _int_fp8_tensor((N, K), E4, 21, lo=-2, hi=3)
    X, x = _int_fp8_tensor((M, K), E4, 22, lo=-2, hi=3)
    ref = torch.from_numpy(x).cuda().double() @ torch.from_numpy(w).cuda().double().T
    for _ in range(2):
        y, *_ = general_gemm(W, X, out_dtype=out_dtype, layout="TN")
        torch.cuda.synchronize()
        assert torch.equal(y.double(), ref.float().to(out_dtype).double())
    # with accumulate (beta = 1) and bias through the same path
    acc = torch.ones(M, N, device="cuda", dtype=torch.float32)
    general_gemm(W, X, layout="TN", out=acc, accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(acc.double(), ref + 1.0)


# ---------------------------------------------------------------------------------------------
# GEMM epilogues (ref common/gemm/cublaslt_gemm.cu:619-689): GELU_AUX[_BIAS], DGELU[_BGRAD], BGRADB,
# FP8 output with D scale / amax.  Integer-valued operands make the pre-activation exact, so aux
# and the linear parts are compared bit-for-bit; the activations within the tanhf tolerance written
# at each assert (device tanhf vs glibc tanhf, both fp32).
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 200, 160), (512, 768, 384)])
@pytest.mark.parametrize("with_bias", [False, True])
def test_gemm_gelu_aux_epilogue(M, N, K, with_bias):
    W, w = _int_fp8_tensor((N, K), E4, 31, scale_inv=0.125, lo=-2, hi=3)
    X, x = _int_fp8_tensor((M, K), E4, 32, scale_inv=0.25, lo=-2, hi=3)
    bias = (torch.arange(N, dtype=torch.float32, device="cuda") % 7 - 3) * 0.5 if with_bias else None
    y, _, aux, _ = general_gemm(W, X, out_dtype=torch.float32, layout="TN", gelu=True, bias=bias)
    torch.cuda.synchronize()
    pre = (x @ w.T * 0.03125).astype(np.float32)
    if with_bias:
        pre = pre + bias.cpu().numpy()[None, :]
        assert aux.dtype == torch.float32          # aux dtype = bias dtype (ref csrc/extensions/gemm.cpp:256)
        assert np.array_equal(aux.cpu().numpy(), pre)
    else:
        assert aux.dtype == torch.bfloat16
        assert np.array_equal(aux.view(torch.int16).cpu().numpy().view(np.uint16), oracle.float_to_bf16_bits(pre))
    # tolerance: 4 ulp-ish of fp32 around |y| <= ~50
    assert np.allclose(y.cpu().numpy(), oracle.gelu(pre), rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 208, 170)])
@pytest.mark.parametrize("bgrad_dtype", [None, torch.float32, torch.bfloat16])
def test_gemm_dgelu_epilogue(M, N, K, bgrad_dtype):
    # dgrad-shaped: D [M, K'] = dY [M, N] W [N, K'] ; here NN layout with A = W [N, Kp], B = dY [M, N]
    Kp = K
    W, w = _int_fp8_tensor((N, Kp), E4, 33, lo=-2, hi=3)
    dY, dy = _int_fp8_tensor((M, N), E5, 34, lo=-2, hi=3, scale_inv=0.5)
    g = np.random.default_rng(35)
    aux_np = g.standard_normal((M, Kp)).astype(np.float32) * 2
    aux = torch.from_numpy(aux_np).cuda()
    bias = torch.empty(Kp, dtype=bgrad_dtype, device="cuda") if bgrad_dtype is not None else None
    out_dtype = bgrad_dtype or torch.float32
    d, bgrad, aux_back, _ = general_gemm(W, dY, out_dtype=out_dtype, layout="NN", gelu=True, gelu_in=aux,
                                         grad=True, bias=bias)
    torch.cuda.synchronize()
    lin = (dy @ w * 0.5).astype(np.float32)
    ref = lin * oracle.dgelu(aux_np)
    got = d.float().cpu().numpy()
    if out_dtype == torch.float32:
        assert np.allclose(got, ref, rtol=2e-6, atol=2e-6 * np.abs(lin).max())
    else:  # bf16 output: one bf16 rounding on top
        assert np.allclose(got, ref, rtol=2 ** -8, atol=1e-5 * np.abs(lin).max())
    if bgrad_dtype is None:
        assert bgrad is None
    else:
        assert bgrad.shape == (Kp,) and bgrad.dtype == bgrad_dtype
        col = ref.astype(np.float64).sum(axis=0)
        scale = np.abs(ref).sum(axis=0).max()
        # fp32 atomics in arbitrary order (+ one bf16 rounding of the result)
        tol = 1e-5 if bgrad_dtype == torch.float32 else 2 ** -7
        assert np.allclose(bgrad.float().cpu().numpy(), col, rtol=tol, atol=tol * scale)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1008, 200, 330)])
def test_gemm_bgradb_epilogue(M, N, K):
    # wgrad NT: A = X [M, K], B = dY [M, N] -> dW [N, K]; bias gradient = sum over tokens of dY
    X, x = _int_fp8_tensor((M, K), E4, 36)
    dY, dy = _int_fp8_tensor((M, N), E5, 37, scale_inv=0.5)
    dw, bgrad, _, _ = general_gemm(X, dY, out_dtype=torch.float32, layout="NT", grad=True,
                                   bias=torch.empty(N, dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    assert np.array_equal(dw.cpu().numpy(), (dy.T @ x * 0.5).astype(np.float32))
    assert bgrad.shape == (N,)
    assert np.array_equal(bgrad.cpu().numpy(), (dy.sum(axis=0) * 0.5).astype(np.float32))   # exact: small integers


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 200, 160)])
@pytest.mark.parametrize("out_fmt", [E4, E5])
def test_gemm_fp8_output_with_scale_and_amax(M, N, K, out_fmt):
    from transformer_engine_amd.tensor import Float8Quantizer
    W, w = _int_fp8_tensor((N, K), E4, 38)
    X, x = _int_fp8_tensor((M, K), E4, 39)
    scale = torch.full((1,), 0.375, device="cuda")
    amax = torch.full((1,), 123.0, device="cuda")       # stale value must be overwritten
    q = Float8Quantizer(scale, amax, out_fmt)
    bias = torch.arange(N, dtype=torch.bfloat16, device="cuda") % 5
    y, *_ = general_gemm(W, X, quantization_params=q, layout="TN", bias=bias)
    torch.cuda.synchronize()
    ref = (x @ w.T).astype(np.float32) + bias.float().cpu().numpy()[None, :]
    assert isinstance(y, Float8Tensor)
    assert np.array_equal(u8(y._data), oracle.float_to_fp8(ref * np.float32(0.375), FMT[out_fmt]))
    assert float(amax.item()) == float(np.abs(ref).max())
    assert float(y._scale_inv.item()) == float(np.float32(1.0) / np.float32(0.375))
    # columnwise copy = byte transpose of the rowwise payload
    assert np.array_equal(u8(y._transpose), u8(y._data).T)


def test_gemm_unfused_output_quantization_current_scaling():
    W, w = _int_fp8_tensor((256, 128), E4, 40)
    X, x = _int_fp8_tensor((256, 128), E4, 41)
    q = Float8CurrentScalingQuantizer(E4, torch.device("cuda"))
    y, *_ = general_gemm(W, X, quantization_params=q, layout="TN", out_dtype=torch.float32)
    torch.cuda.synchronize()
    ref = (x @ w.T).astype(np.float32)
    sc = oracle.scale_from_amax(float(np.abs(ref).max()), oracle.E4M3)
    assert np.array_equal(u8(y._data), oracle.float_to_fp8(ref * sc, oracle.E4M3))


def test_mxfp8_gemm_gelu_epilogue():
    M, N, K = 256, 384, 256
    x = randn_input((M, K), torch.bfloat16, 42)
    w = randn_input((N, K), torch.bfloat16, 43, scale=0.1)
    q = MXFP8Quantizer(E4, columnwise=False)
    qx, qw = q(x), q(w)
    y, _, aux, _ = general_gemm(qw, qx, out_dtype=torch.float32, layout="TN", gelu=True)
    y0, *_ = general_gemm(qw, qx, out_dtype=torch.float32, layout="TN")
    torch.cuda.synchronize()
    pre = y0.cpu().numpy()
    # the plain kernel is the oracle-checked one (test_mxfp8_gemm_TN); the epilogue variant must agree with it
    assert np.array_equal(aux.view(torch.int16).cpu().numpy().view(np.uint16), oracle.float_to_bf16_bits(pre))
    assert np.allclose(y.cpu().numpy(), oracle.gelu(pre), rtol=2e-6, atol=2e-6)


def test_gemm_epilogue_argument_errors():
    from transformer_engine_amd import NVTEError
    from transformer_engine_amd.tensor import Float8Quantizer
    W, _ = _int_fp8_tensor((64, 32), E4, 1)
    X, _ = _int_fp8_tensor((64, 32), E4, 2)
    with pytest.raises(ValueError, match="aux tensor"):
        general_gemm(W, X, out_dtype=torch.float32, layout="TN", gelu=True, grad=True)
    q = Float8Quantizer(torch.ones(1, device="cuda"), torch.zeros(1, device="cuda"), E4)
    with pytest.raises(ValueError, match="Accumulation mode not supported for FP8 output"):
        general_gemm(W, X, quantization_params=q, layout="TN", accumulate=True)
    with pytest.raises(ValueError, match="non-zero beta"):
        general_gemm(W, X, out_dtype=torch.float32, layout="TN", beta=0.5)
