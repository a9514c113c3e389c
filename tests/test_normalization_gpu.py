"""GPU parity of LayerNorm / RMSNorm forward (with fused FP8 output) and backward.

Oracle: restatement of the reference's own host oracle (tests/cpp/operator/test_normalization.h),
shapes and tolerances from tests/cpp/operator/test_normalization.cu:205-237: mu / rsigma rtol 5e-5,
output per getTolerances(otype) (fp32 1e-6/8e-6... bf16 1e-5/8e-3, fp8 1e-2-ish); FP8 bytes are compared
after dequantisation allowing ONE code step where the fp32 pre-quantisation values differ in the last
bits (the sums are taken in a different order than the serial reference loop); backward atol/rtol 5e-4.
"""
import numpy as np
import pytest
import torch

import oracle
from helpers import FMT, randn_input, to_oracle_input, u8
from transformer_engine_amd.normalization import layernorm_bwd, layernorm_fwd, rmsnorm_bwd, rmsnorm_fwd
from transformer_engine_amd.tensor import DType, Float8CurrentScalingQuantizer, Float8Quantizer, MXFP8Quantizer

pytestmark = pytest.mark.gpu
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2

SHAPES = [(71, 229), (29, 541), (768, 6144), (2048, 12288), (256, 4096), (64, 8192)]   # first four: the reference's
# rows that do not fill the vector kernels' (vectors per thread) x (threads) footprint: the guarded instantiations
# of every workgroup width (1x256, 2x256, 3x256, 2x512, 3x512, 2x1024 threads x vectors), plus 16384 (full, 1024 threads)
GUARDED = [(17, 1024), (33, 3072), (21, 5120), (19, 7168), (40, 10240), (13, 14336), (24, 16384)]


def _inputs(N, H, dtype, seed):
    x = randn_input((N, H), dtype, seed) * 1.5 + 0.25
    g = randn_input((H,), dtype, seed + 1) * 0.5 + 1.0
    b = randn_input((H,), dtype, seed + 2) * 0.25
    return x.contiguous(), g.contiguous(), b.contiguous()


@pytest.mark.parametrize("N,H", SHAPES + GUARDED)
@pytest.mark.parametrize("kind", ["ln", "rms"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("zero_centered", [False, True])
def test_norm_fwd_high_precision(N, H, kind, dtype, zero_centered):
    if dtype == torch.float32 and N * H > 8e6:
        pytest.skip("large fp32 case adds nothing over bf16")
    x, g, b = _inputs(N, H, dtype, 3)
    if kind == "ln":
        out, mu, rs = layernorm_fwd(x, g, b, 1e-5, zero_centered_gamma=zero_centered)
    else:
        out, mu, rs = rmsnorm_fwd(x, g, 1e-5, zero_centered_gamma=zero_centered)
    torch.cuda.synchronize()
    ref, rmu, rrs, _ = oracle.norm_forward(to_oracle_input(x), to_oracle_input(g),
                                           to_oracle_input(b) if kind == "ln" else None, 1e-5,
                                           zero_centered_gamma=zero_centered)
    assert np.allclose(rs.cpu().numpy(), rrs, rtol=5e-5, atol=1e-6)
    if kind == "ln":
        assert np.allclose(mu.cpu().numpy(), rmu, rtol=5e-5, atol=1e-6)
    if dtype == torch.float32:
        assert np.allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    else:   # bf16: one rounding of the output (getTolerances(bf16): atol 1e-5, rtol 8e-3)
        assert np.allclose(out.float().cpu().numpy(), ref, rtol=8e-3, atol=1e-5)


@pytest.mark.parametrize("N,H", [(71, 229), (768, 6144), (256, 4096), (2048, 12288)] + GUARDED)
@pytest.mark.parametrize("kind", ["ln", "rms"])
@pytest.mark.parametrize("fmt", [E4, E5])
def test_norm_fwd_fp8_delayed_scaling(N, H, kind, fmt):
    x, g, b = _inputs(N, H, torch.bfloat16, 5)
    scale = torch.full((1,), 13.5, device="cuda")
    amax = torch.zeros(1, device="cuda")
    q = Float8Quantizer(scale, amax, fmt)
    if kind == "ln":
        out, mu, rs = layernorm_fwd(x, g, b, 1e-5, quantizer=q)
    else:
        out, mu, rs = rmsnorm_fwd(x, g, 1e-5, quantizer=q)
    torch.cuda.synchronize()
    ref, _, rrs, ref_amax = oracle.norm_forward(to_oracle_input(x), to_oracle_input(g),
                                                to_oracle_input(b) if kind == "ln" else None, 1e-5, scale=13.5)
    assert np.allclose(rs.cpu().numpy(), rrs, rtol=5e-5, atol=1e-6)
    assert np.isclose(amax.item(), ref_amax, rtol=1e-5)                      # fp32 tolerance, as the reference
    assert out._scale_inv.item() == np.float32(1.0) / np.float32(13.5)
    got = oracle.fp8_to_float(u8(out._data), FMT[fmt])
    want = oracle.fp8_to_float(oracle.float_to_fp8(ref, FMT[fmt]), FMT[fmt])
    # identical codes except where the fp32 value sits on a rounding boundary: at most one code step
    # (relative spacing 2^-3 for e4m3, 2^-2 for e5m2), on a tiny fraction of the elements
    step = 2.0 ** -3 if fmt == E4 else 2.0 ** -2
    sub = 2.0 ** -9 if fmt == E4 else 2.0 ** -16          # spacing of the subnormal codes
    diff = np.abs(got - want)
    # ... or inside the reference's own FP8 output tolerance (getTolerances: atol 1e-2, rtol 1e-2), which
    # covers values that cancel to ~0 in (x - mu)
    assert (diff <= np.maximum(step * np.abs(want) + sub, 1e-2 + 1e-2 * np.abs(want))).all()
    assert (diff != 0).mean() < 2e-3
    # the columnwise copy is the byte transpose of the rowwise payload
    assert np.array_equal(u8(out._transpose), u8(out._data).T)


@pytest.mark.parametrize("fmt", [E4, E5])
@pytest.mark.parametrize("H", [4096, 3072])
def test_norm_fwd_fp8_saturation_and_nan(fmt, H):
    """A scale that pushes part of the output past the format's finite range, an infinite gamma and a NaN
    gamma: out-of-range values saturate to +-max finite, NaN stays NaN, every other byte is what the
    per-value saturating cast of the oracle gives (the vector kernel decides per 8 values whether any clamp
    is needed: both of its branches are taken here)."""
    N = 48
    x, g, b = _inputs(N, H, torch.bfloat16, 31)
    g[5] = float("inf")
    g[H - 3] = float("nan")
    big = 300.0 if fmt == E4 else 40000.0     # |y| reaches ~6: y * scale crosses 448 / 57344 on some vectors only
    scale = torch.full((1,), big, device="cuda")
    amax = torch.zeros(1, device="cuda")
    q = Float8Quantizer(scale, amax, fmt, columnwise=False)
    out, _, _ = rmsnorm_fwd(x, g, 1e-5, quantizer=q)
    torch.cuda.synchronize()
    ref, _, _, _ = oracle.norm_forward(to_oracle_input(x), to_oracle_input(g), None, 1e-5, scale=big)
    got_b = u8(out._data)
    want_b = oracle.float_to_fp8(ref, FMT[fmt])
    got = oracle.fp8_to_float(got_b, FMT[fmt])
    want = oracle.fp8_to_float(want_b, FMT[fmt])
    fmax = 448.0 if fmt == E4 else 57344.0
    assert np.isnan(got[:, H - 3]).all() and np.isnan(want[:, H - 3]).all()
    x5 = x[:, 5].float().cpu().numpy()
    assert (np.abs(got[:, 5][x5 != 0]) == fmax).all()                 # inf saturates, sign kept
    assert (np.sign(got[:, 5][x5 != 0]) == np.sign(x5[x5 != 0])).all()
    sat = np.abs(want) == fmax
    assert sat.mean() > 0.01 and (~sat).mean() > 0.5                   # both branches exercised
    ok = ~np.isnan(want)
    step = 2.0 ** -3 if fmt == E4 else 2.0 ** -2
    assert (np.abs(got[ok] - want[ok]) <= step * np.abs(want[ok]) + 2.0 ** -9).all()
    assert (got_b[ok] != want_b[ok]).mean() < 2e-3


def test_norm_fwd_mxfp8_and_current_scaling_outputs():
    N, H = 256, 1024
    x, g, b = _inputs(N, H, torch.bfloat16, 7)
    hp, _, _ = rmsnorm_fwd(x, g, 1e-5)
    # MXFP8: the normalized rows (in the input dtype) quantized by the MXFP8 kernel
    qm = MXFP8Quantizer(E4)
    out, _, _ = rmsnorm_fwd(x, g, 1e-5, quantizer=qm)
    ref = qm(hp)
    torch.cuda.synchronize()
    assert torch.equal(out._rowwise_data, ref._rowwise_data) and torch.equal(out._rowwise_scale_inv, ref._rowwise_scale_inv)
    assert torch.equal(out._columnwise_data, ref._columnwise_data)
    # current scaling: norm to high precision, then the amax -> scale -> cast sequence
    qc = Float8CurrentScalingQuantizer(E4, torch.device("cuda"))
    out, _, _ = rmsnorm_fwd(x, g, 1e-5, quantizer=qc)
    ref = qc(hp)
    torch.cuda.synchronize()
    assert torch.equal(out._data, ref._data) and torch.equal(out._scale_inv, ref._scale_inv)


@pytest.mark.parametrize("N,H", [(71, 229), (29, 541), (768, 6144), (512, 8192), (2048, 12288), (300, 4096)] + GUARDED)
@pytest.mark.parametrize("kind", ["ln", "rms", "rms_add"])
@pytest.mark.parametrize("zero_centered", [False, True])
def test_norm_bwd(N, H, kind, zero_centered):
    dtype = torch.bfloat16 if H != 541 else torch.float32
    x, g, b = _inputs(N, H, dtype, 9)
    dz = randn_input((N, H), dtype, 12)
    add = randn_input((N, H), dtype, 13) if kind == "rms_add" else None
    if kind == "ln":
        _, mu, rs = layernorm_fwd(x, g, b, 1e-5, zero_centered_gamma=zero_centered)
        dx, dg, db = layernorm_bwd(dz, x, mu, rs, g, zero_centered_gamma=zero_centered)
    else:
        _, mu, rs = rmsnorm_fwd(x, g, 1e-5, zero_centered_gamma=zero_centered)
        dx, dg = rmsnorm_bwd(dz, x, rs, g, zero_centered_gamma=zero_centered, add=add)
        db = None
    torch.cuda.synchronize()
    rdx, rdg, rdb = oracle.norm_backward(to_oracle_input(dz), to_oracle_input(x),
                                         mu.cpu().numpy() if kind == "ln" else None, rs.cpu().numpy(),
                                         to_oracle_input(g), add=to_oracle_input(add) if add is not None else None,
                                         zero_centered_gamma=zero_centered)
    # reference test tolerance 5e-4 abs/rel on values of its dtype; bf16 results carry one bf16 rounding
    # (2^-8 relative), dgamma/dbeta sums over N rows are compared relative to the column's absolute sum
    tol = 5e-4 if dtype == torch.float32 else 2 ** -7
    assert np.allclose(dx.float().cpu().numpy(), rdx, rtol=tol, atol=tol)
    scale_g = np.abs(rdg).max() + 1.0
    assert np.allclose(dg.float().cpu().numpy(), rdg, rtol=tol, atol=tol * scale_g)
    if db is not None:
        assert np.allclose(db.float().cpu().numpy(), rdb, rtol=tol, atol=tol * (np.abs(rdb).max() + 1.0))


def test_norm_argument_errors():
    from transformer_engine_amd import NVTEError
    x, g, b = _inputs(16, 64, torch.bfloat16, 1)
    with pytest.raises(NVTEError, match="Gamma must have the same hidden size"):
        rmsnorm_fwd(x, g[:32].contiguous(), 1e-5)
    with pytest.raises(NVTEError, match="Epsilon must be non-negative"):
        rmsnorm_fwd(x, g, -1.0)


def test_norm_fwd_mixed_dtypes_take_the_general_kernel():
    """bf16 input with fp32 output / fp32 gamma: not a vector-kernel combination, same numbers expected."""
    N, H = 96, 1024
    x, g, b = _inputs(N, H, torch.bfloat16, 21)
    out32, _, rs = rmsnorm_fwd(x, g, 1e-5, out_dtype=torch.float32)
    out_g32, mu, _ = layernorm_fwd(x, g.float(), b.float(), 1e-5)
    torch.cuda.synchronize()
    ref, _, rrs, _ = oracle.norm_forward(to_oracle_input(x), to_oracle_input(g), None, 1e-5)
    assert out32.dtype == torch.float32
    assert np.allclose(out32.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    assert np.allclose(rs.cpu().numpy(), rrs, rtol=5e-5, atol=1e-6)
    ref_ln, _, _, _ = oracle.norm_forward(to_oracle_input(x), to_oracle_input(g.float()), to_oracle_input(b.float()), 1e-5)
    assert np.allclose(out_g32.float().cpu().numpy(), ref_ln, rtol=8e-3, atol=1e-5)


@pytest.mark.parametrize("kind", ["ln", "rms"])
@pytest.mark.parametrize("N,H", [(71, 229), (256, 4096)])
def test_zero_centered_gamma_in_weight_dtype_switch(N, H, kind):
    """ref tests/cpp/operator/test_normalization.h:63-93 (compute_gamma): with the cuDNN backend AND
    nvte_enable_zero_centered_gamma_in_weight_dtype(true) the effective gamma is (gamma + 1) rounded to
    the weight dtype; with either switch off it is gamma + 1 in fp32.  The oracle is fed the pre-rounded
    gamma.  fp32 output so the difference between the two roundings is far above the tolerance."""
    from transformer_engine_amd._lib import api
    x = randn_input((N, H), torch.bfloat16, 31)
    g = (randn_input((H,), torch.bfloat16, 32) * 0.003).contiguous()      # 1 + g is NOT a bf16 number
    b = (randn_input((H,), torch.bfloat16, 33) * 0.25).contiguous()
    dz = randn_input((N, H), torch.bfloat16, 34)
    g_rounded = (g.float() + 1.0).to(torch.bfloat16)

    def run():
        if kind == "ln":
            out, mu, rs = layernorm_fwd(x, g, b, 1e-5, zero_centered_gamma=True, out_dtype=torch.float32)
            dx, dg, db = layernorm_bwd(dz, x, mu, rs, g, zero_centered_gamma=True)
        else:
            out, mu, rs = rmsnorm_fwd(x, g, 1e-5, zero_centered_gamma=True, out_dtype=torch.float32)
            dx, dg = rmsnorm_bwd(dz, x, rs, g, zero_centered_gamma=True)
        torch.cuda.synchronize()
        return out.float().cpu().numpy(), dx.float().cpu().numpy(), mu, rs

    try:
        api.nvte_enable_zero_centered_gamma_in_weight_dtype(True)
        plain_out, plain_dx, _, _ = run()                       # cuDNN switches off: fp32 gamma + 1
        api.nvte_enable_cudnn_norm_fwd(True)
        api.nvte_enable_cudnn_norm_bwd(True)
        w_out, w_dx, mu, rs = run()
    finally:
        api.nvte_enable_zero_centered_gamma_in_weight_dtype(False)
        api.nvte_enable_cudnn_norm_fwd(False)
        api.nvte_enable_cudnn_norm_bwd(False)
    beta = to_oracle_input(b) if kind == "ln" else None
    ref_w, _, _, _ = oracle.norm_forward(to_oracle_input(x), to_oracle_input(g_rounded), beta, 1e-5)
    ref_p, _, _, _ = oracle.norm_forward(to_oracle_input(x), to_oracle_input(g), beta, 1e-5, zero_centered_gamma=True)
    assert np.allclose(w_out, ref_w, rtol=1e-5, atol=1e-5)
    assert np.allclose(plain_out, ref_p, rtol=1e-5, atol=1e-5)
    assert not np.allclose(w_out, plain_out, rtol=1e-4, atol=1e-4)          # the switch changes the numbers
    mu_np = mu.cpu().numpy() if kind == "ln" else None
    rdx_w, _, _ = oracle.norm_backward(to_oracle_input(dz), to_oracle_input(x), mu_np, rs.cpu().numpy(),
                                       to_oracle_input(g_rounded))
    assert np.allclose(w_dx, rdx_w, rtol=8e-3, atol=5e-4)


@pytest.mark.parametrize("N", [1, 3, 257])
@pytest.mark.parametrize("H", [4096, 12288, 1000])
def test_norm_few_rows(N, H):
    """Fewer rows than persistent workgroups (most workgroups exit at once, the others never prefetch a second
    row), and a row count that leaves the last round of workgroups ragged; vector (full / guarded) and scalar paths."""
    x, g, b = _inputs(N, H, torch.bfloat16, 41)
    dz = randn_input((N, H), torch.bfloat16, 42)
    out, mu, rs = layernorm_fwd(x, g, b, 1e-5)
    dx, dg, db = layernorm_bwd(dz, x, mu, rs, g)
    q = Float8Quantizer(torch.full((1,), 4.0, device="cuda"), torch.zeros(1, device="cuda"), E4, columnwise=False)
    out8, _, rs8 = rmsnorm_fwd(x, g, 1e-5, quantizer=q)
    torch.cuda.synchronize()
    ref, rmu, rrs, _ = oracle.norm_forward(to_oracle_input(x), to_oracle_input(g), to_oracle_input(b), 1e-5)
    assert np.allclose(out.float().cpu().numpy(), ref, rtol=8e-3, atol=1e-5)
    assert np.allclose(mu.cpu().numpy(), rmu, rtol=5e-5, atol=1e-6) and np.allclose(rs.cpu().numpy(), rrs, rtol=5e-5, atol=1e-6)
    rdx, rdg, rdb = oracle.norm_backward(to_oracle_input(dz), to_oracle_input(x), mu.cpu().numpy(), rs.cpu().numpy(),
                                         to_oracle_input(g))
    tol = 2 ** -7
    assert np.allclose(dx.float().cpu().numpy(), rdx, rtol=tol, atol=tol)
    assert np.allclose(dg.float().cpu().numpy(), rdg, rtol=tol, atol=tol * (np.abs(rdg).max() + 1.0))
    assert np.allclose(db.float().cpu().numpy(), rdb, rtol=tol, atol=tol * (np.abs(rdb).max() + 1.0))
    ref8, _, _, ramax = oracle.norm_forward(to_oracle_input(x), to_oracle_input(g), None, 1e-5, scale=4.0)
    got = oracle.fp8_to_float(u8(out8._data), oracle.E4M3)
    want = oracle.fp8_to_float(oracle.float_to_fp8(ref8, oracle.E4M3), oracle.E4M3)
    assert (np.abs(got - want) <= np.maximum(2.0 ** -3 * np.abs(want) + 2.0 ** -9, 1e-2 + 1e-2 * np.abs(want))).all()
    assert np.isclose(q.amax.item(), ramax, rtol=1e-5)
