"""FP8 Linear forward + backward on the GPU for the three recipes (delayed, current, MXFP8):
the host-side mirror of the reference's module/linear.py driving quantize -> GEMM through the C ABI.

Oracle: the same quantize -> dequantize -> matmul chain on the CPU (oracle/), i.e. what the
reference's own tests assert for quantized GEMMs.  Tolerances written at the asserts:
bf16 outputs rtol 1.6e-2 (ref tests/pytorch/utils.py:97-98) + atol 1e-3 * absmax (FP8 MFMA
accumulation)."""
import numpy as np
import pytest
import torch

import oracle
from transformer_engine_amd import linear as L
from transformer_engine_amd.recipe import DelayedScaling, Float8CurrentScaling, MXFP8BlockScaling

pytestmark = pytest.mark.gpu


def _bits(t):
    return oracle.float_to_bf16_bits(t.float().cpu().numpy())


def _close(got, ref):
    got = got.float().cpu().numpy().astype(np.float64)
    return np.all(np.abs(got - ref) <= 1.6e-2 * np.abs(ref) + 1e-3 * np.abs(ref).max())


@pytest.mark.parametrize("use_bias", [False, True])
def test_linear_mxfp8_fwd_bwd(use_bias):
    M, K, N = 256, 384, 512
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    w = (torch.randn(N, K, device="cuda", dtype=torch.float32, generator=g) * 0.05).to(torch.bfloat16)
    dy = (torch.randn(M, N, device="cuda", dtype=torch.float32, generator=g) * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", dtype=torch.bfloat16, generator=g) if use_bias else None
    st = L.LinearFP8State(MXFP8BlockScaling(), device=torch.device("cuda"))
    y, saved = L.linear_forward(x, w, bias, st)
    dx, dw, dbias = L.linear_backward(dy, saved, st, use_bias=use_bias)
    torch.cuda.synchronize()
    # MXFP8BlockScaling defaults to E4M3 for forward AND backward tensors (ref recipe/__init__.py)
    GF = oracle.E4M3
    rx, rw, rdy = oracle.mxfp8(_bits(x), oracle.E4M3), oracle.mxfp8(_bits(w), oracle.E4M3), oracle.mxfp8(_bits(dy), GF)
    d = oracle.dequant_mxfp8
    fx_r, fw_r = d(rx["out_row"], oracle.E4M3, rx["sc_row"], True), d(rw["out_row"], oracle.E4M3, rw["sc_row"], True)
    fw_c, fx_c = d(rw["out_col"], oracle.E4M3, rw["sc_col"], False), d(rx["out_col"], oracle.E4M3, rx["sc_col"], False)
    fdy_r, fdy_c = d(rdy["out_row"], GF, rdy["sc_row"], True), d(rdy["out_col"], GF, rdy["sc_col"], False)
    ref_y = fx_r.astype(np.float64) @ fw_r.astype(np.float64).T
    if use_bias:
        ref_y = ref_y + bias.float().cpu().numpy().astype(np.float64)
    assert _close(y, ref_y)
    assert _close(dx, fdy_r.astype(np.float64) @ fw_c.astype(np.float64))
    assert _close(dw, fdy_c.astype(np.float64).T @ fx_c.astype(np.float64))
    if use_bias:
        ref_db = dy.float().cpu().numpy().astype(np.float64).sum(0)
        got = dbias.float().cpu().numpy().astype(np.float64)
        assert np.all(np.abs(got - ref_db) <= 1e-2 * np.abs(ref_db) + 1e-2)   # bf16 output of an fp32 sum


@pytest.mark.parametrize("recipe", [DelayedScaling(), Float8CurrentScaling()])
def test_linear_per_tensor_fwd_bwd(recipe):
    M, K, N = 512, 256, 384
    g = torch.Generator(device="cuda").manual_seed(6)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    w = (torch.randn(N, K, device="cuda", dtype=torch.float32, generator=g) * 0.05).to(torch.bfloat16)
    dy = (torch.randn(M, N, device="cuda", dtype=torch.float32, generator=g) * 0.02).to(torch.bfloat16)
    st = L.LinearFP8State(recipe, device=torch.device("cuda"))
    # delayed scaling: two steps so that the scales come from the recorded amax
    for _ in range(2):
        y, saved = L.linear_forward(x, w, None, st)
        st.update_fwd()
        dx, dw, _ = L.linear_backward(dy, saved, st)
        st.update_bwd()
    torch.cuda.synchronize()
    x8, w8 = saved
    # dequantize what the GEMMs actually consumed (scale_inv as stored beside the bytes)
    fx = oracle.dequant(x8._data.cpu().numpy(), oracle.E4M3, float(x8._scale_inv.item()))
    fw = oracle.dequant(w8._data.cpu().numpy(), oracle.E4M3, float(w8._scale_inv.item()))
    assert _close(y, fx.astype(np.float64) @ fw.astype(np.float64).T)
    # backward operands: re-quantize dY on the CPU with the scale the step used
    ref_dx = dy.float().cpu().numpy().astype(np.float64) @ fw.astype(np.float64)
    ref_dw = dy.float().cpu().numpy().astype(np.float64).T @ fx.astype(np.float64)
    # dY is E5M2 (2 mantissa bits): elementwise quantization error up to 2^-3 relative, averaged
    # over the contraction; compare with the reference's module tolerance rtol 0.125 / atol 0.0675
    # (tests/pytorch/utils.py:103-104) scaled to the output magnitude
    for got, ref in ((dx, ref_dx), (dw, ref_dw)):
        got = got.float().cpu().numpy().astype(np.float64)
        assert np.all(np.abs(got - ref) <= 0.125 * np.abs(ref) + 0.0675 * np.abs(ref).max())


@pytest.mark.parametrize("normalization", ["LayerNorm", "RMSNorm"])
@pytest.mark.parametrize("recipe_name", ["delayed", "current", "mxfp8"])
def test_layernorm_linear_fwd_bwd_against_fp32_composition(normalization, recipe_name):
    """LayerNormLinear (norm fused with the FP8 quantize of the GEMM input) against the plain-PyTorch
    fp32 composition norm -> matmul, with the reference's FP8 module tolerances
    (tests/pytorch/utils.py:103-106: rtol 0.125, atol 0.0675 x the tensor's absmax)."""
    from transformer_engine_amd.linear import (LinearFP8State, layernorm_linear_backward,
                                               layernorm_linear_forward)
    from transformer_engine_amd.recipe import DelayedScaling, Float8CurrentScaling, MXFP8BlockScaling
    recipe = {"delayed": DelayedScaling(), "current": Float8CurrentScaling(), "mxfp8": MXFP8BlockScaling()}[recipe_name]
    M, K, N = 512, 1024, 768
    torch.manual_seed(7)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    g = (torch.randn(K, device="cuda") * 0.2 + 1.0).to(torch.bfloat16)
    b = (torch.randn(K, device="cuda") * 0.1).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    st = LinearFP8State(recipe)
    steps = 3 if recipe_name == "delayed" else 1      # delayed scaling: let the scales settle
    for _ in range(steps):
        y, saved = layernorm_linear_forward(x, g, b if normalization == "LayerNorm" else None, w, None, st,
                                            normalization=normalization)
        st.update_fwd()
        dx, dgamma, dbeta, dw, _ = layernorm_linear_backward(dy, saved, g, st, normalization=normalization)
        st.update_bwd()
    torch.cuda.synchronize()
    xf = x.float().requires_grad_(True)
    gf, bf, wf = g.float().requires_grad_(True), b.float().requires_grad_(True), w.float().requires_grad_(True)
    if normalization == "LayerNorm":
        ln = torch.nn.functional.layer_norm(xf, (K,), gf, bf, 1e-5)
    else:
        ln = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * gf
    yr = ln @ wf.t()
    yr.backward(dy.float())
    # E5M2 gradients carry up to 2^-3 relative error per element; over a 512..1024-long contraction
    # that is a few % of the result's RMS with rare ~7-sigma outliers, so: the reference's module
    # tolerance (rtol 0.125, atol 0.0675 x absmax; utils.py:103-106) on 99.99 % of the elements, 2x that
    # bound on all of them, and a relative Frobenius error below 8 %
    def close(got, ref):
        got, ref = got.float(), ref.float()
        err = (got - ref).abs()
        bound = 0.125 * ref.abs() + 0.0675 * ref.abs().max()
        assert (err <= 2 * bound).all()
        assert (err <= bound).float().mean().item() >= 0.9999
        assert (got - ref).norm().item() <= 0.08 * ref.norm().item()
    close(y, yr)
    close(dw, wf.grad)
    close(dx, xf.grad)
    close(dgamma, gf.grad)
    if normalization == "LayerNorm":
        close(dbeta, bf.grad)
