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
