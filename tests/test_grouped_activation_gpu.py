"""Grouped activation / d-activation / dbias forms (nvte_group_<act>, nvte_group_d<act>,
nvte_group_quantize_dbias[_d<act>]) - ref include/transformer_engine/activation.h:65-275, cast.h:167-410,
cast/dispatch/quantize.cuh:430-540.  One launch over all members, member shapes read on the device.

Checked against the single-tensor fused kernels member by member (those are oracle-checked in
tests/test_activation_gpu.py): payload bytes, e8m0 scales and transposes must be bit-identical - same
arithmetic on the same values; dbias per member against the fp64 column sums of the oracle's activation,
tolerance as tests/test_activation_gpu.py::test_dbias_dact_mxfp8 (rtol 2^-7, bf16 output)."""
import numpy as np
import pytest
import torch

import oracle
from helpers import randn_input, to_oracle_input
from transformer_engine_amd.activation import activation, dactivation
from transformer_engine_amd.grouped import group_activation
from transformer_engine_amd.tensor import DType, Float8Quantizer, MXFP8Quantizer

pytestmark = pytest.mark.gpu
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2
ACTS = ["gelu", "silu", "relu", "qgelu", "srelu"]


def _ref(name, x, grad=None):
    if grad is None:
        return oracle.activation("unary", name, to_oracle_input(x))
    return oracle.activation("dact", name, to_oracle_input(x), to_oracle_input(grad))


def _mx(tcol=True, swz=False, fmt=E4):
    q = MXFP8Quantizer(fmt, rowwise=True, columnwise=True, columnwise_transposed=tcol)
    q.optimize_for_gemm = swz
    return q


def _same_mx_member(out, r0, m, K, one, tcol):
    """member rows [r0, r0+m) of the grouped MXFP8 output == the single-tensor output `one`"""
    M = out.logical_shape[0]
    assert torch.equal(out.rowwise_data.view(M, K)[r0:r0 + m], one._rowwise_data.view(m, K))
    if not out.with_gemm_swizzled_scales:
        assert torch.equal(out.scale_inv.view(-1, K // 32)[r0:r0 + m], one._rowwise_scale_inv[:m, : K // 32])
    if tcol:
        assert torch.equal(out.columnwise_data[r0 * K:(r0 + m) * K].view(K, m), one._columnwise_data.view(K, m))
        if not out.with_gemm_swizzled_scales:
            sc = out.columnwise_scale_inv[K * (r0 // 32): K * ((r0 + m) // 32)].view(K, m // 32)
            assert torch.equal(sc, one._columnwise_scale_inv[:K, : m // 32])
    else:
        assert torch.equal(out.columnwise_data.view(M, K)[r0:r0 + m], one._columnwise_data.view(m, K))
        sc = out.columnwise_scale_inv.view(-1, K)[r0 // 32:(r0 + m) // 32]
        assert torch.equal(sc, one._columnwise_scale_inv[: m // 32, :K])


@pytest.mark.parametrize("name", ACTS)
@pytest.mark.parametrize("tcol", [True, False])
def test_group_activation_mxfp8_equals_per_member(name, tcol):
    splits, K = [256, 128, 0, 384], 256
    x = randn_input((sum(splits), K), torch.bfloat16, 51) * 2
    g = randn_input((sum(splits), K), torch.bfloat16, 52)
    fd = torch.tensor(splits, dtype=torch.int64, device="cuda")
    fwd = group_activation(name, x, _mx(tcol), len(splits), fd)
    bwd = group_activation(name, x, _mx(tcol, fmt=E5), len(splits), fd, grad=g)
    torch.cuda.synchronize()
    r0 = 0
    for m in splits:
        if m:
            _same_mx_member(fwd, r0, m, K, activation(name, x[r0:r0 + m].contiguous(), _mx(tcol)), tcol)
            _same_mx_member(bwd, r0, m, K, dactivation(name, g[r0:r0 + m].contiguous(), x[r0:r0 + m].contiguous(),
                                                       _mx(tcol, fmt=E5)), tcol)
        r0 += m


def test_group_activation_mxfp8_swizzled_scales_feed_the_grouped_gemm():
    """swizzled + transposed layout: consumed by the grouped GEMM; result equals the compact-scale run"""
    from transformer_engine_amd.grouped import GroupedTensor, general_grouped_gemm_for_grouped_tensor, group_quantize, splits_to_offsets
    splits, K, N = [256, 128], 256, 256
    x = randn_input((sum(splits), K), torch.bfloat16, 53)
    w = randn_input((len(splits) * N, K), torch.bfloat16, 54) * 0.05
    fd = torch.tensor(splits, dtype=torch.int64, device="cuda")
    outs = []
    for swz in (False, True):
        a8 = group_activation("gelu", x, _mx(True, swz), len(splits), fd)
        w8 = group_quantize(w, _mx(True, swz), len(splits))
        y = GroupedTensor(num_tensors=len(splits), logical_shape=(sum(splits), N), dtype=torch.float32,
                          rowwise_data=torch.zeros(sum(splits) * N, device="cuda"), first_dims=fd,
                          tensor_offsets=splits_to_offsets(fd, N))
        general_grouped_gemm_for_grouped_tensor(w8, a8, y, layout="TN")
        outs.append(y.rowwise_data.clone())
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    assert float(outs[0].abs().max()) > 0


@pytest.mark.parametrize("name", ["gelu", "srelu"])
def test_group_activation_fp8_per_member_scales(name):
    """FP8 per-tensor scaling: member i is cast with scale[i]; ragged members (multiples of 16 rows)."""
    splits, K = [144, 0, 272, 96], 320
    G = len(splits)
    x = randn_input((sum(splits), K), torch.bfloat16, 55) * 2
    fd = torch.tensor(splits, dtype=torch.int64, device="cuda")
    scales = torch.tensor([3.0, 1.0, 0.5, 17.0], device="cuda")
    q = Float8Quantizer(scales, torch.zeros(G, device="cuda"), E4)
    out = group_activation(name, x, q, G, fd)
    torch.cuda.synchronize()
    r0 = 0
    for i, m in enumerate(splits):
        if m:
            q1 = Float8Quantizer(scales[i:i + 1].clone(), torch.zeros(1, device="cuda"), E4)
            one = activation(name, x[r0:r0 + m].contiguous(), q1)
            assert torch.equal(out.rowwise_data.view(-1, K)[r0:r0 + m], one._data.view(m, K))
            assert torch.equal(out.columnwise_data[r0 * K:(r0 + m) * K].view(K, m), one._data.view(m, K).t())
        r0 += m


@pytest.mark.parametrize("name", [None, "gelu", "silu"])
def test_group_quantize_dbias_mxfp8(name):
    splits, K = [256, 0, 128, 384], 384
    G = len(splits)
    x = randn_input((sum(splits), K), torch.bfloat16, 56) * 2
    g = randn_input((sum(splits), K), torch.bfloat16, 57)
    fd = torch.tensor(splits, dtype=torch.int64, device="cuda")
    if name is None:      # plain cast + dbias of the gradient
        out, dbias = group_activation(None, g, _mx(True, fmt=E5), G, fd, with_dbias=True)
        ref_vals = g.float().cpu().numpy()
    else:
        out, dbias = group_activation(name, x, _mx(True, fmt=E5), G, fd, grad=g, with_dbias=True)
        ref_vals = _ref(name, x, g)
    torch.cuda.synchronize()
    assert dbias.shape == (G, K)
    r0 = 0
    for i, m in enumerate(splits):
        want = ref_vals[r0:r0 + m].astype(np.float64).sum(axis=0)
        tol = 2 ** -7 * max(np.abs(ref_vals[r0:r0 + m]).sum(axis=0).max(), 1e-30) if m else 0.0
        assert np.allclose(dbias[i].float().cpu().numpy(), want, rtol=2 ** -7, atol=tol), (name, i)
        if m:
            q1 = _mx(True, fmt=E5)
            one = q1(g[r0:r0 + m].contiguous()) if name is None else dactivation(name, g[r0:r0 + m].contiguous(),
                                                                                   x[r0:r0 + m].contiguous(), q1)
            _same_mx_member(out, r0, m, K, one, True)
        r0 += m
