"""GPU parity of the grouped (multi-expert) path: offsets helpers, grouped amax / scale / quantize and
the persistent grouped GEMM, all with DEVICE-side member shapes.

Oracle: the same per-tensor CPU restatements as the ungrouped tests, applied member by member
(ref tests/cpp/operator/test_cast_current_scaling_grouped.cu compares grouped results with the
per-tensor reference in the same way; tests/cpp/operator/test_grouped_gemm.cu compares each member
with a single GEMM).  Bytes, amax, scale and scale_inv are compared bit-for-bit; GEMMs on
integer-valued operands exactly, on random operands within the tolerance written at the assert.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from helpers import FMT, randn_input, to_oracle_input, u8
from transformer_engine_amd import NVTEError, api
from transformer_engine_amd.grouped import (GroupedTensor, general_grouped_gemm_for_grouped_tensor,
                                            group_quantize, splits_to_offsets)
from transformer_engine_amd.tensor import (DType, Float8CurrentScalingQuantizer, Float8Quantizer, Float8Tensor,
                                           current_stream_ptr, wrap_plain)

pytestmark = pytest.mark.gpu
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2


def _dev_i64(v):
    return torch.tensor(list(v), dtype=torch.int64, device="cuda")


# ------------------------------------------------------------------------------------------ offsets
def test_splits_to_offsets_family():
    splits = [5, 0, 17, 1024, 3, 0, 9]
    s64 = _dev_i64(splits)
    off = splits_to_offsets(s64, 7)
    assert off.cpu().tolist() == [0] + list(np.cumsum(splits) * 7)
    # 200 entries: more than one element per lane of the scanning wave
    big = np.random.default_rng(0).integers(0, 1000, size=200)
    off = splits_to_offsets(_dev_i64(big), 3)
    assert off.cpu().tolist() == [0] + list(np.cumsum(big) * 3)
    # grouped offsets + int32 -> int64 conversions
    out = torch.empty(len(splits) + 1, dtype=torch.int64, device="cuda")
    api.nvte_compute_grouped_tensor_offsets(s64.data_ptr(), out.data_ptr(), len(splits), 4096, current_stream_ptr())
    assert out.cpu().tolist() == [0] + list(np.cumsum(splits) * 4096)
    s32 = torch.tensor(splits, dtype=torch.int32, device="cuda")
    d64 = torch.empty(len(splits), dtype=torch.int64, device="cuda")
    api.nvte_convert_int32_to_int64(s32.data_ptr(), d64.data_ptr(), len(splits), current_stream_ptr())
    assert d64.cpu().tolist() == splits
    api.nvte_convert_int32_to_int64_with_multiplier(s32.data_ptr(), d64.data_ptr(), len(splits), 11, current_stream_ptr())
    assert d64.cpu().tolist() == [11 * x for x in splits]
    # 2-D: sum of first*last, int32 inputs, int64 output
    last = [3, 4, 5, 6, 7, 8, 9]
    wf, wl = wrap_plain(s32), wrap_plain(torch.tensor(last, dtype=torch.int32, device="cuda"))
    o2 = torch.empty(len(splits) + 1, dtype=torch.int64, device="cuda")
    wo = wrap_plain(o2)
    api.nvte_splits_to_offsets_2d(wf.handle, wl.handle, wo.handle, current_stream_ptr())
    assert o2.cpu().tolist() == [0] + list(np.cumsum(np.array(splits) * np.array(last)))
    # multi: inclusive scan without the leading zero (int32 out) and with it (int64 out)
    oa = torch.empty(len(splits), dtype=torch.int32, device="cuda")
    ob = torch.empty(len(splits) + 1, dtype=torch.int64, device="cuda")
    wa, wb = wrap_plain(oa), wrap_plain(ob)
    outs = (C.c_void_p * 2)(wa.handle, wb.handle)
    strides = (C.c_int64 * 2)(2, 5)
    lead = (C.c_int * 2)(0, 1)
    api.nvte_splits_to_offsets_multi(wf.handle, outs, strides, lead, 2, current_stream_ptr())
    assert oa.cpu().tolist() == list(np.cumsum(splits) * 2)
    assert ob.cpu().tolist() == [0] + list(np.cumsum(splits) * 5)
    for w in (wf, wl, wo, wa, wb):
        w.destroy()


# ------------------------------------------------------------------------------------------ quantize
SPLITS = [
    ([256, 128, 384], 256),          # every member a whole number of 128x128 tiles
    ([48, 0, 208, 16, 1008], 384),   # rows % 16 == 0: vector path with partial tiles, one empty member
    ([5, 33, 1, 130], 72),           # odd rows: element path for the transposes
    ([7, 19], 100),                  # cols % 8 != 0: element path everywhere
    ([16, 0, 48, 144, 32, 0, 16, 160] * 8, 128),    # 64 members: the largest group the per-wave member lookup takes
    ([16, 0, 48, 144, 32] * 16, 128),               # 80 members: the block-scan lookup (group_lookup.h)
]


@pytest.mark.parametrize("splits,cols", SPLITS)
@pytest.mark.parametrize("in_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("fmt", [E4, E5])
def test_group_quantize_current_scaling_varying_first_dim(splits, cols, in_dtype, fmt):
    rows = sum(splits)
    x = randn_input((rows, cols), in_dtype, 11)
    # very different magnitudes per member, so that a scale applied to the wrong member shows
    r0 = 0
    for i, s in enumerate(splits):
        x[r0:r0 + s] *= 4.0 ** (i % 5 - 1)
        r0 += s
    q = Float8CurrentScalingQuantizer(fmt, torch.device("cuda"))
    g = group_quantize(x, q, len(splits), first_dims=_dev_i64(splits))
    torch.cuda.synchronize()
    xn = to_oracle_input(x)
    row_b, col_b = u8(g.rowwise_data), u8(g.columnwise_data)
    amax, scale, sinv = g.amax.cpu().numpy(), g.scale.cpu().numpy(), g.scale_inv.cpu().numpy()
    assert g.columnwise_scale_inv is g.scale_inv
    r0 = 0
    for i, s in enumerate(splits):
        mem = xn[r0:r0 + s]
        ref_amax = oracle.amax(mem) if s else np.float32(0)
        ref_scale = oracle.scale_from_amax(float(ref_amax), FMT[fmt])
        assert amax[i] == ref_amax, (i, amax[i], ref_amax)                       # atol 0
        assert scale[i] == ref_scale and sinv[i] == np.float32(1.0) / ref_scale  # atol 0
        if s:
            ref_c, ref_t, _, _ = oracle.cast_transpose(mem, float(ref_scale), FMT[fmt])
            assert np.array_equal(row_b[r0 * cols:(r0 + s) * cols].reshape(s, cols), ref_c), f"member {i} rowwise"
            assert np.array_equal(col_b[r0 * cols:(r0 + s) * cols].reshape(cols, s), ref_t), f"member {i} transpose"
        r0 += s


def test_group_quantize_same_shape_and_rowwise_only():
    T, M, N = 4, 160, 256
    x = randn_input((T * M, N), torch.bfloat16, 12)
    q = Float8CurrentScalingQuantizer(E4, torch.device("cuda"), columnwise=False, force_pow_2_scales=True)
    g = group_quantize(x, q, T)
    torch.cuda.synchronize()
    assert g.columnwise_data is None
    xn = to_oracle_input(x)
    for i in range(T):
        mem = xn[i * M:(i + 1) * M]
        sc = oracle.scale_from_amax(float(oracle.amax(mem)), oracle.E4M3, force_pow_2=True)
        assert g.scale[i].item() == sc
        ref_c, _, _, _ = oracle.cast_transpose(mem, float(sc), oracle.E4M3, transpose=False)
        assert np.array_equal(u8(g.rowwise_data)[i * M * N:(i + 1) * M * N].reshape(M, N), ref_c)


def test_group_quantize_delayed_scaling_and_noop():
    splits, cols = [64, 192, 32], 128
    x = randn_input((sum(splits), cols), torch.bfloat16, 13)
    scale = torch.tensor([2.0, 0.5, 8.0], device="cuda")
    amax = torch.zeros(3, device="cuda")
    q = Float8Quantizer(scale, amax, E4)
    g = group_quantize(x, q, 3, first_dims=_dev_i64(splits))
    torch.cuda.synchronize()
    xn = to_oracle_input(x)
    r0 = 0
    for i, s in enumerate(splits):
        ref_c, ref_t, ref_amax, _ = oracle.cast_transpose(xn[r0:r0 + s], float(scale[i]), oracle.E4M3)
        assert np.array_equal(u8(g.rowwise_data)[r0 * cols:(r0 + s) * cols].reshape(s, cols), ref_c)
        assert np.array_equal(u8(g.columnwise_data)[r0 * cols:(r0 + s) * cols].reshape(cols, s), ref_t)
        assert amax[i].item() == ref_amax
        r0 += s
    # noop flag set: nothing is written (ref cast/fp8/group_quantize_fp8.cuh noop handling)
    before = g.rowwise_data.clone()
    noop = torch.ones(1, device="cuda")
    g2 = group_quantize(x * 3, Float8CurrentScalingQuantizer(E4, torch.device("cuda")), 3,
                        first_dims=_dev_i64(splits), noop_flag=noop)
    g2.rowwise_data.copy_(before)
    gi = GroupedTensor(num_tensors=3, logical_shape=x.shape, dtype=x.dtype, rowwise_data=x * 3,
                       first_dims=g.first_dims, tensor_offsets=g.tensor_offsets)
    wi, wo, wn = gi.wrap(), g2.wrap(), wrap_plain(noop)
    from transformer_engine_amd.tensor import QuantizationConfig
    cfg = QuantizationConfig(False, 0.0, wn)
    api.nvte_group_quantize(wi.handle, wo.handle, cfg.handle, current_stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(g2.rowwise_data, before)
    for w in (wi, wo, wn):
        w.destroy()


def test_group_dequantize_roundtrip():
    splits, cols = [48, 0, 80], 64
    x = randn_input((sum(splits), cols), torch.bfloat16, 14)
    g = group_quantize(x, Float8CurrentScalingQuantizer(E4, torch.device("cuda"), columnwise=False), 3,
                       first_dims=_dev_i64(splits))
    out = torch.empty_like(x, dtype=torch.float32)
    go = GroupedTensor(num_tensors=3, logical_shape=x.shape, dtype=torch.float32, rowwise_data=out,
                       first_dims=g.first_dims, tensor_offsets=g.tensor_offsets)
    wi, wo = g.wrap(), go.wrap()
    api.nvte_group_dequantize(wi.handle, wo.handle, current_stream_ptr())
    torch.cuda.synchronize()
    wi.destroy(); wo.destroy()
    r0 = 0
    for i, s in enumerate(splits):
        if s:
            ref = oracle.dequant(u8(g.rowwise_data)[r0 * cols:(r0 + s) * cols], oracle.E4M3, float(g.scale_inv[i]))
            assert np.array_equal(out[r0:r0 + s].cpu().numpy().reshape(-1), ref)
        r0 += s


# ------------------------------------------------------------------------------------------ GEMM
def _int_group(shapes, fp8, seed, scale_inv, lo=-3, hi=4):
    """Grouped FP8 tensor whose members hold small integers exactly; both layouts present."""
    g = np.random.default_rng(seed)
    vals = [g.integers(lo, hi, size=s).astype(np.float32) for s in shapes]
    rowb = np.concatenate([oracle.float_to_fp8(v, FMT[fp8]).reshape(-1) for v in vals]) if vals else np.zeros(0, np.uint8)
    colb = np.concatenate([np.ascontiguousarray(oracle.float_to_fp8(v, FMT[fp8]).T).reshape(-1) for v in vals])
    first = [s[0] for s in shapes]
    last = [s[1] for s in shapes]
    numel = int(sum(a * b for a, b in shapes))
    same_last = len(set(last)) == 1
    same_first = len(set(first)) == 1
    if same_last:
        logical = (sum(first), last[0])
    elif same_first:
        logical = (first[0], sum(last))
    else:
        logical = (1, numel)
    sinv = torch.tensor(scale_inv, dtype=torch.float32, device="cuda")
    offs = _dev_i64([0] + list(np.cumsum([a * b for a, b in shapes])))
    t = GroupedTensor(num_tensors=len(shapes), logical_shape=logical, dtype=torch.float32, fp8_dtype=fp8,
                      rowwise_data=torch.from_numpy(rowb).cuda(), columnwise_data=torch.from_numpy(colb).cuda(),
                      scale_inv=sinv, columnwise_scale_inv=sinv,
                      first_dims=None if (same_first and same_last) else (_dev_i64(first) if not same_first else None),
                      last_dims=None if same_last else _dev_i64(last),
                      tensor_offsets=None if (same_first and same_last) else offs)
    return t, vals


def _hp_group(shapes, dtype, fill=None):
    first = [s[0] for s in shapes]
    numel = int(sum(a * b for a, b in shapes))
    data = torch.zeros(numel, dtype=dtype, device="cuda") if fill is None else fill.clone()
    same_first = len(set(first)) == 1
    return GroupedTensor(num_tensors=len(shapes), logical_shape=(sum(first), shapes[0][1]), dtype=dtype,
                         rowwise_data=data, first_dims=None if same_first else _dev_i64(first),
                         tensor_offsets=None if same_first else _dev_i64([0] + list(np.cumsum([a * b for a, b in shapes]))))


M_SPLITS = [[256, 512, 256], [300, 0, 16, 1000, 48], [1, 2, 3]]


@pytest.mark.parametrize("m_splits", M_SPLITS)
@pytest.mark.parametrize("discrete_a", [False, True])
def test_grouped_gemm_fprop_exact(m_splits, discrete_a):
    N, K = 320, 192
    G = len(m_splits)
    sx = [0.5 ** i for i in range(G)]
    sw = [2.0 + i for i in range(G)]
    X, xs = _int_group([(m, K) for m in m_splits], E4, 1, sx)
    W, ws = _int_group([(N, K)] * G, E4, 2, sw)
    out = _hp_group([(m, N) for m in m_splits], torch.float32)
    if discrete_a:
        A = []
        for i in range(G):
            codes = oracle.float_to_fp8(ws[i], oracle.E4M3)
            A.append(Float8Tensor(shape=(N, K), dtype=torch.float32, data=torch.from_numpy(codes).cuda(),
                                  fp8_scale_inv=torch.full((1,), sw[i], device="cuda"), fp8_dtype=E4,
                                  data_transpose=torch.from_numpy(np.ascontiguousarray(codes.T)).cuda()))
    else:
        A = W
    general_grouped_gemm_for_grouped_tensor(A, X, out, layout="TN")
    torch.cuda.synchronize()
    y = out.rowwise_data.cpu().numpy()
    r0 = 0
    for i, m in enumerate(m_splits):
        ref = (xs[i] @ ws[i].T * np.float32(sx[i] * sw[i])).astype(np.float32)
        assert np.array_equal(y[r0 * N:(r0 + m) * N].reshape(m, N), ref), f"member {i}"     # exact
        r0 += m


@pytest.mark.parametrize("sm_count", [1, 10, 100, 250])
def test_grouped_gemm_capped_workgroup_count_exact(sm_count):
    """kNVTEGroupedMatmulConfigSMCount caps the persistent grid.  With a grid that is not a multiple of 8
    the XCD-aware remap sends a workgroup's NEXT tile to an earlier member (e.g. 100 tiles on 10
    workgroups: workgroup 6 walks remapped tiles 76, 2, ...), so the member lookup must not assume that
    tiles only move forward (gemm_fp8.hip member lookup)."""
    m_splits = [512, 256, 0, 1280, 300, 768, 48, 1024]      # 4 tile columns x 17 tile rows + partial tiles
    N, K = 1024, 256
    G = len(m_splits)
    sx = [0.5 ** (i % 4) for i in range(G)]
    sw = [1.0 + i for i in range(G)]
    X, xs = _int_group([(m, K) for m in m_splits], E4, 41, sx)
    W, ws = _int_group([(N, K)] * G, E4, 42, sw)
    out = _hp_group([(m, N) for m in m_splits], torch.float32, fill=torch.full((sum(m_splits) * N,), float("nan"), device="cuda"))
    general_grouped_gemm_for_grouped_tensor(W, X, out, layout="TN", sm_count=sm_count)
    torch.cuda.synchronize()
    y = out.rowwise_data.cpu().numpy()
    r0 = 0
    for i, m in enumerate(m_splits):
        ref = (xs[i] @ ws[i].T * np.float32(sx[i] * sw[i])).astype(np.float32)
        assert np.array_equal(y[r0 * N:(r0 + m) * N].reshape(m, N), ref), f"member {i}"     # exact
        r0 += m


@pytest.mark.parametrize("operands", ["transposed", "k_major", "k_major_list"])
def test_grouped_gemm_dgrad_wgrad_exact_with_accumulate_and_alpha_beta(operands):
    """operands: "transposed" = per-member transposed copies (K-contiguous reads); "k_major" = row-wise payloads only,
    the kernel reads W (dgrad) and X, dY (wgrad) K-major through ds_read_b64_tr_b8; "k_major_list" = additionally W as
    a discrete list of tensors without transposes (nvte_grouped_gemm_with_discrete_inputA)."""
    m_splits, N, K = [256, 48, 0, 512], 256, 384
    G = len(m_splits)
    dY, dys = _int_group([(m, N) for m in m_splits], E5, 3, [1.0, 0.5, 2.0, 0.25])
    W, ws = _int_group([(N, K)] * G, E4, 4, [1.0, 2.0, 4.0, 0.5])
    X, xs = _int_group([(m, K) for m in m_splits], E4, 5, [2.0, 1.0, 1.0, 0.5])
    A_w = W
    if operands != "transposed":
        for t in (dY, W, X):
            t.columnwise_data = None
            t.columnwise_scale_inv = None
        if operands == "k_major_list":
            A_w = [Float8Tensor(shape=(N, K), dtype=torch.float32,
                                data=torch.from_numpy(oracle.float_to_fp8(ws[i], oracle.E4M3)).cuda(),
                                fp8_scale_inv=torch.full((1,), [1.0, 2.0, 4.0, 0.5][i], device="cuda"), fp8_dtype=E4,
                                data_transpose=None) for i in range(G)]
    # dgrad: dX_i [m_i, K] = dY_i W_i   (NN: A = W, B = dY)
    dx = _hp_group([(m, K) for m in m_splits], torch.bfloat16)
    general_grouped_gemm_for_grouped_tensor(A_w, dY, dx, layout="NN")
    # wgrad: dW_i [N, K] = dY_i^T X_i   (NT: A = X, B = dY), discrete outputs, beta = per-member accumulate
    base = [torch.full((N, K), float(i + 1), dtype=torch.float32, device="cuda") for i in range(G)]
    outs = [b.clone() for b in base]
    alpha = torch.tensor([1.0, 2.0, 1.0, 0.5], device="cuda")
    beta = torch.tensor([1.0, 0.0, 3.0, 0.5], device="cuda")
    general_grouped_gemm_for_grouped_tensor(X, dY, outs, layout="NT", accumulate=True, alpha=alpha, beta=beta)
    torch.cuda.synchronize()
    r0 = 0
    sdy, sw, sx = [1.0, 0.5, 2.0, 0.25], [1.0, 2.0, 4.0, 0.5], [2.0, 1.0, 1.0, 0.5]
    for i, m in enumerate(m_splits):
        ref = (dys[i] @ ws[i] * np.float32(sdy[i] * sw[i])).astype(np.float32)
        got = dx.rowwise_data.float().cpu().numpy()[r0 * K:(r0 + m) * K].reshape(m, K)
        assert np.array_equal(got, oracle.bf16_bits_to_float(oracle.float_to_bf16_bits(ref)).reshape(m, K))
        refw = (alpha[i].item() * (dys[i].T @ xs[i]) * np.float32(sdy[i] * sx[i]) + beta[i].item() * (i + 1)).astype(np.float32)
        assert np.array_equal(outs[i].cpu().numpy(), refw), f"wgrad member {i}"   # an empty member leaves beta * C
        r0 += m


def test_grouped_gemm_c5_shaped_random_against_dequantized_matmul():
    # a down-scaled C5 (8 experts, ragged token counts from a fixed-seed multinomial, tokens padded to 16)
    G, K, N, tokens = 8, 512, 768, 4096
    g = np.random.default_rng(1234)
    counts = np.bincount(g.integers(0, G, size=tokens), minlength=G)
    m_splits = [int((c + 15) // 16 * 16) for c in counts]
    x = randn_input((sum(m_splits), K), torch.bfloat16, 21)
    w = randn_input((G * N, K), torch.bfloat16, 22, scale=0.05)
    qx = Float8CurrentScalingQuantizer(E4, torch.device("cuda"))
    gx = group_quantize(x, qx, G, first_dims=_dev_i64(m_splits))
    gw = group_quantize(w, Float8CurrentScalingQuantizer(E4, torch.device("cuda")), G)
    out = _hp_group([(m, N) for m in m_splits], torch.bfloat16)
    general_grouped_gemm_for_grouped_tensor(gw, gx, out, layout="TN")
    torch.cuda.synchronize()
    y = out.rowwise_data.float().cpu().numpy()
    r0 = 0
    for i, m in enumerate(m_splits):
        fx = oracle.dequant(u8(gx.rowwise_data)[r0 * K:(r0 + m) * K].reshape(m, K), oracle.E4M3, float(gx.scale_inv[i]))
        fw = oracle.dequant(u8(gw.rowwise_data)[i * N * K:(i + 1) * N * K].reshape(N, K), oracle.E4M3, float(gw.scale_inv[i]))
        ref = oracle.gemm_tn(fx, fw, fast=True)
        # Tolerance (stated): bf16 output rounding 2^-8 relative + 1e-3 of the member's absmax
        assert np.allclose(y[r0 * N:(r0 + m) * N].reshape(m, N), ref, rtol=2 ** -7, atol=1e-3 * np.abs(ref).max())
        r0 += m


def test_grouped_gemm_inconsistent_member_is_loud_not_out_of_bounds():
    # member 1 has K = 40 on the B side (not a multiple of 16 and != A's K): its output must be NaN,
    # the other members are untouched by it
    N = 64
    X, xs = _int_group([(32, 64), (16, 40)], E4, 6, [1.0, 1.0])
    W, ws = _int_group([(N, 64), (N, 64)], E4, 7, [1.0, 1.0])
    out = _hp_group([(32, N), (16, N)], torch.float32)
    general_grouped_gemm_for_grouped_tensor(W, X, out, layout="TN")
    torch.cuda.synchronize()
    y = out.rowwise_data.cpu().numpy()
    assert np.array_equal(y[:32 * N].reshape(32, N), (xs[0] @ ws[0].T).astype(np.float32))
    assert np.isnan(y[32 * N:]).all()


def test_grouped_bias_add_and_scaled():
    m_splits, N = [3, 0, 130, 7], 96
    rows = sum(m_splits)
    base = randn_input((rows, N), torch.bfloat16, 31)
    bias = randn_input((4, N), torch.bfloat16, 32)
    out = _hp_group([(m, N) for m in m_splits], torch.bfloat16, fill=base.reshape(-1))
    gb = GroupedTensor(num_tensors=4, logical_shape=(4, N), dtype=torch.bfloat16, rowwise_data=bias.reshape(-1))
    wo, wb = out.wrap(), gb.wrap()
    api.nvte_grouped_bias_add(wo.handle, wb.handle, current_stream_ptr())
    torch.cuda.synchronize()
    member = np.repeat(np.arange(4), m_splits)
    ref = (base.float() + bias.float()[torch.from_numpy(member).cuda()]).to(torch.bfloat16)
    assert torch.equal(out.rowwise_data.reshape(rows, N), ref)
    scale = torch.rand(rows, device="cuda")
    out.rowwise_data.copy_(base.reshape(-1))
    ws_ = wrap_plain(scale)
    api.nvte_grouped_scaled_bias_add(wo.handle, wb.handle, ws_.handle, current_stream_ptr())
    torch.cuda.synchronize()
    ref = (base.float() + bias.float()[torch.from_numpy(member).cuda()] * scale[:, None]).to(torch.bfloat16)
    assert torch.equal(out.rowwise_data.reshape(rows, N), ref)
    for w in (wo, wb, ws_):
        w.destroy()


# ------------------------------------------------------------------------------------------ GroupedLinear
def test_grouped_linear_fwd_bwd_equals_per_expert_linear():
    """The grouped step must give, member by member, what the (oracle-checked) ungrouped Linear path
    gives for that expert alone: same quantization, same kernel arithmetic -> bit-identical."""
    from transformer_engine_amd.grouped_linear import (GroupedLinearFP8State, grouped_linear_backward,
                                                       grouped_linear_forward)
    from transformer_engine_amd.linear import LinearFP8State, linear_backward, linear_forward
    from transformer_engine_amd.recipe import Float8CurrentScaling
    G, K, N = 4, 256, 384
    m_splits = [128, 0, 400, 48]
    M = sum(m_splits)
    x = randn_input((M, K), torch.bfloat16, 51)
    w = randn_input((G * N, K), torch.bfloat16, 52, scale=0.05)
    dy = randn_input((M, N), torch.bfloat16, 53)
    recipe = Float8CurrentScaling()
    st = GroupedLinearFP8State(recipe, G)
    fd = _dev_i64(m_splits)
    y, saved = grouped_linear_forward(x, w, fd, st)
    main_grads = [torch.full((N, K), 0.25 * i, dtype=torch.float32, device="cuda") for i in range(G)]
    grads = [g.clone() for g in main_grads]
    dx, dws = grouped_linear_backward(dy, saved, fd, st, wgrad_out=grads, accumulate_wgrad=True)
    torch.cuda.synchronize()
    r0 = 0
    for i, m in enumerate(m_splits):
        if m == 0:
            assert torch.equal(dws[i], main_grads[i])      # empty expert: main_grad untouched
            continue
        s1 = LinearFP8State(recipe)
        yi, sv = linear_forward(x[r0:r0 + m].contiguous(), w[i * N:(i + 1) * N].contiguous(), None, s1)
        gi = main_grads[i].clone()
        dxi, dwi, _ = linear_backward(dy[r0:r0 + m].contiguous(), sv, s1, wgrad_out=gi, accumulate_wgrad=True)
        torch.cuda.synchronize()
        assert torch.equal(y[r0:r0 + m], yi), f"fprop member {i}"
        assert torch.equal(dx[r0:r0 + m], dxi), f"dgrad member {i}"
        assert torch.equal(dws[i], gi), f"wgrad member {i}"
        r0 += m


def test_grouped_step_replays_in_a_hip_graph_with_new_splits():
    """Member shapes are device data: a step captured ONCE in a HIP graph must give the right answer
    when only the CONTENTS of first_dims change between replays (same total token count)."""
    from transformer_engine_amd.grouped_linear import (GroupedLinearFP8State, grouped_linear_backward,
                                                       grouped_linear_forward)
    from transformer_engine_amd.recipe import Float8CurrentScaling
    G, K, N, M = 4, 128, 256, 512
    x = randn_input((M, K), torch.bfloat16, 61)
    w = randn_input((G * N, K), torch.bfloat16, 62, scale=0.05)
    dy = randn_input((M, N), torch.bfloat16, 63)
    fd = _dev_i64([128, 128, 128, 128])
    st = GroupedLinearFP8State(Float8CurrentScaling(), G)

    def step():
        y, saved = grouped_linear_forward(x, w, fd, st)
        dx, dws = grouped_linear_backward(dy, saved, fd, st)
        return y, dx, dws

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()                                    # warm-up outside capture (workspace allocation)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            gy, gdx, gdws = step()
    torch.cuda.current_stream().wait_stream(side)
    for splits in ([128, 128, 128, 128], [16, 304, 0, 192], [512, 0, 0, 0]):
        fd.copy_(_dev_i64(splits))
        graph.replay()
        torch.cuda.synchronize()
        got = (gy.clone(), gdx.clone(), [t.clone() for t in gdws])
        ey, edx, edws = step()                    # eager reference with the same splits
        torch.cuda.synchronize()
        assert torch.equal(got[0], ey) and torch.equal(got[1], edx), splits
        for a, b, m in zip(got[2], edws, splits):
            if m:                                  # an empty expert's fresh wgrad buffer is beta*C with beta=0
                assert torch.equal(a, b), splits
