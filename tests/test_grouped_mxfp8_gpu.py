"""Grouped MXFP8 path (the MXFP8 half): nvte_group_quantize on an MXFP8 grouped
output and nvte_grouped_gemm on MXFP8 members, member shapes read on the device.

Reference: common/cast/mxfp8/group_quantize_mxfp8.cuh (same-shape / varying-first-dim grouped tensors are
quantized as one [sum first, last] tensor, :478, :795), gemm/cublaslt_grouped_gemm.cu:283-300, :486-510,
:1117-1198 (operand storage choice and per-member scale offsets), tests/cpp/operator/test_grouped_gemm.cu.
Oracle: oracle.mxfp8 (the restated tests/cpp/operator/test_cast_mxfp8.cu:39-166 loops) per member -
quantized bytes and e8m0 scales bit-exact (ref tolerance: scales 0 mismatches, :309-320); GEMM against
the dequantized fp64 matmul with the MXFP8 GEMM tolerance of tests/test_gemm_gpu.py (rtol 1e-3 / atol
1e-3 absmax, fp32 output) and bit-exact against the un-grouped MXFP8 GEMM of the same member."""
import numpy as np
import pytest
import torch

import oracle
from transformer_engine_amd import general_gemm
from transformer_engine_amd._lib import NVTEError
from transformer_engine_amd.grouped import (GroupedTensor, general_grouped_gemm_for_grouped_tensor, group_quantize,
                                            splits_to_offsets)
from transformer_engine_amd.tensor import DType, MXFP8Quantizer

pytestmark = pytest.mark.gpu
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2


def _bits(t):
    return oracle.float_to_bf16_bits(t.float().cpu().numpy())


def _u8(t):
    return t.cpu().numpy().astype(np.uint8)


def _inputs(splits, K, seed, scale=1.0):
    g = np.random.default_rng(seed)
    M = sum(splits)
    x = g.standard_normal((M, K), dtype=np.float32) * scale
    # very different block magnitudes, so a scale paired with the wrong block or member shows
    x[:, : K // 2] *= 32.0
    r0 = 0
    for i, m in enumerate(splits):
        x[r0:r0 + m] *= 4.0 ** (i % 3)
        r0 += m
    return torch.from_numpy(x).cuda().to(torch.bfloat16)


def _unswizzle_rows(sw, rows, stride):
    """GEMM-swizzled [rows, stride] scale matrix back to compact (ref cast/mxfp8/swizzle.cuh:30-41)."""
    out = np.zeros((rows, stride), np.uint8)
    i, j = np.meshgrid(np.arange(rows), np.arange(stride), indexing="ij")
    idx = ((i // 128) * (stride // 4) + j // 4) * 512 + (i % 32) * 16 + ((i % 128) // 32) * 4 + j % 4
    out[i, j] = sw[idx]
    return out


@pytest.mark.parametrize("swizzled", [False, True])
@pytest.mark.parametrize("splits", [[256, 128, 384], [128, 0, 256, 0, 128], [512]])
@pytest.mark.parametrize("fmt,ofmt", [(E4, oracle.E4M3), (E5, oracle.E5M2)])
def test_group_quantize_mxfp8_bit_exact(splits, swizzled, fmt, ofmt):
    K = 256
    x = _inputs(splits, K, 3)
    M = x.shape[0]
    fd = torch.tensor(splits, dtype=torch.int64, device="cuda")
    # (a) row + transposed columnwise (what the grouped GEMMs read), (b) row + reference columnwise
    for tcol in (True, False):
        q = MXFP8Quantizer(fmt, rowwise=True, columnwise=True, columnwise_transposed=tcol)
        q.optimize_for_gemm = swizzled
        out = group_quantize(x, q, len(splits), first_dims=fd)
        torch.cuda.synchronize()
        stride_r = K // 32
        sc_row_all = _u8(out.scale_inv)
        sc_row_all = _unswizzle_rows(sc_row_all, M, stride_r) if swizzled else sc_row_all.reshape(-1, stride_r)
        row_all = _u8(out.rowwise_data).reshape(M, K)
        col_flat, sc_col_flat = _u8(out.columnwise_data), _u8(out.columnwise_scale_inv)
        r0 = 0
        for m in splits:
            if m == 0:
                continue
            ref = oracle.mxfp8(_bits(x[r0:r0 + m]), ofmt)
            assert np.array_equal(row_all[r0:r0 + m], ref["out_row"])
            assert np.array_equal(sc_row_all[r0:r0 + m], ref["sc_row"][:m, :stride_r])
            if tcol:
                got = col_flat[r0 * K:(r0 + m) * K].reshape(K, m)
                assert np.array_equal(got, ref["out_col"].T)
                sc = sc_col_flat[K * (r0 // 32): K * ((r0 + m) // 32)]       # pad128(K) == K here
                sc = _unswizzle_rows(sc, K, m // 32) if swizzled else sc.reshape(K, m // 32)
                assert np.array_equal(sc, ref["sc_col"][: m // 32, :K].T)
            else:
                got = col_flat[r0 * K:(r0 + m) * K].reshape(m, K)
                assert np.array_equal(got, ref["out_col"])
                if not swizzled:
                    sc = sc_col_flat.reshape(-1, K)[r0 // 32:(r0 + m) // 32]
                    assert np.array_equal(sc, ref["sc_col"][: m // 32, :K])
                else:
                    # GEMM-swizzled: a matrix per member, [pad128(K), m / 32] swizzled, behind those of the members
                    # before it (ref group_quantize_mxfp8.cuh:114-126) - the bytes of the transposed layout's scales
                    sc = _unswizzle_rows(sc_col_flat[K * (r0 // 32): K * ((r0 + m) // 32)], K, m // 32)
                    assert np.array_equal(sc, ref["sc_col"][: m // 32, :K].T)
            r0 += m


def test_group_quantize_mxfp8_member_rows_off_the_128_contract_stay_inside_the_member():
    """Members whose rows are multiples of 32 but not of 128 take the kernel's guarded tile body (grouped.hip):
    every whole 32-row block is still quantized bit-exactly and nothing of the neighbour is touched."""
    splits, K = [160, 96], 256
    x = _inputs(splits, K, 9)
    fd = torch.tensor(splits, dtype=torch.int64, device="cuda")
    q = MXFP8Quantizer(E4, rowwise=True, columnwise=True, columnwise_transposed=False)
    out = group_quantize(x, q, len(splits), first_dims=fd)
    torch.cuda.synchronize()
    M = sum(splits)
    row_all, col_all = _u8(out.rowwise_data).reshape(M, K), _u8(out.columnwise_data).reshape(M, K)
    sc_row_all = _u8(out.scale_inv).reshape(-1, K // 32)
    sc_col_all = _u8(out.columnwise_scale_inv).reshape(-1, K)
    r0 = 0
    for m in splits:
        ref = oracle.mxfp8(_bits(x[r0:r0 + m]), oracle.E4M3)
        assert np.array_equal(row_all[r0:r0 + m], ref["out_row"])
        assert np.array_equal(sc_row_all[r0:r0 + m], ref["sc_row"][:m, :K // 32])
        assert np.array_equal(col_all[r0:r0 + m], ref["out_col"])
        assert np.array_equal(sc_col_all[r0 // 32:(r0 + m) // 32], ref["sc_col"][: m // 32, :K])
        r0 += m


def test_group_quantize_mxfp8_uniform_members_equals_single_tensor_quantize():
    """Same-shape members: the grouped launch writes exactly what MXFP8Quantizer writes for the stacked tensor
    (rowwise) and for each member (transposed columnwise)."""
    G, N, K = 4, 256, 384
    w = _inputs([N] * G, K, 5, scale=0.05)
    q = MXFP8Quantizer(E4, rowwise=True, columnwise=True, columnwise_transposed=True)
    gq = group_quantize(w, q, G)
    one = q(w)
    torch.cuda.synchronize()
    assert torch.equal(gq.rowwise_data.view(G * N, K), one._rowwise_data.view(G * N, K))
    assert torch.equal(gq.scale_inv.view(-1, K // 32)[: G * N], one._rowwise_scale_inv[: G * N, : K // 32])
    for i in range(G):
        mem = q(w[i * N:(i + 1) * N].contiguous())
        assert torch.equal(gq.columnwise_data[i * N * K:(i + 1) * N * K].view(K, N), mem._columnwise_data.view(K, N))


def _grouped_case(splits, K, N, seed):
    G = len(splits)
    x = _inputs(splits, K, seed)
    w = _inputs([N] * G, K, seed + 1, scale=0.05)
    dy = _inputs(splits, N, seed + 2, scale=0.01)
    fd = torch.tensor(splits, dtype=torch.int64, device="cuda")
    return G, x, w, dy, fd


def _hp(fd, rows, cols, G, dtype=torch.float32):
    return GroupedTensor(num_tensors=G, logical_shape=(rows, cols), dtype=dtype,
                         rowwise_data=torch.full((rows * cols,), float("nan"), dtype=dtype, device="cuda"),
                         first_dims=fd, tensor_offsets=splits_to_offsets(fd, cols))


@pytest.mark.parametrize("swizzled", [False, True])
@pytest.mark.parametrize("splits", [[256, 128, 384, 256], [128, 0, 512, 0]])
@pytest.mark.parametrize("col_layout", ["transposed", "reference"])
def test_grouped_gemm_mxfp8_fprop_dgrad_wgrad(splits, swizzled, col_layout):
    """col_layout "reference": the members' columnwise copies keep the members' own shape, the layout the reference's
    grouped quantize writes (ref cublaslt_grouped_gemm.cu:551-565 hands them to cuBLASLt as non-TN operands); the
    grouped kernel reads them K-major (ds_read_b64_tr_b8).  "transposed": the amd_ext.h extension."""
    K, N = 512, 384
    G, x, w, dy, fd = _grouped_case(splits, K, N, 11)
    M = x.shape[0]

    def mk(dt):
        q = MXFP8Quantizer(dt, rowwise=True, columnwise=True, columnwise_transposed=col_layout == "transposed")
        q.optimize_for_gemm = swizzled
        return q

    x8 = group_quantize(x, mk(E4), G, first_dims=fd)
    w8 = group_quantize(w, mk(E4), G)
    dy8 = group_quantize(dy, mk(E5), G, first_dims=fd)
    y = _hp(fd, M, N, G)
    general_grouped_gemm_for_grouped_tensor(w8, x8, y, layout="TN")
    dx = _hp(fd, M, K, G)
    general_grouped_gemm_for_grouped_tensor(w8, dy8, dx, layout="NN")
    dws = [torch.full((N, K), float("nan"), dtype=torch.float32, device="cuda") for _ in range(G)]
    general_grouped_gemm_for_grouped_tensor(x8, dy8, dws, layout="NT")
    torch.cuda.synchronize()
    y, dx = y.rowwise_data.view(M, N).cpu().numpy(), dx.rowwise_data.view(M, K).cpu().numpy()
    d = oracle.dequant_mxfp8
    r0 = 0
    for i, m in enumerate(splits):
        wi = w[i * N:(i + 1) * N]
        if m == 0:
            # ref cublaslt_grouped_gemm: an empty member writes nothing to Y / dX; its wgrad is all zeros
            assert torch.equal(dws[i], torch.zeros_like(dws[i]))
            continue
        rx, rw, rdy = oracle.mxfp8(_bits(x[r0:r0 + m]), oracle.E4M3), oracle.mxfp8(_bits(wi), oracle.E4M3), \
            oracle.mxfp8(_bits(dy[r0:r0 + m]), oracle.E5M2)
        fx_r, fw_r = d(rx["out_row"], oracle.E4M3, rx["sc_row"], True), d(rw["out_row"], oracle.E4M3, rw["sc_row"], True)
        fdy_r, fw_c = d(rdy["out_row"], oracle.E5M2, rdy["sc_row"], True), d(rw["out_col"], oracle.E4M3, rw["sc_col"], False)
        fdy_c, fx_c = d(rdy["out_col"], oracle.E5M2, rdy["sc_col"], False), d(rx["out_col"], oracle.E4M3, rx["sc_col"], False)
        refs = ((y[r0:r0 + m], fx_r.astype(np.float64) @ fw_r.astype(np.float64).T),
                (dx[r0:r0 + m], fdy_r.astype(np.float64) @ fw_c.astype(np.float64)),
                (dws[i].cpu().numpy(), fdy_c.astype(np.float64).T @ fx_c.astype(np.float64)))
        for got, ref in refs:
            assert got.shape == ref.shape
            # Tolerance (stated): fp32 output, rtol 1e-3 / atol 1e-3 * absmax (FP8 MFMA internal precision)
            assert np.all(np.abs(got - ref) <= 1e-3 * np.abs(ref) + 1e-3 * np.abs(ref).max())
        # and bit-exact against the un-grouped MXFP8 GEMM on the same member (same tiles, same K order)
        qm = MXFP8Quantizer(E4, rowwise=True, columnwise=False)
        y1, *_ = general_gemm(qm(wi.contiguous()), qm(x[r0:r0 + m].contiguous()), out_dtype=torch.float32, layout="TN")
        assert np.array_equal(y1.cpu().numpy(), y[r0:r0 + m])
        r0 += m


def test_grouped_gemm_mxfp8_accumulate_into_wgrad():
    splits, K, N = [256, 128], 256, 256
    G, x, w, dy, fd = _grouped_case(splits, K, N, 21)
    q = lambda dt: MXFP8Quantizer(dt, rowwise=True, columnwise=True, columnwise_transposed=True)
    x8, dy8 = group_quantize(x, q(E4), G, first_dims=fd), group_quantize(dy, q(E5), G, first_dims=fd)
    base = [torch.randn(N, K, device="cuda") for _ in range(G)]
    acc = [b.clone() for b in base]
    fresh = [torch.empty(N, K, device="cuda") for _ in range(G)]
    general_grouped_gemm_for_grouped_tensor(x8, dy8, fresh, layout="NT")
    general_grouped_gemm_for_grouped_tensor(x8, dy8, acc, layout="NT", accumulate=True)
    torch.cuda.synchronize()
    for a, b, f in zip(acc, base, fresh):
        assert torch.allclose(a, b + f, rtol=1e-6, atol=1e-6)


def test_grouped_gemm_mxfp8_member_off_the_128_contract_is_loud():
    """A member whose first dim is not a multiple of 128 (amd_ext.h contract; m_splits live on the device, so the
    host cannot check) misplaces the scale offsets of itself AND of every later member: those members' outputs
    are NaN, the members before it are still exact."""
    K, N = 256, 256
    for splits, bad_from in (([256, 160, 224], 1), ([128, 256], None), ([160, 96], 0)):
        G, x, w, dy, fd = _grouped_case(splits, K, N, 41)
        M = x.shape[0]
        q = lambda dt: MXFP8Quantizer(dt, rowwise=True, columnwise=True, columnwise_transposed=True)
        x8, w8 = group_quantize(x, q(E4), G, first_dims=fd), group_quantize(w, q(E4), G)
        y = _hp(fd, M, N, G)
        general_grouped_gemm_for_grouped_tensor(w8, x8, y, layout="TN")
        torch.cuda.synchronize()
        yv = y.rowwise_data.view(M, N)
        r0 = 0
        for i, m in enumerate(splits):
            blk = yv[r0:r0 + m]
            if bad_from is not None and i >= bad_from:
                assert torch.isnan(blk).all(), f"splits {splits}: member {i} must be NaN"
            else:
                qm = MXFP8Quantizer(E4, rowwise=True, columnwise=False)
                y1, *_ = general_gemm(qm(w[i * N:(i + 1) * N].contiguous()), qm(x[r0:r0 + m].contiguous()),
                                      out_dtype=torch.float32, layout="TN")
                assert torch.equal(y1, blk), f"splits {splits}: member {i} differs from the un-grouped GEMM"
            r0 += m


@pytest.mark.parametrize("sm_count", [0, 10])
@pytest.mark.parametrize("swizzled", [False, True])
def test_grouped_gemm_mxfp8_reference_columnwise_layout_equals_transposed(swizzled, sm_count):
    """(Round 2 refused the reference's columnwise layout here.)  dgrad and wgrad on members in the reference layout
    are bit-equal to the same GEMMs on the transposed copies: same tiles, same K order.  Ragged members, an empty
    one, tiles past the member's rows (N = 384 is 1.5 tiles)."""
    splits, K, N = [384, 0, 128, 640], 768, 384
    G, x, w, dy, fd = _grouped_case(splits, K, N, 31)
    M = x.shape[0]
    outs = {}
    for tcol in (True, False):
        def mk(dt):
            q = MXFP8Quantizer(dt, rowwise=True, columnwise=True, columnwise_transposed=tcol)
            q.optimize_for_gemm = swizzled
            return q
        x8 = group_quantize(x, mk(E4), G, first_dims=fd)
        w8 = group_quantize(w, mk(E4), G)
        dy8 = group_quantize(dy, mk(E5), G, first_dims=fd)
        dx = _hp(fd, M, K, G)
        cap = 0 if tcol else sm_count      # (the K-major launch also with a capped, odd workgroup count)
        general_grouped_gemm_for_grouped_tensor(w8, dy8, dx, layout="NN", sm_count=cap)
        dws = [torch.full((N, K), float("nan"), dtype=torch.float32, device="cuda") for _ in range(G)]
        general_grouped_gemm_for_grouped_tensor(x8, dy8, dws, layout="NT", sm_count=cap)
        torch.cuda.synchronize()
        outs[tcol] = (dx.rowwise_data.clone(), [d.clone() for d in dws], x8, dy8)
    assert torch.equal(outs[True][0], outs[False][0])
    assert not torch.isnan(outs[False][0]).any()
    for a, b in zip(outs[True][1], outs[False][1]):
        assert torch.equal(a, b)
    # what the K-major path does not take is refused loudly: one operand in each layout for wgrad
    with pytest.raises(NVTEError, match="columnwise layout"):
        general_grouped_gemm_for_grouped_tensor(outs[True][2], outs[False][3],
                                                [torch.empty(N, K, device="cuda") for _ in range(G)], layout="NT")


def test_group_quantize_mxfp8_rejects_unaligned_shapes():
    x = torch.randn(200, 256, device="cuda", dtype=torch.bfloat16)
    with pytest.raises((NVTEError, ValueError), match="128"):
        group_quantize(x, MXFP8Quantizer(E4), 2)


@pytest.mark.parametrize("mx_columnwise", ["reference", "transposed"])
def test_grouped_linear_mxfp8_fwd_bwd_matches_per_member_linear(mx_columnwise):
    """GroupedLinear with the MXFP8 recipe against the un-grouped MXFP8 Linear per expert (the same
    kernels member by member: bit-exact), in both columnwise layouts."""
    from transformer_engine_amd import linear as L
    from transformer_engine_amd.grouped_linear import (GroupedLinearFP8State, grouped_linear_backward,
                                                       grouped_linear_forward)
    from transformer_engine_amd.recipe import MXFP8BlockScaling
    splits, K, N = [256, 384, 128], 512, 256
    G, x, w, dy, fd = _grouped_case(splits, K, N, 41)
    st = GroupedLinearFP8State(MXFP8BlockScaling(), G, torch.device("cuda"), mx_columnwise=mx_columnwise)
    y, saved = grouped_linear_forward(x, w, fd, st)
    dx, dws = grouped_linear_backward(dy, saved, fd, st)
    torch.cuda.synchronize()
    r0 = 0
    for i, m in enumerate(splits):
        s1 = L.LinearFP8State(MXFP8BlockScaling(), device=torch.device("cuda"))
        y1, sv = L.linear_forward(x[r0:r0 + m].contiguous(), w[i * N:(i + 1) * N].contiguous(), None, s1)
        dx1, dw1, _ = L.linear_backward(dy[r0:r0 + m].contiguous(), sv, s1)
        torch.cuda.synchronize()
        assert torch.equal(y[r0:r0 + m], y1)
        assert torch.equal(dx[r0:r0 + m], dx1)
        assert torch.equal(dws[i], dw1)
        r0 += m


@pytest.mark.parametrize("tcol", [True, False])
def test_swizzle_grouped_scaling_factors_round_trip(tcol):
    """nvte_swizzle_grouped_scaling_factors / nvte_unswizzle_grouped_scaling_factors (ref swizzle.h:118-134,
    same-shape members): swizzling the compact scales of a grouped quantize gives exactly the scales the
    quantize emits when asked for the GEMM layout directly (oracle-checked above); unswizzling returns the input."""
    import copy
    from transformer_engine_amd._lib import api
    from transformer_engine_amd.tensor import current_stream_ptr
    G, N, K = 3, 256, 384
    w = _inputs([N] * G, K, 61, scale=0.05)

    def mk(swz):
        q = MXFP8Quantizer(E4, rowwise=True, columnwise=True, columnwise_transposed=tcol)
        q.optimize_for_gemm = swz
        return q

    compact = group_quantize(w, mk(False), G)
    direct = group_quantize(w, mk(True), G)
    swz = copy.copy(compact)
    swz.scale_inv = torch.zeros_like(compact.scale_inv)
    swz.columnwise_scale_inv = torch.zeros_like(compact.columnwise_scale_inv)
    swz.with_gemm_swizzled_scales = True
    back = copy.copy(compact)
    back.scale_inv = torch.zeros_like(compact.scale_inv)
    back.columnwise_scale_inv = torch.zeros_like(compact.columnwise_scale_inv)
    wc, ws, wb = compact.wrap(), swz.wrap(), back.wrap()
    try:
        api.nvte_swizzle_grouped_scaling_factors(wc.handle, ws.handle, current_stream_ptr())
        api.nvte_unswizzle_grouped_scaling_factors(ws.handle, wb.handle, current_stream_ptr())
    finally:
        wc.destroy(); ws.destroy(); wb.destroy()
    torch.cuda.synchronize()
    assert torch.equal(back.scale_inv, compact.scale_inv)
    assert torch.equal(back.columnwise_scale_inv, compact.columnwise_scale_inv)
    assert not torch.equal(swz.scale_inv, compact.scale_inv)
    assert torch.equal(swz.scale_inv, direct.scale_inv)
    assert torch.equal(swz.columnwise_scale_inv, direct.columnwise_scale_inv)
    if not tcol:
        # reference column-wise layout: member i against the oracle's swizzle of its [N/32, K] scale matrix
        sc = _u8(compact.columnwise_scale_inv).reshape(G, N // 32, K)
        got = _u8(swz.columnwise_scale_inv).reshape(G, -1)
        for i in range(G):
            assert np.array_equal(got[i], oracle.swizzle_col_scales(sc[i]))
    # a swizzled input to swizzle (and a compact one to unswizzle) is an error, as for single tensors
    wc, ws = compact.wrap(), swz.wrap()
    try:
        with pytest.raises(NVTEError, match="already in GEMM layout"):
            api.nvte_swizzle_grouped_scaling_factors(ws.handle, ws.handle, current_stream_ptr())
        with pytest.raises(NVTEError, match="not in GEMM layout"):
            api.nvte_unswizzle_grouped_scaling_factors(wc.handle, wc.handle, current_stream_ptr())
    finally:
        wc.destroy(); ws.destroy()
