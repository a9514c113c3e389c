"""GPU parity tests added in round 2 (VERDICT r1 "What's weak" #1): entry points that were exported
but never called, the MXFP8 GEMM on many-round persistent walks, the committed golden fixtures fed
through the HIP path, and the reference MXFP8 shapes that had been dropped.

Everything goes through the C ABI (ctypes).  Oracles: the CPU restatement (`oracle/`), the golden
fixtures under tests/golden/ (generated from the reference's own `scale_from_amax_tensor`, see
gen_goldens.py), or exact integer / power-of-two arithmetic computed in fp64 on the GPU.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import FMT, randn_input, to_oracle_input, u8, uniform_input
from transformer_engine_amd import _lib, general_gemm
from transformer_engine_amd._lib import api
from transformer_engine_amd.gemm import general_grouped_gemm
from transformer_engine_amd.tensor import (DType, Float8CurrentScalingQuantizer, Float8Quantizer, Float8Tensor,
                                           MXFP8Quantizer, MXFP8Tensor, QuantizationConfig, TensorWrapper,
                                           current_stream_ptr, wrap_plain)

pytestmark = pytest.mark.gpu
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _int_fp8_tensor(shape, fp8, seed, lo=-3, hi=4, scale_inv=1.0):
    g = np.random.default_rng(seed)
    vals = g.integers(lo, hi, size=shape).astype(np.float32)
    codes = oracle.float_to_fp8(vals, FMT[fp8])
    sinv = torch.full((1,), scale_inv, dtype=torch.float32, device="cuda")
    t = Float8Tensor(shape=shape, dtype=torch.float32, data=torch.from_numpy(codes).cuda(), fp8_scale_inv=sinv,
                     fp8_dtype=fp8, data_transpose=torch.from_numpy(np.ascontiguousarray(codes.T)).cuda())
    return t, vals


def _pad(a, b):
    return (a + b - 1) // b * b


def _int_mx_tensor(rows, k, fp8, seed, swizzled):
    """Rowwise MXFP8 tensor built by hand: integer payload in [-3, 3], e8m0 block scales 2^-2 .. 2^2.
    Every product and every partial sum of a GEMM over such operands is exact in fp32.
    Returns (tensor, dequantized fp64 matrix on the GPU)."""
    g = np.random.default_rng(seed)
    vals = g.integers(-3, 4, size=(rows, k)).astype(np.float32)
    codes = torch.from_numpy(oracle.float_to_fp8(vals, FMT[fp8])).cuda()
    nb = k // 32
    sc = np.zeros((_pad(rows, 128), _pad(nb, 4)), np.uint8)
    sc[:rows, :nb] = g.integers(125, 130, size=(rows, nb)).astype(np.uint8)
    sc_t = torch.from_numpy(sc).cuda()
    t = MXFP8Tensor(shape=(rows, k), dtype=torch.float32, rowwise_data=codes, rowwise_scale_inv=sc_t,
                    columnwise_data=None, columnwise_scale_inv=None, fp8_dtype=fp8)
    if swizzled:
        out = MXFP8Tensor(shape=(rows, k), dtype=torch.float32, rowwise_data=codes,
                          rowwise_scale_inv=torch.zeros_like(sc_t), columnwise_data=None, columnwise_scale_inv=None,
                          fp8_dtype=fp8, with_gemm_swizzled_scales=True)
        wi, wo = t.wrap(), out.wrap()
        api.nvte_swizzle_scaling_factors(wi.handle, wo.handle, current_stream_ptr())
        wi.destroy(); wo.destroy()
        t = out
    scale = torch.from_numpy(np.exp2(sc[:rows, :nb].astype(np.float64) - 127.0)).cuda()
    deq = torch.from_numpy(vals).cuda().double() * scale.repeat_interleave(32, dim=1)
    return t, deq


# ------------------------------------------------------------------------------------------------
# (i) MXFP8 GEMM: persistent walk over many output tiles with the cross-tile scale staging
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [
    (4352, 4352, 384),    # 17 x 17 = 289 tiles > 256 workgroups, odd K-tile count (3)
    (8192, 4608, 640),    # 576 tiles = 2.25 rounds, 5 K tiles
    (4400, 4200, 160),    # ragged edge tiles, K tail inside the second K tile (scale keep/fill mask)
    (8192, 9216, 1152),   # 1152 tiles = 4.5 rounds, 9 K tiles
])
@pytest.mark.parametrize("swizzled", [False, True])
@pytest.mark.parametrize("fa,fb", [(E4, E4), (E4, E5)])
def test_mxfp8_gemm_exact_many_rounds(M, N, K, swizzled, fa, fb):
    if (fa, fb) != (E4, E4) and (M, N, K) != (4352, 4352, 384):
        pytest.skip("mixed formats are covered on one shape")
    W, wd = _int_mx_tensor(N, K, fa, 101, swizzled)
    X, xd = _int_mx_tensor(M, K, fb, 102, swizzled)
    ref = xd @ wd.T                                   # fp64, exact
    for _ in range(2):                                # state-free launches
        y, *_ = general_gemm(W, X, out_dtype=torch.float32, layout="TN")
        torch.cuda.synchronize()
        assert y.shape == (M, N)
        assert torch.equal(y.double(), ref)           # exact: integers x powers of two
    yb, *_ = general_gemm(W, X, out_dtype=torch.bfloat16, layout="TN")
    torch.cuda.synchronize()
    assert torch.equal(yb.float(), ref.float().to(torch.bfloat16).float())


@pytest.mark.parametrize("swizzled", [False, True])
def test_mxfp8_gemm_full_size_configs2_against_gpu_matmul(swizzled):
    """BASELINE configs[2]: MXFP8 8192 x 8192 x 8192 (1024 tiles = 4 rounds per workgroup, 64 K tiles),
    quantized by this library, checked against torch.matmul (fp32, GPU) of operands dequantized with
    torch's own FP8 / e8m0 decode.  Tolerance: rtol 1.6e-2 (bf16 output rounding + MFMA accumulation),
    atol 1e-3 * max|ref| - the same bound as test_gemm_full_size_configs1_against_gpu_matmul."""
    n = 8192
    g = torch.Generator(device="cuda").manual_seed(4321)
    a = torch.randn(n, n, device="cuda", dtype=torch.bfloat16, generator=g)
    b = (torch.randn(n, n, device="cuda", dtype=torch.float32, generator=g) * 0.05).to(torch.bfloat16)
    a[:, : n // 2] *= 16.0                             # block scales that differ a lot along K
    q = MXFP8Quantizer(E4, columnwise=False)
    q.optimize_for_gemm = swizzled
    qa, qb = q(a), q(b)
    qc = MXFP8Quantizer(E4, columnwise=False)          # compact scales for the decode
    ca, cb = (qa, qb) if not swizzled else (qc(a), qc(b))

    def deq(t):
        sc = t._rowwise_scale_inv[:n, : n // 32].to(torch.int32)
        scale = torch.exp2((sc - 127).float()).repeat_interleave(32, dim=1)
        return t._rowwise_data.view(torch.float8_e4m3fn).to(torch.float32) * scale

    old = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        ref = deq(ca) @ deq(cb).T
        out, *_ = general_gemm(qb, qa, out_dtype=torch.bfloat16, layout="TN")
        torch.cuda.synchronize()
        err = (out.float() - ref).abs()
        bound = 1.6e-2 * ref.abs() + 1e-3 * ref.abs().max()
        assert bool((err <= bound).all()), (float(err.max()), float(ref.abs().max()))
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old


# ------------------------------------------------------------------------------------------------
# (ii) entry points that no test called in round 1
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", ["TN", "NN", "NT"])
def test_multi_tensor_gemm_list_form_exact(layout):
    """nvte_multi_tensor_gemm through general_grouped_gemm (ref pytorch/cpp_extensions/gemm.py:277,
    TE's default GroupedLinear path): 6 independent GEMMs (> the 4 pool streams) of different sizes,
    small integers -> exact."""
    K, N = 192, 320
    m_splits = [256, 48, 512, 16, 304, 1024]
    sx = [0.5, 1.0, 2.0, 0.25, 1.0, 4.0]
    As, Bs, refs, outs = [], [], [], []
    for i, m in enumerate(m_splits):
        W, w = _int_fp8_tensor((N, K), E4, 200 + i, scale_inv=2.0)
        X, x = _int_fp8_tensor((m, K), E4, 300 + i, scale_inv=sx[i])
        dY, dy = _int_fp8_tensor((m, N), E5, 400 + i, scale_inv=0.5)
        if layout == "TN":
            As.append(W); Bs.append(X); refs.append(x @ w.T * np.float32(2.0 * sx[i])); shape = (m, N)
        elif layout == "NN":
            As.append(W); Bs.append(dY); refs.append(dy @ w * np.float32(1.0)); shape = (m, K)
        else:
            As.append(X); Bs.append(dY); refs.append(dy.T @ x * np.float32(0.5 * sx[i])); shape = (N, K)
        outs.append(torch.full(shape, float("nan"), dtype=torch.float32, device="cuda"))
    general_grouped_gemm(As, Bs, outs, torch.float32, layout=layout)
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert np.array_equal(o.cpu().numpy(), r.astype(np.float32)), f"gemm {i}"
    # accumulate into the outputs + the single_output form (one buffer split by m_splits)
    if layout == "TN":
        base = [torch.full_like(o, 3.0) for o in outs]
        general_grouped_gemm(As, Bs, base, torch.float32, layout=layout, accumulate=True)
        whole = torch.empty(sum(m_splits), N, dtype=torch.float32, device="cuda")
        bias = [torch.arange(N, dtype=torch.float32, device="cuda") * (i + 1) for i in range(len(m_splits))]
        general_grouped_gemm(As, Bs, [whole], torch.float32, layout=layout, m_splits=m_splits, bias=bias,
                             single_output=True)
        torch.cuda.synchronize()
        r0 = 0
        for i, m in enumerate(m_splits):
            assert np.array_equal(base[i].cpu().numpy(), refs[i].astype(np.float32) + 3.0)
            assert np.array_equal(whole[r0:r0 + m].cpu().numpy(),
                                  refs[i].astype(np.float32) + bias[i].cpu().numpy()[None, :])
            r0 += m


def test_cublas_gemm_and_scaled_deprecated_entry_points():
    """nvte_cublas_gemm / nvte_cublas_gemm_scaled (ref gemm.h:178,231) called directly."""
    M, N, K = 300, 256, 160
    W, w = _int_fp8_tensor((N, K), E4, 11, scale_inv=0.5)
    X, x = _int_fp8_tensor((M, K), E5, 12, scale_inv=2.0)
    bias = torch.arange(N, dtype=torch.bfloat16, device="cuda") % 9
    base = torch.full((M, N), 2.0, dtype=torch.float32, device="cuda")
    ref = (x @ w.T).astype(np.float32)
    wa, wb = W.wrap(), X.wrap()
    empty, ws = TensorWrapper(), TensorWrapper()
    wbias = wrap_plain(bias)
    s = current_stream_ptr()
    # plain, bias epilogue
    d = torch.empty(M, N, dtype=torch.float32, device="cuda"); wd = wrap_plain(d)
    api.nvte_cublas_gemm(wa.handle, wb.handle, wd.handle, wbias.handle, empty.handle, True, False, False,
                         ws.handle, False, False, 0, s)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), ref + bias.float().cpu().numpy()[None, :])
    # accumulate = beta 1
    d2 = base.clone(); wd2 = wrap_plain(d2)
    api.nvte_cublas_gemm(wa.handle, wb.handle, wd2.handle, empty.handle, empty.handle, True, False, False,
                         ws.handle, True, True, 0, s)
    torch.cuda.synchronize()
    assert np.array_equal(d2.cpu().numpy(), ref + 2.0)
    # scaled: alpha / beta by value
    d3 = base.clone(); wd3 = wrap_plain(d3)
    api.nvte_cublas_gemm_scaled(wa.handle, wb.handle, wd3.handle, empty.handle, empty.handle, True, False, False,
                                ws.handle, 0.5, 4.0, False, 0, s)
    torch.cuda.synchronize()
    assert np.array_equal(d3.cpu().numpy(), 0.5 * ref + 8.0)
    for t in (wa, wb, empty, ws, wbias, wd, wd2, wd3):
        t.destroy()


@pytest.mark.parametrize("mode", ["delayed_ct", "delayed_row", "mxfp8"])
def test_multi_tensor_quantize_equals_oracle(mode):
    """nvte_multi_tensor_quantize (ref cast/cast.cu:57-89): 6 tensors over the 4-stream pool."""
    shapes = [(256, 384), (128, 128), (37, 200), (1024, 512), (16, 48), (512, 256)]
    xs = [uniform_input(s, torch.bfloat16, seed=70 + i) for i, s in enumerate(shapes)]
    n = len(xs)
    outs, wins, wouts, amaxes = [], [], [], []
    for i, x in enumerate(xs):
        if mode == "mxfp8":
            q = MXFP8Quantizer(E4)
            o = q.make_empty(x.shape, dtype=x.dtype, device=x.device)
            wo = o.wrap()
        else:
            scale = torch.full((1,), 0.5 + i, dtype=torch.float32, device="cuda")
            amax = torch.zeros(1, dtype=torch.float32, device="cuda")
            q = Float8Quantizer(scale, amax, E4, columnwise=(mode == "delayed_ct"))
            o = q.make_empty(x.shape, dtype=x.dtype, device=x.device)
            wo = q._wrap_out(o)
            wo.set_scale(scale).set_amax(amax)
            amaxes.append(amax)
        outs.append(o); wouts.append(wo); wins.append(wrap_plain(x))
    cfg = QuantizationConfig()
    ia = (C.c_void_p * n)(*[w.handle for w in wins])
    oa = (C.c_void_p * n)(*[w.handle for w in wouts])
    api.nvte_multi_tensor_quantize(ia, oa, cfg.handle, n, current_stream_ptr())
    torch.cuda.synchronize()
    for i, (x, o) in enumerate(zip(xs, outs)):
        xin = to_oracle_input(x)
        if mode == "mxfp8":
            ref = oracle.mxfp8(xin, oracle.E4M3, True, True)
            r, c = x.shape
            assert np.array_equal(u8(o._rowwise_data), ref["out_row"]) and np.array_equal(u8(o._columnwise_data), ref["out_col"])
            assert np.array_equal(u8(o._rowwise_scale_inv)[:r, :(c + 31) // 32], ref["sc_row"][:r, :(c + 31) // 32])
            assert np.array_equal(u8(o._columnwise_scale_inv)[:(r + 31) // 32, :c], ref["sc_col"][:(r + 31) // 32, :c])
        else:
            rc, rt, ra, rs = oracle.cast_transpose(xin, 0.5 + i, oracle.E4M3, transpose=(mode == "delayed_ct"))
            assert np.array_equal(u8(o._data), rc), i
            if mode == "delayed_ct":
                assert np.array_equal(u8(o._transpose), rt), i
            assert amaxes[i].item() == ra and o._scale_inv.item() == rs
    for w in wins + wouts:
        w.destroy()


def test_multi_cast_transpose_equals_oracle():
    """nvte_multi_cast_transpose (ref transpose/multi_cast_transpose.cu:335), the reference's own test
    shapes (tests/cpp/operator/test_multi_cast_transpose.cu: a mix of aligned and odd sizes)."""
    shapes = [(1, 1), (1, 768), (768, 1), (768, 768), (43, 43), (43, 256), (256, 43), (256, 256)]
    n = len(shapes)
    xs = [uniform_input(s, torch.bfloat16, seed=90 + i) for i, s in enumerate(shapes)]
    outs, wins, wouts, amaxes = [], [], [], []
    for i, x in enumerate(xs):
        scale = torch.full((1,), 1.25 * (i + 1), dtype=torch.float32, device="cuda")
        amax = torch.zeros(1, dtype=torch.float32, device="cuda")
        q = Float8Quantizer(scale, amax, E5)
        o = q.make_empty(x.shape, dtype=x.dtype, device=x.device)
        wo = q._wrap_out(o); wo.set_scale(scale).set_amax(amax)
        outs.append(o); wouts.append(wo); wins.append(wrap_plain(x)); amaxes.append(amax)
    ia = (C.c_void_p * n)(*[w.handle for w in wins])
    oa = (C.c_void_p * n)(*[w.handle for w in wouts])
    api.nvte_multi_cast_transpose(n, ia, oa, current_stream_ptr())
    torch.cuda.synchronize()
    for i, (x, o) in enumerate(zip(xs, outs)):
        rc, rt, ra, rs = oracle.cast_transpose(to_oracle_input(x), 1.25 * (i + 1), oracle.E5M2)
        assert np.array_equal(u8(o._data), rc) and np.array_equal(u8(o._transpose), rt), shapes[i]
        assert amaxes[i].item() == ra and o._scale_inv.item() == rs
    for w in wins + wouts:
        w.destroy()


def test_multi_tensor_swizzle_and_unswizzle_scaling_factors():
    """nvte_multi_tensor_swizzle_scaling_factors / nvte_multi_tensor_unswizzle_scaling_factors
    (ref swizzle.h:31-46) against the index formula of ref cast/mxfp8/swizzle.cuh:30-41 (oracle)."""
    shapes = [(128, 128), (256, 512), (384, 1024), (1024, 256)]
    q = MXFP8Quantizer(E4)
    comps = [q(uniform_input(s, torch.bfloat16, seed=50 + i)) for i, s in enumerate(shapes)]
    swz = []
    for c in comps:
        t = q.make_empty(c.shape, dtype=torch.bfloat16, device="cuda")
        t._with_gemm_swizzled_scales = True
        swz.append(t)
    n = len(shapes)
    wi = [c.wrap() for c in comps]
    wo = [t.wrap() for t in swz]
    ia = (C.c_void_p * n)(*[w.handle for w in wi])
    oa = (C.c_void_p * n)(*[w.handle for w in wo])
    api.nvte_multi_tensor_swizzle_scaling_factors(ia, oa, n, current_stream_ptr())
    torch.cuda.synchronize()
    for c, t in zip(comps, swz):
        assert np.array_equal(u8(t._rowwise_scale_inv).reshape(-1), oracle.swizzle_scales(u8(c._rowwise_scale_inv)))
        assert np.array_equal(u8(t._columnwise_scale_inv).reshape(-1), oracle.swizzle_col_scales(u8(c._columnwise_scale_inv)))
    back = [q.make_empty(c.shape, dtype=torch.bfloat16, device="cuda") for c in comps]
    wb = [t.wrap() for t in back]
    ba = (C.c_void_p * n)(*[w.handle for w in wb])
    api.nvte_multi_tensor_unswizzle_scaling_factors(oa, ba, n, current_stream_ptr())
    torch.cuda.synchronize()
    for c, t in zip(comps, back):
        assert torch.equal(t._rowwise_scale_inv, c._rowwise_scale_inv)
        assert torch.equal(t._columnwise_scale_inv, c._columnwise_scale_inv)
    for w in wi + wo + wb:
        w.destroy()


def test_fp8_transpose_dbias():
    """nvte_fp8_transpose_dbias (ref transpose/transpose_fusion.cu:498): transposed copy of an FP8
    tensor + dbias[c] = scale_inv * sum_r float(in[r, c]) in the dbias dtype, two-call workspace protocol."""
    for shape, fp8 in (((256, 384), E4), ((1000, 136), E5), ((64, 64), E4)):
        X, x = _int_fp8_tensor(shape, fp8, 7, scale_inv=0.25)
        out = torch.empty((shape[1], shape[0]), dtype=torch.uint8, device="cuda")
        dbias = torch.empty(shape[1], dtype=torch.bfloat16, device="cuda")
        wi = TensorWrapper().set_rowwise_data(X._data, fp8, shape)
        wi.set_rowwise_scale_inv(X._scale_inv)
        wo = TensorWrapper().set_rowwise_data(out, fp8)
        wo.set_rowwise_scale_inv(X._scale_inv)          # the reference reads the OUTPUT's scale_inv
        wd, ws = wrap_plain(dbias), TensorWrapper()
        s = current_stream_ptr()
        api.nvte_fp8_transpose_dbias(wi.handle, wo.handle, wd.handle, ws.handle, s)    # workspace query
        bt = ws.get_param(0)
        wshape = [bt.shape.data[i] for i in range(bt.shape.ndim)]
        assert bt.dtype == int(DType.kFloat32) and len(wshape) >= 1
        work = torch.empty(wshape, dtype=torch.float32, device="cuda")
        ws.set_rowwise_data(work)
        api.nvte_fp8_transpose_dbias(wi.handle, wo.handle, wd.handle, ws.handle, s)
        torch.cuda.synchronize()
        assert torch.equal(out, X._data.t().contiguous())
        ref = (x.sum(axis=0) * 0.25).astype(np.float32)           # exact: small integers x 2^-2
        assert np.array_equal(dbias.float().cpu().numpy(), torch.from_numpy(ref).to(torch.bfloat16).float().numpy())
        for w in (wi, wo, wd, ws):
            w.destroy()


# ------------------------------------------------------------------------------------------------
# (iii) the committed golden fixtures straight through the HIP path
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,fp8", [("e4m3_std", E4), ("e4m3_pow2", E4), ("e5m2_std", E5), ("e5m2_pow2", E5)])
def test_golden_current_scaling_128x256_through_hip(tag, fp8):
    """tests/golden/current_scaling_128x256.npz (recipe of ref tests/pytorch/
    test_float8_current_scaling_exact.py:786-809; scale from the reference's own scale_from_amax_tensor):
    amax, scale, scale_inv, bytes and transposed bytes from the HIP kernels equal the fixture."""
    g = np.load(os.path.join(GOLD, "current_scaling_128x256.npz"))
    x = torch.from_numpy(g["x_bf16_bits"].view(np.int16)).view(torch.bfloat16).cuda()
    q = Float8CurrentScalingQuantizer(fp8, torch.device("cuda"), force_pow_2_scales=tag.endswith("pow2"))
    out = q(x)
    torch.cuda.synchronize()
    assert out._amax.item() == g[f"amax_{tag}"][0]
    assert out._scale.item() == g[f"scale_{tag}"][0]
    assert out._scale_inv.item() == g[f"scale_inv_{tag}"][0]
    assert np.array_equal(u8(out._data), g[f"q_{tag}"])
    assert np.array_equal(u8(out._transpose), g[f"qt_{tag}"])


def test_golden_fp8_decode_tables_through_hip():
    """tests/golden/fp8_decode_tables.npz: all 256 codes of both formats dequantized by the HIP kernel."""
    g = np.load(os.path.join(GOLD, "fp8_decode_tables.npz"))
    for key, fp8 in (("e4m3", E4), ("e5m2", E5)):
        name = [k for k in g.files if key in k.lower()]
        assert name, g.files
        table = g[name[0]].astype(np.float32)
        codes = torch.arange(256, dtype=torch.int32).to(torch.uint8).cuda().reshape(16, 16)
        t = Float8Tensor(shape=(16, 16), dtype=torch.float32, data=codes,
                         fp8_scale_inv=torch.ones(1, device="cuda"), fp8_dtype=fp8)
        got = t.dequantize(torch.float32).cpu().numpy().reshape(-1)
        fin = np.isfinite(table)
        assert np.array_equal(got[fin], table[fin])
        assert np.array_equal(np.isnan(got), np.isnan(table))


# ------------------------------------------------------------------------------------------------
# (iv) the reference MXFP8 shapes that round 1 dropped (ref tests/cpp/operator/test_cast_mxfp8.cu:524-533)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("direction", ["row", "col", "both"])
@pytest.mark.parametrize("fp8", [E4, E5])
def test_mxfp8_quantize_shape_1x16(direction, fp8):
    x = uniform_input((1, 16), torch.bfloat16, seed=5)
    q = MXFP8Quantizer(fp8, rowwise=direction in ("row", "both"), columnwise=direction in ("col", "both"))
    out = q(x)
    torch.cuda.synchronize()
    ref = oracle.mxfp8(to_oracle_input(x), FMT[fp8], q.rowwise_usage, q.columnwise_usage)
    if q.rowwise_usage:
        assert np.array_equal(u8(out._rowwise_data), ref["out_row"])
        assert u8(out._rowwise_scale_inv)[0, 0] == ref["sc_row"][0, 0]
    if q.columnwise_usage:
        assert np.array_equal(u8(out._columnwise_data), ref["out_col"])
        assert np.array_equal(u8(out._columnwise_scale_inv)[0, :16], ref["sc_col"][0, :16])


@pytest.mark.parametrize("fp8", [E4, E5])
def test_mxfp8_quantize_shape_8192x7168(fp8):
    rows, cols = 8192, 7168
    x = uniform_input((rows, cols), torch.bfloat16, seed=6)
    q = MXFP8Quantizer(fp8)
    out = q(x)
    torch.cuda.synchronize()
    ref = oracle.mxfp8(to_oracle_input(x), FMT[fp8], True, True)
    assert np.array_equal(u8(out._rowwise_scale_inv)[:rows, :cols // 32], ref["sc_row"][:rows, :cols // 32])   # 0 mismatches
    assert np.array_equal(u8(out._columnwise_scale_inv)[:rows // 32, :cols], ref["sc_col"][:rows // 32, :cols])
    assert np.array_equal(u8(out._rowwise_data), ref["out_row"])
    assert np.array_equal(u8(out._columnwise_data), ref["out_col"])
