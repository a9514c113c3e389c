"""FP8 weight-update path on the GPU through the C ABI, against the CPU oracle
(pinned in tests/test_oracle_weight_update.py).  Integer / byte results and everything the
reference's tests compare with rtol = atol = 0 are bit-exact here:

  * nvte_mxfp8_scaling_compute_partial_amax / _partial_cast: ref tests/pytorch/test_partial_cast.py
    (its shard cases incl. 16384 x 6144, checked at full size with the test's own torch formulas),
  * nvte_multi_tensor_compute_scale_and_scale_inv_cuda / _scale_inv_e8m0_cuda:
    ref tests/pytorch/test_multi_tensor.py:337-406 (size pairs, repeat 55 -> more tensors than one
    launch table holds),
  * nvte_multi_tensor_scale_cuda: ref tests/pytorch/test_multi_tensor.py (scale + overflow flag),
  * nvte_multi_tensor_adam_fp8_cuda: bit-exact with the oracle's unfused fp32 arithmetic (the
    reference binary may fuse multiply-adds: parity with IT is the test_fused_optimizer.py
    tolerance, pinned oracle-side),
  * quantize_master_weights (ref pytorch/tensor/utils.py:74-181,
    tests/pytorch/distributed/run_cast_master_weights_to_fp8.py): sharded master weights of two
    ranks reproduce the quantization of the whole weight, byte for byte.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from transformer_engine_amd import weight_update as WU
from transformer_engine_amd.tensor import (DType, Float8CurrentScalingQuantizer, Float8Quantizer,
                                           MXFP8Quantizer)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"
applier = WU.multi_tensor_applier

SHARD_CASES = [(3, 32, 64, 31), (64 * 64 - 2, 64, 64, 1), (32768, 256, 128, 0), (131072, 768, 256, 0),
               (65536, 768, 256, 131072), (98304, 128, 768, 0), (1000, 96, 160, 777), (40000, 96, 1056, 8 * 700),
               (0, 32, 32, 5)]
NP_OF = {torch.float32: np.float32, torch.float16: np.float16}


def _np(t: torch.Tensor) -> np.ndarray:
    """Oracle view of a tensor (bf16 travels as uint16)."""
    if t.dtype == torch.bfloat16:
        return t.cpu().view(torch.int16).numpy().view(np.uint16)
    return t.cpu().numpy()


def _f32(t: torch.Tensor) -> np.ndarray:
    return t.float().cpu().numpy()


def _pad(x, p):
    return (x + p - 1) // p * p


def _scale_shapes(h, w):
    return (_pad(h, 128), _pad(w // 32, 4)), (_pad(h // 32, 4), _pad(w, 128))


# ------------------------------------------------------------------ shard amax / cast
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("n,h,w,start", SHARD_CASES)
def test_partial_amax_and_cast_against_oracle(n, h, w, start, dtype):
    g = torch.Generator(device=DEV).manual_seed(n + h + start)
    # the shard and the rest of the matrix (the scales below come from the WHOLE matrix)
    full = (torch.randn(h * w, device=DEV, generator=g) * 3).to(dtype)
    inp = full[start:start + n].clone()
    rs, cs = _scale_shapes(h, w)
    ar = torch.full(rs, 7.0, dtype=dtype, device=DEV)  # stale values: the kernel must overwrite
    ac = torch.full(cs, 7.0, dtype=dtype, device=DEV)
    WU.mxfp8_scaling_compute_partial_amax(inp, ar, ac, h, w, start)
    ref_r, ref_c = oracle.mxfp8_partial_amax(_np(inp), h, w, start)
    cbw = _pad(w // 32, 4)
    assert np.array_equal(_f32(ar)[:h, :cbw], ref_r[:h, :cbw])       # incl. zeroed column padding
    assert np.array_equal(_f32(ac)[:h // 32, :], ref_c[:h // 32, :])
    assert (ar[h:] == 7.0).all() and (ac[h // 32:] == 7.0).all()     # padded rows are not the kernel's

    if dtype != torch.bfloat16:
        return  # the e8m0 step takes bf16 amaxes (ref compute_scale.cu:94)
    far = torch.zeros(rs, dtype=dtype, device=DEV)
    fac = torch.zeros(cs, dtype=dtype, device=DEV)
    WU.mxfp8_scaling_compute_partial_amax(full, far, fac, h, w, 0)
    sr = torch.empty(rs, dtype=torch.uint8, device=DEV)
    sc = torch.empty(cs, dtype=torch.uint8, device=DEV)
    applier(WU.multi_tensor_compute_scale_inv_e8m0, None, [[far, fac], [sr, sc]])
    assert np.array_equal(sr.cpu().numpy(), oracle.e8m0_from_amax(_f32(far)))
    assert np.array_equal(sc.cpu().numpy(), oracle.e8m0_from_amax(_f32(fac)))
    out_r = torch.full((n,), 0xAA, dtype=torch.uint8, device=DEV)
    out_c = torch.full((n,), 0xAA, dtype=torch.uint8, device=DEV)
    WU.mxfp8_scaling_partial_cast(inp, out_r, out_c, sr, sc, h, w, start)
    ref_or, ref_oc = oracle.mxfp8_partial_cast(_np(inp), sr.cpu().numpy(), sc.cpu().numpy(), h, w, start)
    assert np.array_equal(out_r.cpu().numpy(), ref_or)
    assert np.array_equal(out_c.cpu().numpy(), ref_oc)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_partial_cast_equals_full_quantize(dtype):
    """Shards of one matrix, cast separately with the all-reduced scales, concatenate to exactly
    what nvte_quantize produces for the whole matrix (MXFP8 rowwise + columnwise)."""
    h, w = 384, 640
    g = torch.Generator(device=DEV).manual_seed(9)
    full = torch.randn(h, w, device=DEV, generator=g).to(torch.bfloat16)
    q = MXFP8Quantizer(DType.kFloat8E4M3)(full)
    rs, cs = _scale_shapes(h, w)
    cuts = [0, 8 * 1111, 8 * 1111 + 13, 100000, h * w]
    flat = full.view(-1)
    amax_r = torch.zeros(rs, dtype=torch.bfloat16, device=DEV)
    amax_c = torch.zeros(cs, dtype=torch.bfloat16, device=DEV)
    for a, b in zip(cuts[:-1], cuts[1:]):
        ar, ac = torch.zeros_like(amax_r), torch.zeros_like(amax_c)
        WU.mxfp8_scaling_compute_partial_amax(flat[a:b].clone(), ar, ac, h, w, a)
        amax_r, amax_c = torch.maximum(amax_r, ar), torch.maximum(amax_c, ac)   # the MAX all-reduce
    sr = torch.empty(rs, dtype=torch.uint8, device=DEV)
    sc = torch.empty(cs, dtype=torch.uint8, device=DEV)
    applier(WU.multi_tensor_compute_scale_inv_e8m0, None, [[amax_r, amax_c], [sr, sc]])
    assert torch.equal(sr[:h, :w // 32], q._rowwise_scale_inv[:h, :w // 32])
    assert torch.equal(sc[:h // 32, :w], q._columnwise_scale_inv[:h // 32, :w])
    out_r = torch.empty(h * w, dtype=torch.uint8, device=DEV)
    out_c = torch.empty(h * w, dtype=torch.uint8, device=DEV)
    for a, b in zip(cuts[:-1], cuts[1:]):
        src = flat[a:b].clone()
        WU.mxfp8_scaling_partial_cast(src if dtype == torch.bfloat16 else src.float(), out_r[a:b], out_c[a:b],
                                      sr, sc, h, w, a)
    assert torch.equal(out_r.view(h, w), q._rowwise_data)
    assert torch.equal(out_c.view(h, w), q._columnwise_data)


def test_partial_amax_and_cast_full_size_case():
    """ref test_partial_cast.py:133 run_one_case(16384 * 6144, 16384, 6144, 0), with the test's own
    torch formulas as the check (the scalar oracle would take minutes at 100 M elements)."""
    h, w = 16384, 6144
    g = torch.Generator(device=DEV).manual_seed(1234)
    inp = torch.randn(h * w, device=DEV, generator=g, dtype=torch.bfloat16)
    rs, cs = _scale_shapes(h, w)
    ar = torch.zeros(rs, dtype=torch.bfloat16, device=DEV)
    ac = torch.zeros(cs, dtype=torch.bfloat16, device=DEV)
    WU.mxfp8_scaling_compute_partial_amax(inp, ar, ac, h, w, 0)
    a = inp.abs()
    assert torch.equal(ar[:h, :w // 32], a.view(h, w // 32, 32).amax(dim=2))
    assert torch.equal(ac[:h // 32, :w], a.view(h // 32, 32, w).amax(dim=1))
    sr = torch.empty(rs, dtype=torch.uint8, device=DEV)
    sc = torch.empty(cs, dtype=torch.uint8, device=DEV)
    applier(WU.multi_tensor_compute_scale_inv_e8m0, None, [[ar, ac], [sr, sc]])
    out_r = torch.empty(h * w, dtype=torch.uint8, device=DEV)
    out_c = torch.empty(h * w, dtype=torch.uint8, device=DEV)
    WU.mxfp8_scaling_partial_cast(inp, out_r, out_c, sr, sc, h, w, 0)
    row_scale = ((254 - sr.int()) * 2 ** 23).view(torch.float32)[:h, :w // 32].contiguous()
    col_scale = ((254 - sc.int()) * 2 ** 23).view(torch.float32)[:h // 32, :w].contiguous()
    f = inp.float()
    ref = (f.view(-1, 32) * row_scale.view(-1, 1)).view(-1).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(out_r, ref)
    del ref
    ref = (f.view(h // 32, 32, w) * col_scale.view(h // 32, 1, w)).view(-1).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(out_c, ref)


def test_partial_cast_rejects_bad_shapes():
    from transformer_engine_amd._lib import NVTEError
    inp = torch.zeros(64, dtype=torch.bfloat16, device=DEV)
    rs, cs = _scale_shapes(32, 64)
    ar = torch.zeros(rs, dtype=torch.bfloat16, device=DEV)
    ac = torch.zeros(cs, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(NVTEError, match="rows must be divisible by 32"):
        WU.mxfp8_scaling_compute_partial_amax(inp, ar, ac, 33, 64, 0)
    with pytest.raises(NVTEError, match="Invalid start_offset"):
        WU.mxfp8_scaling_compute_partial_amax(inp, ar, ac, 32, 64, 32 * 64 - 10)
    with pytest.raises(NVTEError, match="Wrong dtype of amax_rowwise"):
        WU.mxfp8_scaling_compute_partial_amax(inp, ar.float(), ac, 32, 64, 0)
    with pytest.raises(NVTEError, match="Wrong padding"):
        WU.mxfp8_scaling_compute_partial_amax(inp, ar[:100], ac, 32, 64, 0)


# ------------------------------------------------------------------ multi-tensor kernels
SIZE_PAIRS = [(7777 * 77, 555 * 555), (777, 555), (555, 2048 * 32 + 1), (2048 * 32 + 1, 555), (1, 1)]


@pytest.mark.parametrize("sizes", SIZE_PAIRS)
@pytest.mark.parametrize("repeat", [1, 55])
@pytest.mark.parametrize("fmt", [oracle.E4M3, oracle.E5M2])
@pytest.mark.parametrize("pow2", [False, True])
@pytest.mark.parametrize("eps", [0.0, 100.0])
def test_multi_tensor_compute_scale_and_scale_inv(sizes, repeat, fmt, pow2, eps):
    if repeat == 55 and sizes[0] > 100000:
        sizes = (sizes[0] // 50, sizes[1] // 50)
    a = torch.randn(sizes[0], device=DEV).abs()
    b = torch.randn(sizes[1], device=DEV).abs()
    a[0], b[0] = 0.0, float("inf")
    amax = []
    for _ in range(repeat):
        amax += [a.clone(), b.clone()]
    scale = [torch.empty_like(x) for x in amax]
    sinv = [torch.empty_like(x) for x in amax]
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    applier(WU.multi_tensor_compute_scale_and_scale_inv, flag, [amax, scale, sinv], oracle.FP8_MAX[fmt], pow2, eps)
    ra = oracle.scale_and_scale_inv(_f32(a), fmt, pow2, eps)
    rb = oracle.scale_and_scale_inv(_f32(b), fmt, pow2, eps)
    for i in range(len(amax)):
        rs, ri = ra if i % 2 == 0 else rb
        assert np.array_equal(_f32(scale[i]).view(np.uint32), rs.view(np.uint32))
        assert np.array_equal(_f32(sinv[i]).view(np.uint32), ri.view(np.uint32))


@pytest.mark.parametrize("sizes", SIZE_PAIRS)
@pytest.mark.parametrize("repeat", [1, 70])
def test_multi_tensor_compute_scale_inv_e8m0(sizes, repeat):
    if repeat > 1 and sizes[0] > 100000:
        sizes = (sizes[0] // 50, sizes[1] // 50)
    a = torch.randn(sizes[0], device=DEV, dtype=torch.bfloat16).abs()
    b = (torch.randn(sizes[1], device=DEV) * 1e-3).to(torch.bfloat16).abs()
    amax = []
    for _ in range(repeat):
        amax += [a.clone(), b.clone()]
    out = [torch.empty_like(x).to(torch.uint8) for x in amax]
    applier(WU.multi_tensor_compute_scale_inv_e8m0, None, [amax, out])
    ra, rb = oracle.e8m0_from_amax(_f32(a)), oracle.e8m0_from_amax(_f32(b))
    for i in range(len(amax)):
        assert np.array_equal(out[i].cpu().numpy(), ra if i % 2 == 0 else rb)


@pytest.mark.parametrize("in_dtype,out_dtype", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                                (torch.bfloat16, torch.float32), (torch.float16, torch.float16)])
def test_multi_tensor_scale(in_dtype, out_dtype):
    sizes = [1, 33, 2048 * 32, 2048 * 32 + 5, 3 * 2048 * 32 + 1] + [17] * 130
    ins = [torch.randn(s, device=DEV).to(in_dtype) for s in sizes]
    outs = [torch.full((s,), 9.0, device=DEV, dtype=out_dtype) for s in sizes]
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    applier(WU.multi_tensor_scale, flag, [ins, outs], 0.375)
    for i, o in zip(ins, outs):
        assert torch.equal(o, (i.float() * 0.375).to(out_dtype))
    assert flag.item() == 0
    ins[3][sizes[3] - 1] = float("nan")
    applier(WU.multi_tensor_scale, flag, [ins, outs], 1.0)
    assert flag.item() == 1
    from transformer_engine_amd._lib import NVTEError
    with pytest.raises(NVTEError, match="elements, but list 0 has"):
        applier(WU.multi_tensor_scale, flag, [ins[:2], [outs[0], outs[0]]], 1.0)


@pytest.mark.parametrize("g_dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("fmt", [oracle.E4M3, oracle.E5M2])
def test_multi_tensor_adam_fp8_bit_exact_with_oracle(g_dtype, mode, fmt):
    torch.manual_seed(17)
    sizes = [1, 5, 4096, 2048 * 32 + 3, 2 * 2048 * 32] + [257] * 48   # > one table (44 tensors)
    te_fmt = DType.kFloat8E4M3 if fmt == oracle.E4M3 else DType.kFloat8E5M2
    hp = dict(lr=3e-3, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.1)
    # tensor 3 starts one element into its allocation: the unaligned (element-wise) path
    def alloc(s, dtype, off):
        return torch.zeros(s + off, device=DEV, dtype=dtype)[off:]
    offs = [0, 0, 0, 1, 0] + [0] * 48
    pm = [alloc(s, torch.float32, o).copy_(torch.randn(s)) for s, o in zip(sizes, offs)]
    m = [alloc(s, torch.float32, o) for s, o in zip(sizes, offs)]
    v = [alloc(s, torch.float32, o) for s, o in zip(sizes, offs)]
    p8 = [alloc(s, torch.uint8, o) for s, o in zip(sizes, offs)]
    scale = [torch.tensor([200.0 / (1 + i % 3)], device=DEV) for i in range(len(sizes))]
    amax = [torch.zeros(1, device=DEV) for _ in sizes]
    sinv = [torch.zeros(1, device=DEV) for _ in sizes]
    ref = [(_f32(a).copy(), np.zeros(s, np.float32), np.zeros(s, np.float32)) for a, s in zip(pm, sizes)]
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    for step in (1, 2, 3):
        g = [alloc(s, g_dtype, o).copy_(torch.randn(s)) for s, o in zip(sizes, offs)]
        for t in amax:
            t.zero_()
        applier(WU.multi_tensor_adam_fp8, flag, [g, p8, m, v, pm, scale, amax, sinv], hp["lr"], hp["beta1"],
                hp["beta2"], hp["eps"], step, mode, 1, hp["weight_decay"], te_fmt)
        for i in range(len(sizes)):
            r8, rp, rm, rv, ramax = oracle.adam_fp8(_np(g[i]), *ref[i], float(scale[i].item()), fmt, step=step,
                                                    mode=mode, bias_correction=1, **hp)
            ref[i] = (rp, rm, rv)
            assert np.array_equal(_f32(pm[i]).view(np.uint32), rp.view(np.uint32)), (step, i)
            assert np.array_equal(_f32(m[i]).view(np.uint32), rm.view(np.uint32))
            assert np.array_equal(_f32(v[i]).view(np.uint32), rv.view(np.uint32))
            assert np.array_equal(p8[i].cpu().numpy(), r8)
            assert amax[i].item() == ramax
            assert sinv[i].item() == np.float32(1.0) / np.float32(scale[i].item())


def test_multi_tensor_adam_fp8_absent_meta_and_errors():
    from transformer_engine_amd._lib import NVTEError
    n = 1000
    g, pm = torch.randn(n, device=DEV), torch.randn(n, device=DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    p8 = torch.zeros(n, dtype=torch.uint8, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ref = oracle.adam_fp8(_f32(g), _f32(pm), _f32(m), _f32(v), 1.0, oracle.E4M3, lr=1e-3, beta1=0.9, beta2=0.999,
                          eps=1e-8, step=1, mode=1, bias_correction=0, weight_decay=0.0)
    # scale / amax / scale_inv absent: scale reads as 1, nothing else is written (ref adam.cu:94-97,166-172)
    applier(WU.multi_tensor_adam_fp8, flag, [[g], [p8], [m], [v], [pm], [None], [None], [None]], 1e-3, 0.9, 0.999,
            1e-8, 1, 1, 0, 0.0, DType.kFloat8E4M3)
    assert np.array_equal(p8.cpu().numpy(), ref[0])
    assert np.array_equal(_f32(pm), ref[1])
    with pytest.raises(NVTEError, match="First moment tensor 0 has dtype"):
        applier(WU.multi_tensor_adam_fp8, flag, [[g], [p8], [m.bfloat16()], [v], [pm], [None], [None], [None]],
                1e-3, 0.9, 0.999, 1e-8, 1, 1, 0, 0.0, DType.kFloat8E4M3)
    with pytest.raises(NVTEError, match="zero-sized"):
        applier(WU.multi_tensor_scale, flag, [[g[:0]], [pm[:0]]], 1.0)


# ------------------------------------------------------------------ quantize_master_weights
def _shards(numel, world):
    """Contiguous, deliberately uneven shards of a flattened weight (None = rank holds nothing)."""
    cuts = [0] + [min(numel, (numel * (r + 1) // world) // 7 * 7 + 3) for r in range(world - 1)] + [numel]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _model_weights(recipe, shapes, dev):
    ws = []
    for i, shp in enumerate(shapes):
        if recipe == "delayed":
            q = Float8Quantizer(torch.full((1,), 40.0 + i, device=dev), torch.zeros(1, device=dev), DType.kFloat8E4M3)
        elif recipe == "current":
            q = Float8CurrentScalingQuantizer(DType.kFloat8E4M3, device=dev)
        else:
            q = MXFP8Quantizer(DType.kFloat8E4M3)
        w = q.make_empty(shp, dtype=torch.bfloat16, device=dev)
        for name in ("_data", "_transpose", "_rowwise_data", "_columnwise_data"):
            if getattr(w, name, None) is not None:
                getattr(w, name).fill_(0x55)
        ws.append(w)
    return ws


def _check_against_full_quantize(recipe, model_weights, masters_full, dev):
    """Every rank's model weights must now equal the quantization of the whole master weights."""
    for mw, full in zip(model_weights, masters_full):
        src = full.to(torch.bfloat16).to(dev)
        if recipe == "delayed":
            q = Float8Quantizer(mw._quantizer.scale.clone(), torch.zeros(1, device=dev), DType.kFloat8E4M3,
                                columnwise=False)
            ref = q(src)
            assert torch.equal(mw._data, ref._data)
            assert torch.equal(mw._scale_inv, ref._scale_inv)
            assert mw._quantizer.amax.item() == src.abs().max().item()
        elif recipe == "current":
            ref = Float8CurrentScalingQuantizer(DType.kFloat8E4M3, device=dev, columnwise=False)(src)
            assert torch.equal(mw._data, ref._data)
            assert torch.equal(mw._scale_inv, ref._scale_inv)
        else:
            ref = MXFP8Quantizer(DType.kFloat8E4M3)(src)
            h, w = src.shape
            assert torch.equal(mw._rowwise_data, ref._rowwise_data)
            assert torch.equal(mw._columnwise_data, ref._columnwise_data)
            assert torch.equal(mw._rowwise_scale_inv[:h, :w // 32], ref._rowwise_scale_inv[:h, :w // 32])
            assert torch.equal(mw._columnwise_scale_inv[:h // 32, :w], ref._columnwise_scale_inv[:h // 32, :w])


SHAPES = [(256, 384), (128, 128), (96, 1056)]


@pytest.mark.parametrize("recipe", ["delayed", "current", "mxfp8"])
def test_quantize_master_weights_single_rank(recipe):
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(21)
    masters_full = [torch.randn(*s, generator=g) * (0.5 + i) for i, s in enumerate(SHAPES)]
    mws = _model_weights(recipe, SHAPES, dev)
    masters = [m.to(dev).view(-1) for m in masters_full]
    WU.quantize_master_weights(mws, masters, [0] * len(mws), None)
    torch.cuda.synchronize()
    _check_against_full_quantize(recipe, mws, masters_full, dev)
    if recipe != "mxfp8":
        assert all(w._transpose_invalid for w in mws)
        WU.post_all_gather_processing(mws)
        assert all(torch.equal(w._transpose, w._data.t().contiguous()) for w in mws)
    with pytest.raises(ValueError, match="exceeds model_weight numel"):
        WU.quantize_master_weights(mws[:1], masters[:1], [5], None)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, recipe, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        g = torch.Generator().manual_seed(22)
        masters_full = [torch.randn(*s, generator=g) * (0.5 + i) for i, s in enumerate(SHAPES)]
        mws = _model_weights(recipe, SHAPES, dev)
        masters, offsets = [], []
        for i, full in enumerate(masters_full):
            a, b = _shards(full.numel(), world)[rank]
            if i == 1 and rank == 1:
                a, b = 0, full.numel()          # weight 1 lives entirely on rank 1 ...
            elif i == 1:
                masters.append(None); offsets.append(None)   # ... and rank 0 holds none of it
                continue
            masters.append(full.view(-1)[a:b].clone().to(dev)); offsets.append(a)
        WU.quantize_master_weights(mws, masters, offsets, None)
        # the all-gather of the updated bytes (ZeRO: every rank sends the slice it owns)
        for i, (mw, full) in enumerate(zip(mws, masters_full)):
            bufs = ([mw._data] if recipe != "mxfp8" else [mw._rowwise_data, mw._columnwise_data])
            for buf in bufs:
                flat = buf.view(-1)
                owned = torch.zeros(flat.numel(), dtype=torch.int32)
                if masters[i] is not None:
                    owned[offsets[i]:offsets[i] + masters[i].numel()] = flat[offsets[i]:offsets[i] + masters[i].numel()].cpu().int() + 1
                dist.all_reduce(owned, op=dist.ReduceOp.MAX)
                assert (owned > 0).all()
                flat.copy_((owned - 1).to(torch.uint8).to(dev))
        torch.cuda.synchronize()
        _check_against_full_quantize(recipe, mws, masters_full, dev)
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("recipe", ["delayed", "current", "mxfp8"])
def test_quantize_master_weights_two_ranks_match_full_quantize(recipe):
    world = 2
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), recipe, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)
