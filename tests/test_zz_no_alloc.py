"""nvte_amd_grouped_gemm_noalloc: the grouped GEMM entry point that allocates nothing (its setup records
live in a static device array of the library) against nvte_grouped_gemm with a caller workspace.
Two calls in a row, the second on a smaller problem: the array keeps the first call's contents and the
second call must not see them.  Bit-exact (same kernel, integer-valued operands: also equal to the exact
fp64 product)."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_grouped_gpu import _dev_i64, _hp_group, _int_group
from transformer_engine_amd import NVTEError, _lib
from transformer_engine_amd._lib import api
from transformer_engine_amd.grouped import general_grouped_gemm_for_grouped_tensor
from transformer_engine_amd.tensor import DType, current_stream_ptr, wrap_plain

pytestmark = pytest.mark.gpu
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2


def _noalloc(A, B, out, layout, alpha, beta, accumulate):
    transa, transb = layout[0] == "T", layout[1] == "T"
    wa, wb, wd = A.wrap(), B.wrap(), out.wrap()
    wal, wbe = wrap_plain(alpha), wrap_plain(beta)
    cfg = api.nvte_create_grouped_matmul_config()
    try:
        api.nvte_amd_grouped_gemm_noalloc(wa.handle, int(transa), wb.handle, int(transb),
                                          wd.handle if accumulate else None, wd.handle, wal.handle, wbe.handle,
                                          cfg, current_stream_ptr())
    finally:
        _lib.raw_lib.nvte_destroy_grouped_matmul_config(cfg)
        for w in (wa, wb, wd, wal, wbe):
            w.destroy()


def _one_problem(m_splits, N, K, seed, accumulate):
    G = len(m_splits)
    sx = [0.5 ** (i % 3) for i in range(G)]
    sw = [1.0 + (i % 4) for i in range(G)]
    X, xs = _int_group([(m, K) for m in m_splits], E4, seed, sx)
    W, ws = _int_group([(N, K)] * G, E5, seed + 1, sw)
    base = torch.arange(sum(m_splits) * N, dtype=torch.float32, device="cuda") % 13
    alpha = torch.tensor([1.0 + 0.5 * (i % 2) for i in range(G)], device="cuda")
    beta = torch.tensor([0.25 * (i % 3) for i in range(G)], device="cuda")
    ref_out = _hp_group([(m, N) for m in m_splits], torch.float32, fill=base)
    general_grouped_gemm_for_grouped_tensor(W, X, ref_out, layout="TN", accumulate=accumulate, alpha=alpha, beta=beta)
    got_out = _hp_group([(m, N) for m in m_splits], torch.float32, fill=base)
    _noalloc(W, X, got_out, "TN", alpha, beta, accumulate)
    torch.cuda.synchronize()
    assert torch.equal(got_out.rowwise_data, ref_out.rowwise_data)            # bit-exact vs the existing entry point
    y = got_out.rowwise_data.cpu().numpy()
    b = base.cpu().numpy()
    r0 = 0
    for i, m in enumerate(m_splits):
        exact = alpha[i].item() * (xs[i] @ ws[i].T) * np.float32(sx[i] * sw[i])
        if accumulate:
            exact = exact + beta[i].item() * b[r0 * N:(r0 + m) * N].reshape(m, N)
        assert np.array_equal(y[r0 * N:(r0 + m) * N].reshape(m, N), exact.astype(np.float32)), f"member {i}"
        r0 += m


def test_noalloc_grouped_gemm_two_calls_second_smaller():
    api.nvte_cublas_handle_init()      # resolves the static array's address on this device (optional)
    _one_problem([512, 256, 0, 1280, 300, 768, 48], 512, 256, seed=11, accumulate=True)
    _one_problem([48, 16], 256, 128, seed=21, accumulate=False)     # fewer members, fewer tiles, no C
    _one_problem([1024], 320, 384, seed=31, accumulate=True)


def test_noalloc_grouped_gemm_rejects_too_many_members():
    limit = int(api.nvte_amd_grouped_gemm_noalloc_max_tensors())
    assert limit == 256
    G = limit + 1
    K, N = 32, 32
    X, _ = _int_group([(16, K)] * G, E4, 5, [1.0] * G)
    W, _ = _int_group([(N, K)] * G, E4, 6, [1.0] * G)
    out = _hp_group([(16, N)] * G, torch.float32)
    one = torch.ones(G, device="cuda")
    with pytest.raises(NVTEError, match="exceed the static scratch limit"):
        _noalloc(W, X, out, "TN", one, one, False)
    # exactly the limit is accepted
    G = limit
    X, xs = _int_group([(16, K)] * G, E4, 7, [1.0] * G)
    W, ws = _int_group([(N, K)] * G, E4, 8, [1.0] * G)
    out = _hp_group([(16, N)] * G, torch.float32)
    one, zero = torch.ones(G, device="cuda"), torch.zeros(G, device="cuda")
    _noalloc(W, X, out, "TN", one, zero, False)
    torch.cuda.synchronize()
    y = out.rowwise_data.cpu().numpy().reshape(G, 16, N)
    for i in (0, 1, G // 2, G - 1):
        assert np.array_equal(y[i], (xs[i] @ ws[i].T).astype(np.float32))
