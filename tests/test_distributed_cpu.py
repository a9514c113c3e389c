"""world_size-2 gloo tests (CPU) of the TP/SP chunk bookkeeping in transformer_engine_amd.distributed:
the collectives move the right rows to the right ranks and the overlap schedules produce the same
result as gather-then-matmul / matmul-then-reduce-scatter."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn_name, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        globals()[fn_name](rank, world)
        ret[rank] = "ok"
    except Exception as e:  # pragma: no cover
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def _run(fn_name, world=2):
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, fn_name, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)


def _case_gather_and_gemm(rank, world):
    from transformer_engine_amd import distributed as D
    g = torch.Generator().manual_seed(7)
    full = torch.randint(0, 255, (world * 6, 8), generator=g, dtype=torch.uint8)   # "FP8 bytes"
    w = torch.randn(5, 8, generator=g)
    local = full[rank * 6:(rank + 1) * 6].clone()
    y = torch.zeros(world * 6, 5)
    order = []

    def gemm(shard, r):
        order.append(r)
        y[r * 6:(r + 1) * 6] = shard.float() @ w.t()

    gathered = D.gather_and_gemm_by_shard(local, gemm, None)
    assert torch.equal(gathered, full)                     # bit-exact all-gather of bytes
    assert order[0] == rank and sorted(order) == list(range(world))
    assert torch.equal(y, full.float() @ w.t())
    # plain all-gather agrees
    out, _ = D.all_gather_rows(local, None)
    assert torch.equal(out, full)


def _case_gemm_and_reduce_scatter(rank, world):
    from transformer_engine_amd import distributed as D
    g = torch.Generator().manual_seed(9)
    dy_all = [torch.randn(world * 4, 3, generator=g) for _ in range(world)]   # per-rank partial operands
    w_all = [torch.randn(3, 6, generator=g) for _ in range(world)]
    expect = sum(dy_all[r] @ w_all[r] for r in range(world))

    def gemm(row0, rows):
        return dy_all[rank][row0:row0 + rows] @ w_all[rank]

    mine = D.gemm_and_reduce_scatter_by_block(world * 4, gemm, None)
    assert torch.allclose(mine, expect[rank * 4:(rank + 1) * 4], atol=1e-5)


def _case_kchunked_gather_and_gemm(rank, world):
    """AG + GEMM cut along the contraction dim: every chunk is one full all-gather, the chunk GEMMs
    accumulate, the gathered chunks side by side are the plain all-gather."""
    from transformer_engine_amd import distributed as D
    g = torch.Generator().manual_seed(17)
    K = 640
    full = torch.randint(0, 255, (world * 6, K), generator=g, dtype=torch.uint8)
    w = torch.randn(5, K, generator=g)
    local = full[rank * 6:(rank + 1) * 6].clone()
    y = torch.zeros(world * 6, 5)
    seen = []

    def gemm(buf, i, k0, k1):
        assert buf.shape == (world * 6, k1 - k0)
        seen.append((i, k0, k1))
        y.add_(buf.float() @ w[:, k0:k1].t())

    bufs = D.gather_and_gemm_kchunked(local, gemm, None, chunks=3)
    assert [s[0] for s in seen] == [0, 1, 2] and seen[0][1] == 0 and seen[-1][2] == K
    assert all(k0 % 128 == 0 for _, k0, _ in seen)
    assert torch.equal(torch.cat(bufs, dim=1), full)          # bit-exact all-gather of bytes
    assert torch.allclose(y, full.float() @ w.t(), rtol=1e-5, atol=1e-2)
    assert D.split_points(384, 2) == [(0, 256), (256, 384)] and D.split_points(100, 4) == [(0, 100)]


def _case_nchunked_gemm_and_reduce_scatter(rank, world):
    from transformer_engine_amd import distributed as D
    g = torch.Generator().manual_seed(19)
    N = 384
    x_all = [torch.randn(world * 4, 7, generator=g) for _ in range(world)]
    w_all = [torch.randn(N, 7, generator=g) for _ in range(world)]
    expect = sum(x_all[r] @ w_all[r].t() for r in range(world))

    def gemm(c0, c1):
        return x_all[rank] @ w_all[rank][c0:c1].t()

    mine = D.gemm_and_reduce_scatter_nchunked(N, gemm, None, chunks=2)
    assert mine.shape == (4, N)
    assert torch.allclose(mine, expect[rank * 4:(rank + 1) * 4], atol=1e-4)


def _case_amax_reduction(rank, world):
    """delayed-scaling amax buffer: ONE MAX all-reduce of the packed buffer (ref quantization.py:666-674)."""
    buf = torch.tensor([1.0 + rank, 5.0 - rank, 0.0])
    dist.all_reduce(buf, op=dist.ReduceOp.MAX)
    assert torch.equal(buf, torch.tensor([float(world), 5.0, 0.0]))


def test_gather_and_gemm_overlap_schedule():
    _run("_case_gather_and_gemm")


def test_gemm_and_reduce_scatter_overlap_schedule():
    _run("_case_gemm_and_reduce_scatter")


def test_kchunked_gather_and_gemm_schedule():
    _run("_case_kchunked_gather_and_gemm")


def test_nchunked_gemm_and_reduce_scatter_schedule():
    _run("_case_nchunked_gemm_and_reduce_scatter")


def test_amax_max_reduction():
    _run("_case_amax_reduction")


# ---------------------------------------------------------------- FP8 weight update: the exchange step
def _case_weight_update_amax_exchange(rank, world):
    """quantize_master_weights' only collective: ONE MAX all-reduce of the packed amaxes; ranks that
    hold no shard of a weight contribute 0 (ref pytorch/tensor/utils.py:322-365)."""
    from transformer_engine_amd import weight_update as WU
    packed = torch.zeros(5)
    mine = {0: [0.5, 0.0, 3.0, 0.0, 1.0], 1: [0.25, 7.0, 2.0, 0.0, 1.5]}[rank]
    packed.copy_(torch.tensor(mine))
    WU._all_reduce_max(packed, None)
    assert torch.equal(packed, torch.tensor([0.5, 7.0, 3.0, 0.0, 1.5]))
    # bf16 block amaxes travel in the master dtype (MXFP8 path, ref :841-883)
    blk = torch.tensor([rank + 1.0, 0.0, 4.0 - rank], dtype=torch.bfloat16)
    WU._all_reduce_max(blk, None)
    assert torch.equal(blk, torch.tensor([2.0, 0.0, 4.0], dtype=torch.bfloat16))


def test_weight_update_amax_exchange():
    _run("_case_weight_update_amax_exchange")


def test_weight_update_host_checks():
    """Host logic of the shard bookkeeping (no GPU): offsets are validated before any kernel, CPU
    tensors are refused (there is no CPU fallback), list lengths must agree."""
    from transformer_engine_amd import weight_update as WU

    class W:  # stand-in for a model weight: only numel() is consulted
        def numel(self):
            return 100

    m = torch.zeros(40)
    assert WU._check_shard(W(), m, 60) == 100
    with pytest.raises(ValueError, match="must not be None"):
        WU._check_shard(W(), m, None)
    with pytest.raises(ValueError, match="non-negative"):
        WU._check_shard(W(), m, -1)
    with pytest.raises(ValueError, match="exceeds model_weight numel"):
        WU._check_shard(W(), m, 61)
    with pytest.raises(ValueError, match="same length"):
        WU._Lists([[m], [m, m]])
    with pytest.raises(RuntimeError, match="GPU only"):
        WU._Lists([[m], [m]])
    with pytest.raises(ValueError, match="not supported yet"):
        class MW:
            _quantizer = object()
            dtype = torch.float32
        WU.quantize_master_weights([MW()], [None], [None], None)
