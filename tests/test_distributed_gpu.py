"""Two ranks sharing the one GPU of the test box (gloo moves the bytes; RCCL refuses two ranks on
one device): the column-parallel + sequence-parallel FP8 Linear must reproduce the un-sharded
FP8 Linear.  Pattern follows the reference's tests/pytorch/distributed/run_numerics.py: build the
single-GPU result, build the sharded one from the same parameters, compare - quantized all-gather
bit-exact (:511-610), outputs within FP8 GEMM tolerance."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, overlap, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import transformer_engine_amd as te
        from transformer_engine_amd import distributed as D, linear as L
        from transformer_engine_amd.recipe import DelayedScaling
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        M_local, K, N = 256, 384, 512
        g = torch.Generator().manual_seed(5)
        x_full = torch.randn(world * M_local, K, generator=g).to(torch.bfloat16).to(dev)
        dy_full = (torch.randn(world * M_local, N, generator=g) * 0.01).to(torch.bfloat16).to(dev)
        mod = D.ColumnParallelLinearSP(K, N, world, rank, DelayedScaling(amax_history_len=4), device=dev, overlap=overlap)
        # all ranks need the full weight for the un-sharded reference: gather the shards
        w_full, _ = D.all_gather_rows(mod.weight, None)
        x_local = x_full[rank * M_local:(rank + 1) * M_local].contiguous()
        dy_local_cols = dy_full[:, rank * (N // world):(rank + 1) * (N // world)].contiguous()
        for it in range(3):      # step 0 runs with scale 1, later steps with history-derived scales
            y, dx, dw = mod.step(x_local, dy_local_cols, is_first_microbatch=True)
        torch.cuda.synchronize()
        # un-sharded reference with the SAME scales the sharded run used in its last step
        ref = L.LinearFP8State(DelayedScaling(amax_history_len=4), device=dev)
        for it in range(3):
            ry, saved = L.linear_forward(x_full, w_full, None, ref, is_first_microbatch=True)
            ref.update_fwd()
            rdx, rdw, _ = L.linear_backward(dy_full, saved, ref)
            ref.update_bwd()
        torch.cuda.synchronize()
        # the input scale is the same (amax MAX-reduced over the group == amax of the full tensor)
        assert torch.equal(mod.state.fwd_state.scale[0], ref.fwd_state.scale[0])
        assert torch.equal(mod.state.fwd_state.amax_history[:, 0], ref.fwd_state.amax_history[:, 0])
        ncol = N // world
        # weight / grad scales are per shard in the TP run, so compare with FP8 tolerances
        # (ref run_numerics.py:193-203: fp8 rtol .125 atol .0625)
        def close(a, b, name):
            a, b = a.float(), b.float()
            tol = 0.0625 * b.abs().max().item() + 1e-6
            assert torch.allclose(a, b, rtol=0.125, atol=tol), (name, (a - b).abs().max().item(), tol)
        close(y, ry[:, rank * ncol:(rank + 1) * ncol], "y")
        close(dw, rdw[rank * ncol:(rank + 1) * ncol], "dw")
        close(dx, rdx[rank * M_local:(rank + 1) * M_local], "dx")
        # quantized all-gather is bit-exact (same scale): gathered bytes == quantize(full) bytes
        q = mod.state.fwd_q[L.FWD_INPUT]
        q.set_usage(rowwise=True, columnwise=False)
        x8_full = q(x_full)
        x8_g = D.gather_along_first_dim(x_local, None, quantizer=q)
        assert torch.equal(x8_g._data, x8_full._data)
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_column_parallel_sp_matches_unsharded(overlap):
    world = 2
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), overlap, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)
