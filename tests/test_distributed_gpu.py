"""Two ranks sharing the one GPU of the test box (gloo moves the bytes; RCCL refuses two ranks on
one device): the tensor-parallel + sequence-parallel FP8 layers must reproduce the un-sharded
FP8 layers.  Pattern follows the reference's tests/pytorch/distributed/run_numerics.py: build the
single-GPU result, build the sharded one from the same parameters, compare - quantized all-gather
bit-exact (:511-610, rtol = atol = 0), outputs within the FP8 tolerances of :193-203 (fp8 rtol .125
atol .0625 of the tensor's absmax; current scaling rtol .4 atol .25 there, the tighter pair is used
for every recipe here)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


LOOSE = False      # set per worker: True for current scaling, see _close


def _close(a, b, name):
    """rtol .125, atol .0625 of the tensor's absmax (ref run_numerics.py:193-203 fp8 pair).  Current
    scaling: the operands that stay sharded (dY columns / X features) are quantized with the SHARD's amax,
    the un-sharded run with the full tensor's - two different FP8 roundings of the same values, so the
    reference's own wider current-scaling pair (rtol .4) with atol .125 of absmax applies."""
    a, b = a.float(), b.float()
    rtol, frac = (0.4, 0.125) if LOOSE else (0.125, 0.0625)
    tol = frac * b.abs().max().item() + 1e-6
    assert torch.allclose(a, b, rtol=rtol, atol=tol), (name, (a - b).abs().max().item(), tol)


def _recipe(name):
    from transformer_engine_amd.recipe import DelayedScaling, Float8CurrentScaling, MXFP8BlockScaling
    return {"delayed": DelayedScaling(amax_history_len=4), "current": Float8CurrentScaling(),
            "mxfp8": MXFP8BlockScaling()}[name]


def _case_column(rank, world, dev, recipe_name, overlap):
    from transformer_engine_amd import distributed as D, linear as L
    M_local, K, N = 256, 384, 512
    g = torch.Generator().manual_seed(5)
    x_full = torch.randn(world * M_local, K, generator=g).to(torch.bfloat16).to(dev)
    x_full[:M_local] *= 3.0            # rank 0's tokens carry the global amax: local scales would differ
    dy_full = (torch.randn(world * M_local, N, generator=g) * 0.01).to(torch.bfloat16).to(dev)
    mod = D.ColumnParallelLinearSP(K, N, world, rank, _recipe(recipe_name), device=dev, overlap=overlap)
    w_full, _ = D.all_gather_rows(mod.weight, None)
    x_local = x_full[rank * M_local:(rank + 1) * M_local].contiguous()
    ncol = N // world
    dy_local_cols = dy_full[:, rank * ncol:(rank + 1) * ncol].contiguous()
    for it in range(3):      # delayed scaling: step 0 runs with scale 1, later steps with history-derived scales
        y, dx, dw = mod.step(x_local, dy_local_cols, is_first_microbatch=True)
    torch.cuda.synchronize()
    ref = L.LinearFP8State(_recipe(recipe_name), device=dev)
    for it in range(3):
        ry, saved = L.linear_forward(x_full, w_full, None, ref, is_first_microbatch=True)
        ref.update_fwd()
        rdx, rdw, _ = L.linear_backward(dy_full, saved, ref)
        ref.update_bwd()
    torch.cuda.synchronize()
    if recipe_name == "delayed":
        # the input scale is the same (amax MAX-reduced over the group == amax of the full tensor)
        assert torch.equal(mod.state.fwd_state.scale[0], ref.fwd_state.scale[0])
        assert torch.equal(mod.state.fwd_state.amax_history[:, 0], ref.fwd_state.amax_history[:, 0])
    _close(y, ry[:, rank * ncol:(rank + 1) * ncol], "y")
    _close(dw, rdw[rank * ncol:(rank + 1) * ncol], "dw")
    _close(dx, rdx[rank * M_local:(rank + 1) * M_local], "dx")
    # quantized all-gather is bit-exact: gathered bytes == quantize(full) bytes, for every recipe
    q = mod.state.fwd_q[L.FWD_INPUT]
    x8_g = D.gather_along_first_dim(x_local, None, quantizer=q)
    if recipe_name == "mxfp8":
        from transformer_engine_amd.tensor import MXFP8Quantizer
        assert not q.columnwise_transposed        # the layers gather the reference layout (plain row gathers)
        nb = world * M_local // 32
        for tcol in (False, True):                # ... and the helper takes both
            ql = MXFP8Quantizer(q.dtype, columnwise_transposed=tcol)
            x8_g = D.gather_along_first_dim(x_local, None, quantizer=ql)
            x8_full = MXFP8Quantizer(q.dtype, columnwise_transposed=tcol)(x_full)
            assert x8_g._columnwise_transposed == tcol
            assert torch.equal(x8_g._rowwise_data, x8_full._rowwise_data)
            assert torch.equal(x8_g._rowwise_scale_inv[:world * M_local], x8_full._rowwise_scale_inv[:world * M_local])
            assert torch.equal(x8_g._columnwise_data.view(-1), x8_full._columnwise_data.view(-1))
            if tcol:
                assert torch.equal(x8_g._columnwise_scale_inv[:K, :nb], x8_full._columnwise_scale_inv[:K, :nb])
            else:
                assert torch.equal(x8_g._columnwise_scale_inv[:nb, :K], x8_full._columnwise_scale_inv[:nb, :K])
    else:
        if recipe_name == "current":
            from transformer_engine_amd.tensor import Float8CurrentScalingQuantizer
            qf = Float8CurrentScalingQuantizer(q.dtype, dev, columnwise=False)      # no reduction: sees the full tensor
        else:
            qf = q
            qf.set_usage(rowwise=True, columnwise=False)
        x8_full = qf(x_full)
        assert torch.equal(x8_g._scale_inv, x8_full._scale_inv)
        assert torch.equal(x8_g._data, x8_full._data)


def _case_row(rank, world, dev, recipe_name, overlap):
    from transformer_engine_amd import distributed as D, linear as L
    M_local, K, N = 256, 512, 384
    Kl = K // world
    g = torch.Generator().manual_seed(6)
    x_full = torch.randn(world * M_local, K, generator=g).to(torch.bfloat16).to(dev)
    dy_full = (torch.randn(world * M_local, N, generator=g) * 0.01).to(torch.bfloat16).to(dev)
    dy_full[:M_local] *= 5.0           # rank 0's tokens carry the global amax of dY
    mod = D.RowParallelLinearSP(K, N, world, rank, _recipe(recipe_name), device=dev, overlap=overlap)
    # full weight [N, K] = the shards side by side along K
    wt_full, _ = D.all_gather_rows(mod.weight.t().contiguous(), None)        # [K, N]
    w_full = wt_full.t().contiguous()
    x_cols = x_full[:, rank * Kl:(rank + 1) * Kl].contiguous()               # all tokens, local features
    dy_local = dy_full[rank * M_local:(rank + 1) * M_local].contiguous()
    for it in range(3):
        y, dx, dw = mod.step(x_cols, dy_local, is_first_microbatch=True)
    torch.cuda.synchronize()
    ref = L.LinearFP8State(_recipe(recipe_name), device=dev)
    for it in range(3):
        ry, saved = L.linear_forward(x_full, w_full, None, ref, is_first_microbatch=True)
        ref.update_fwd()
        rdx, rdw, _ = L.linear_backward(dy_full, saved, ref)
        ref.update_bwd()
    torch.cuda.synchronize()
    _close(y, ry[rank * M_local:(rank + 1) * M_local], "y")                   # reduce-scattered output rows
    _close(dx, rdx[:, rank * Kl:(rank + 1) * Kl], "dx")
    _close(dw, rdw[:, rank * Kl:(rank + 1) * Kl], "dw")
    if recipe_name != "mxfp8":
        # the FP8 all-gather of dY is bit-exact against quantize(full dY) (one scale over the group)
        qg = mod.state.bwd_q[L.BWD_GRAD_OUTPUT]
        dy8_g = D.gather_along_first_dim(dy_local, None, quantizer=qg)
        if recipe_name == "current":
            from transformer_engine_amd.tensor import Float8CurrentScalingQuantizer
            qf = Float8CurrentScalingQuantizer(qg.dtype, dev, columnwise=False)
        else:
            qf = qg
            qf.set_usage(rowwise=True, columnwise=False)
        dy8_full = qf(dy_full)
        assert torch.equal(dy8_g._scale_inv, dy8_full._scale_inv)
        assert torch.equal(dy8_g._data, dy8_full._data)


def _case_lnlinear_c4(rank, world, dev, recipe_name, overlap):
    """BASELINE configs[3] at its per-rank shapes (Llama-3-70B QKV projection: hidden 8192 -> 10240, tokens
    1024 per rank), run at tp = 2: every rank holds 5120 output columns instead of 1280."""
    _case_lnlinear(rank, world, dev, recipe_name, overlap, dims=(1024, 8192, 10240), norms=("RMSNorm",))


def _case_lnlinear(rank, world, dev, recipe_name, overlap, dims=(256, 512, 768), norms=("RMSNorm", "LayerNorm")):
    """BASELINE configs[3] shape class: LayerNormLinear, column-parallel + sequence-parallel."""
    from transformer_engine_amd import distributed as D, linear as L
    M_local, K, N = dims
    g = torch.Generator().manual_seed(7)
    x_full = torch.randn(world * M_local, K, generator=g).to(torch.bfloat16).to(dev)
    dy_full = (torch.randn(world * M_local, N, generator=g) * 0.01).to(torch.bfloat16).to(dev)
    gamma = (torch.randn(K, generator=g) * 0.2 + 1.0).to(torch.bfloat16).to(dev)
    for norm in norms:
        mod = D.ColumnParallelLayerNormLinearSP(K, N, world, rank, _recipe(recipe_name), device=dev, overlap=overlap,
                                                normalization=norm, allreduce_norm_grads=True)
        mod.layer_norm_weight.copy_(gamma)
        w_full, _ = D.all_gather_rows(mod.weight, None)
        x_local = x_full[rank * M_local:(rank + 1) * M_local].contiguous()
        ncol = N // world
        dy_cols = dy_full[:, rank * ncol:(rank + 1) * ncol].contiguous()
        for it in range(3):
            y, dx, dw, dgamma, dbeta = mod.step(x_local, dy_cols, is_first_microbatch=True)
        torch.cuda.synchronize()
        ref = L.LinearFP8State(_recipe(recipe_name), device=dev)
        beta = torch.zeros(K, dtype=torch.bfloat16, device=dev) if norm == "LayerNorm" else None
        for it in range(3):
            ry, saved = L.layernorm_linear_forward(x_full, gamma, beta, w_full, None, ref, normalization=norm,
                                                   is_first_microbatch=True)
            ref.update_fwd()
            rdx, rdgamma, rdbeta, rdw, _ = L.layernorm_linear_backward(dy_full, saved, gamma, ref, normalization=norm)
            ref.update_bwd()
        torch.cuda.synchronize()
        _close(y, ry[:, rank * ncol:(rank + 1) * ncol], norm + " y")
        _close(dw, rdw[rank * ncol:(rank + 1) * ncol], norm + " dw")
        # dx / dgamma of the sharded run only hold THIS rank's N/tp output columns' contribution in the
        # un-sharded reference terms?  No: dgrad is reduce-scattered over the TP group, i.e. summed over all
        # output columns, so the local rows of dx equal the reference's rows.
        _close(dx, rdx[rank * M_local:(rank + 1) * M_local], norm + " dx")
        _close(dgamma, rdgamma, norm + " dgamma")          # all-reduced over the token shards
        if norm == "LayerNorm":
            _close(dbeta, rdbeta, norm + " dbeta")


def _worker(rank, world, port, case, recipe_name, overlap, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        globals()["LOOSE"] = recipe_name == "current"
        globals()["_case_" + case](rank, world, torch.device("cuda:0"), recipe_name, overlap)
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def _run(case, recipe_name, overlap):
    world = 2
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), case, recipe_name, overlap, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("recipe_name", ["delayed", "current", "mxfp8"])
def test_column_parallel_sp_matches_unsharded(recipe_name, overlap):
    _run("column", recipe_name, overlap)


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("recipe_name", ["delayed", "current", "mxfp8"])
def test_row_parallel_sp_matches_unsharded(recipe_name, overlap):
    _run("row", recipe_name, overlap)


@pytest.mark.parametrize("recipe_name", ["delayed", "current"])
def test_layernorm_linear_tp_sp_matches_unsharded(recipe_name):
    _run("lnlinear", recipe_name, True)


def test_layernorm_linear_tp_sp_configs3_shapes():
    _run("lnlinear_c4", "delayed", True)
