"""The RCCL branch of every tensor-parallel helper, executed on the ONE GPU a test box has.

A fresh child process starts a world-size-1 ``nccl`` process group (backend "nccl" IS RCCL on ROCm) before it
touches the GPU and switches ``distributed.force_collectives`` on, so the helpers do not early-return for a
single rank: ``all_gather_into_tensor``, ``reduce_scatter_tensor``, ``all_reduce(MAX)`` and the ``work.wait()``
stream ordering run on RCCL's communication stream beside the persistent GEMMs.  (The 2-rank tests of
test_distributed_gpu.py use gloo, whose branch is host-staged and synchronous.)  With one rank the sharded layer
IS the un-sharded layer, so the results must be bit-equal without the chunked overlap, and equal to within the
fp32 rounding of the per-chunk partial sums with it.  Pattern: ref tests/pytorch/distributed/run_numerics.py:193-211,
:511-610; what it replaces: ref common/comm_gemm_overlap/comm_gemm_overlap.cpp:1044-1216.
"""
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


def _recipe(name):
    from transformer_engine_amd.recipe import DelayedScaling, Float8CurrentScaling, MXFP8BlockScaling
    return {"delayed": DelayedScaling(amax_history_len=4), "current": Float8CurrentScaling(),
            "mxfp8": MXFP8BlockScaling()}[name]


def _counted(fn_name, counts):
    """Count calls of a torch.distributed collective (proof that the RCCL branch ran)."""
    orig = getattr(dist, fn_name)

    def wrapper(*a, **k):
        counts[fn_name] = counts.get(fn_name, 0) + 1
        return orig(*a, **k)
    setattr(dist, fn_name, wrapper)


def _near(a, b, name):
    """overlap=True: the chunk GEMMs accumulate in fp32 and are rounded once; against the unsplit GEMM that is
    at most one rounding step of the output dtype (bf16: 2^-8 relative) on top of fp32 noise."""
    a, b = a.float(), b.float()
    tol = 2.0 ** -7
    assert torch.allclose(a, b, rtol=tol, atol=tol * b.abs().max().item() * 2.0 ** -4), \
        (name, (a - b).abs().max().item(), b.abs().max().item())


def _child(rank, port, recipe_name, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        from transformer_engine_amd import distributed as D, linear as L
        D.force_collectives(True)
        assert dist.get_backend() == "nccl" and D.collectives_active(None)
        counts = {}
        for fn in ("all_gather_into_tensor", "reduce_scatter_tensor", "all_reduce"):
            _counted(fn, counts)
        g = torch.Generator().manual_seed(5)
        M, K, N = 512, 768, 1024
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        dy = (torch.randn(M, N, generator=g) * 0.01).to(torch.bfloat16).to(dev)
        gamma = (torch.randn(K, generator=g) * 0.2 + 1.0).to(torch.bfloat16).to(dev)

        def run_ref(w, ln=None):
            ref = L.LinearFP8State(_recipe(recipe_name), device=dev)
            for it in range(3):
                if ln is None:
                    ry, saved = L.linear_forward(x, w, None, ref, is_first_microbatch=True)
                    ref.update_fwd()
                    rdx, rdw, _ = L.linear_backward(dy, saved, ref)
                    extra = ()
                else:
                    ry, saved = L.layernorm_linear_forward(x, gamma, None, w, None, ref, normalization=ln,
                                                           is_first_microbatch=True)
                    ref.update_fwd()
                    rdx, rdgamma, _, rdw, _ = L.layernorm_linear_backward(dy, saved, gamma, ref, normalization=ln)
                    extra = (rdgamma,)
                ref.update_bwd()
            return (ry, rdx, rdw) + extra

        for overlap in (False, True):
            before = dict(counts)
            # column-parallel + sequence-parallel: AG (+ K-chunked GEMMs), dgrad RS beside wgrad, amax MAX-AR
            col = D.ColumnParallelLinearSP(K, N, 1, 0, _recipe(recipe_name), device=dev, overlap=overlap, chunks=3)
            for it in range(3):
                out = col.step(x, dy, is_first_microbatch=True)
            # row-parallel: GEMM (+ N-chunked) RS, FP8 AG of dY
            row = D.RowParallelLinearSP(K, N, 1, 0, _recipe(recipe_name), device=dev, overlap=overlap, chunks=3)
            for it in range(3):
                out_r = row.step(x, dy, is_first_microbatch=True)
            outs = [("column", out, run_ref(col.weight)), ("row", out_r, run_ref(row.weight))]
            if recipe_name != "current":
                lnl = D.ColumnParallelLayerNormLinearSP(K, N, 1, 0, _recipe(recipe_name), device=dev, overlap=overlap,
                                                        chunks=3, normalization="RMSNorm", allreduce_norm_grads=True)
                lnl.layer_norm_weight.copy_(gamma)
                for it in range(3):
                    y, dx, dw, dgamma, _ = lnl.step(x, dy, is_first_microbatch=True)
                outs.append(("ln_column", (y, dx, dw, dgamma), run_ref(lnl.weight, "RMSNorm")))
            torch.cuda.synchronize()
            for fn in ("all_gather_into_tensor", "reduce_scatter_tensor", "all_reduce"):
                assert counts.get(fn, 0) > before.get(fn, 0), f"{fn} was never called (overlap={overlap})"
            for layer, got, ref in outs:
                for name, a, b in zip(("y", "dx", "dw", "dgamma"), got, ref):
                    if overlap and layer != "row" and name == "y":
                        _near(a, b, f"{layer}.{name} overlap")      # K-chunked: fp32 partial sums, one rounding
                    else:
                        # everything else is the unsplit computation (row-parallel chunks cut the OUTPUT columns:
                        # every piece holds the unsplit GEMM's own values)
                        assert torch.equal(a, b), (layer, name, overlap, (a.float() - b.float()).abs().max().item())
        ret[0] = "ok " + str(counts)
    except Exception:
        import traceback
        ret[0] = traceback.format_exc()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("recipe_name", ["delayed", "current", "mxfp8"])
def test_tp_sp_layers_on_rccl_world_size_one(recipe_name):
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_child, args=(_free_port(), recipe_name, ret), nprocs=1, join=True)
    assert str(ret.get(0, "")).startswith("ok"), ret.get(0)
