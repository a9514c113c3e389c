"""Tensor-parallel / sequence-parallel collectives for the FP8 Linear, on RCCL over xGMI.

Mirrors the reference's ``transformer_engine/pytorch/distributed.py``:
``gather_along_first_dim`` (:1665) / ``_all_gather_fp8`` (:986-1077) / ``_all_gather_mxfp8``
(:1524-1662), ``reduce_scatter_along_first_dim`` (:928-955), ``allreduce`` (:2051-2065), and
REPLACES ``common/comm_gemm_overlap`` (userbuffers: IPC buffers + hand-written ring kernels,
``comm_gemm_overlap.cpp:444-1390``) with plain RCCL collectives issued on PyTorch's
communication stream and consumed chunk by chunk by GEMMs on the compute stream:

  * AG + GEMM (``split_overlap_ag``, :1044):  one ``broadcast`` per source rank instead of a
    ring - the 8-GPU xGMI node is fully connected, so every shard goes straight to its 7
    readers; GEMM on shard r starts when broadcast r has landed (stream-ordered, no host
    sync), while later shards are still in flight.
  * GEMM + RS (``split_overlap_rs``, :611): the output is produced in ``tp`` row blocks; block c
    is reduced to rank c (``reduce``) while block c+1 is being multiplied.
  * bulk overlap (:444): an independent collective (dgrad reduce-scatter) runs beside a whole
    GEMM (wgrad).

The byte-moving helpers take plain tensors and a ``gemm`` callback so that the chunk
bookkeeping is testable on CPU with gloo (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

from .recipe import DelayedScaling, reduce_and_update_fp8_tensors
from .tensor import (DType, Float8CurrentScalingQuantizer, Float8Quantizer, Float8Tensor,
                     MXFP8Quantizer, MXFP8Tensor)


def get_distributed_world_size(group=None) -> int:
    if not dist.is_initialized():
        return 1
    return dist.get_world_size(group=group)


# ---------------------------------------------------------------------------------------
# plain-tensor collectives
# ---------------------------------------------------------------------------------------
def all_gather_rows(local: torch.Tensor, group=None, async_op: bool = False):
    """[m, ...] per rank -> [tp*m, ...] (rank-major).  One all_gather_into_tensor."""
    tp = get_distributed_world_size(group)
    if tp == 1:
        return local, None
    out = torch.empty((local.shape[0] * tp,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    handle = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
    return out, handle


def reduce_scatter_along_first_dim(x: torch.Tensor, group=None, async_op: bool = False):
    """[tp*m, ...] partial sums -> [m, ...] (ref pytorch/distributed.py:928-955)."""
    tp = get_distributed_world_size(group)
    if tp == 1:
        return x, None
    assert x.shape[0] % tp == 0, "first dim must divide by the tensor-parallel size"
    if dist.get_backend(group) == "gloo":
        # gloo (CPU rehearsals / single-GPU tests) has no reduce_scatter: all-reduce and slice
        # (on the host: gloo's own CUDA staging was seen to stall when two ranks share one GPU)
        full = x.detach().to("cpu", copy=True).contiguous()
        if full.dtype in (torch.bfloat16, torch.float16):
            full = full.float()
        dist.all_reduce(full, group=group)
        m = x.shape[0] // tp
        r = dist.get_rank(group)
        return full[r * m:(r + 1) * m].to(device=x.device, dtype=x.dtype).contiguous(), None
    out = torch.empty((x.shape[0] // tp,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    handle = dist.reduce_scatter_tensor(out, x.contiguous(), group=group, async_op=async_op)
    return out, handle


def allreduce(x: torch.Tensor, group=None, async_op: bool = False):
    """ref pytorch/distributed.py:2051-2065"""
    if get_distributed_world_size(group) == 1:
        return x, None
    handle = dist.all_reduce(x, group=group, async_op=async_op)
    return x, handle


def gather_and_gemm_by_shard(local: torch.Tensor, gemm: Callable[[torch.Tensor, int], None],
                             group=None, gathered: Optional[torch.Tensor] = None) -> torch.Tensor:
    """AG + GEMM overlap.  ``gemm(shard, r)`` is called once per source rank r with that rank's
    rows (a view of the gathered buffer, rows [r*m, (r+1)*m)); the call for the local shard comes
    first and needs no communication.  Returns the gathered buffer."""
    tp = get_distributed_world_size(group)
    rank = dist.get_rank(group) if tp > 1 else 0
    m = local.shape[0]
    if gathered is None:
        gathered = torch.empty((m * tp,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    shards = [gathered[r * m:(r + 1) * m] for r in range(tp)]
    shards[rank].copy_(local)
    works = []
    if tp > 1:
        for r in range(tp):       # every rank issues the same collectives in the same order
            src = dist.get_global_rank(group, r) if group is not None else r
            works.append(dist.broadcast(shards[r], src=src, group=group, async_op=True))
    gemm(shards[rank], rank)      # local rows: overlaps with the first broadcasts
    for r in range(tp):
        if r == rank:
            continue
        works[r].wait()           # stream-orders the consumer behind broadcast r
        gemm(shards[r], r)
    if tp > 1:
        works[rank].wait()
    return gathered


def gemm_and_reduce_scatter_by_block(total_rows: int, gemm: Callable[[int, int], torch.Tensor],
                                     group=None) -> torch.Tensor:
    """GEMM + RS overlap.  ``gemm(row0, rows)`` returns the partial-sum block for output rows
    [row0, row0+rows); block c is sum-reduced to rank c while block c+1 is computed.  Returns this
    rank's reduced block [total_rows/tp, ...]."""
    tp = get_distributed_world_size(group)
    rank = dist.get_rank(group) if tp > 1 else 0
    assert total_rows % tp == 0
    rows = total_rows // tp
    works, mine = [], None
    for c in range(tp):
        blk = gemm(c * rows, rows)
        if tp > 1:
            dst = dist.get_global_rank(group, c) if group is not None else c
            works.append((dist.reduce(blk, dst=dst, group=group, async_op=True), blk))
        if c == rank:
            mine = blk
    for w, _ in works:
        w.wait()
    return mine


# ---------------------------------------------------------------------------------------
# quantized gathers
# ---------------------------------------------------------------------------------------
def _all_gather_fp8(q: Float8Tensor, group=None, need_transpose: bool = False) -> Float8Tensor:
    """All-gather the ROWWISE FP8 bytes; scale_inv is identical on every rank by construction
    (delayed scaling: the amax history is max-reduced over the group; current scaling: the amax
    is all-reduced before the scale is computed) - ref pytorch/distributed.py:1056-1075."""
    data, _ = all_gather_rows(q._data.view(-1, q.shape[-1]), group)
    out = Float8Tensor(shape=(data.shape[0], q.shape[-1]), dtype=q.dtype, data=data,
                       fp8_scale_inv=q._scale_inv, fp8_dtype=q._fp8_dtype, quantizer=q._quantizer)
    if need_transpose:
        out._create_transpose()
    return out


def _all_gather_mxfp8(q: MXFP8Tensor, group=None) -> MXFP8Tensor:
    """Gather rowwise data and rowwise e8m0 scales (compact layout, padding rows stripped and
    re-added) - ref pytorch/distributed.py:1614-1659."""
    assert not q._with_gemm_swizzled_scales, "all-gather needs compact (unswizzled) scales"
    tp = get_distributed_world_size(group)
    rows = q.numel() // q.shape[-1]
    data, _ = all_gather_rows(q._rowwise_data.view(rows, q.shape[-1]), group)
    sc_local = q._rowwise_scale_inv[:rows].contiguous()
    sc, _ = all_gather_rows(sc_local, group)
    pad = (-sc.shape[0]) % 128
    if pad:
        sc = torch.cat([sc, torch.zeros(pad, sc.shape[1], dtype=sc.dtype, device=sc.device)])
    return MXFP8Tensor(shape=(rows * tp, q.shape[-1]), dtype=q.dtype, rowwise_data=data,
                       rowwise_scale_inv=sc, columnwise_data=None, columnwise_scale_inv=None,
                       fp8_dtype=q._fp8_dtype, quantizer=q._quantizer)


def gather_along_first_dim(inp, group=None, quantizer=None, need_transpose: bool = False):
    """ref pytorch/distributed.py:1665: quantize locally (rowwise only: a gathered transpose is
    not the transpose of the gather, linear.py:361-363), then all-gather bytes."""
    if isinstance(inp, torch.Tensor):
        if quantizer is None:
            out, _ = all_gather_rows(inp, group)
            return out
        quantizer.set_usage(rowwise=True, columnwise=False)
        inp = quantizer(inp)
    if isinstance(inp, Float8Tensor):
        return _all_gather_fp8(inp, group, need_transpose)
    if isinstance(inp, MXFP8Tensor):
        return _all_gather_mxfp8(inp, group)
    raise TypeError(type(inp))


# ---------------------------------------------------------------------------------------
# column-parallel + sequence-parallel Linear (the C4 pattern, ref module/linear.py:314-549,
# :760-1261 with sequence_parallel=True, parallel_mode="column")
# ---------------------------------------------------------------------------------------
class ColumnParallelLinearSP:
    """Weight shard [N/tp, K]; activations sharded on tokens.  forward: FP8 all-gather of X
    overlapped with the per-shard GEMMs; backward: dgrad partial sums reduce-scattered to the
    token shard while wgrad runs; amax all-reduce (MAX) once per fwd and once per bwd."""

    def __init__(self, in_features: int, out_features: int, tp: int, rank: int, recipe=None,
                 device=None, group=None, params_dtype=torch.bfloat16, overlap: bool = True):
        from .linear import LinearFP8State
        assert out_features % tp == 0
        self.K, self.N_local, self.tp, self.rank, self.group = in_features, out_features // tp, tp, rank, group
        self.device = device or torch.device("cuda")
        g = torch.Generator(device="cpu").manual_seed(4321 + rank)
        self.weight = (torch.randn(self.N_local, in_features, generator=g) * 0.02).to(params_dtype).to(self.device)
        self.state = LinearFP8State(recipe or DelayedScaling(), device=self.device)
        self.overlap = overlap
        self.out_dtype = params_dtype

    def forward(self, x_local: torch.Tensor, is_first_microbatch=None):
        from .gemm import general_gemm
        from .linear import FWD_INPUT, FWD_WEIGHT
        st = self.state
        qi, qw = st.fwd_q[FWD_INPUT], st.fwd_q[FWD_WEIGHT]
        if st.weight_cache is None or is_first_microbatch is None or is_first_microbatch:
            qw.set_usage(rowwise=True, columnwise=True)
            st.weight_cache = qw(self.weight)
        w8 = st.weight_cache
        qi.set_usage(rowwise=True, columnwise=False)
        x8_local = qi(x_local)
        m, tp = x_local.shape[0], self.tp
        y = torch.empty(m * tp, self.N_local, dtype=self.out_dtype, device=self.device)
        if self.overlap and tp > 1 and isinstance(x8_local, Float8Tensor):
            def shard_gemm(shard_bytes, r):
                xs = Float8Tensor(shape=(m, self.K), dtype=x_local.dtype, data=shard_bytes,
                                  fp8_scale_inv=x8_local._scale_inv, fp8_dtype=x8_local._fp8_dtype)
                general_gemm(w8, xs, layout="TN", out=y[r * m:(r + 1) * m])
            gathered = gather_and_gemm_by_shard(x8_local._data, shard_gemm, self.group)
            x8 = Float8Tensor(shape=(m * tp, self.K), dtype=x_local.dtype, data=gathered,
                              fp8_scale_inv=x8_local._scale_inv, fp8_dtype=x8_local._fp8_dtype)
        else:
            x8 = gather_along_first_dim(x8_local, self.group)
            general_gemm(w8, x8, layout="TN", out=y)
        st.update_fwd(self.group)
        return y, (x8, w8)

    def backward(self, dy: torch.Tensor, saved):
        from .gemm import general_gemm
        from .linear import BWD_GRAD_OUTPUT
        x8, w8 = saved
        st = self.state
        qg = st.bwd_q[BWD_GRAD_OUTPUT]
        qg.set_usage(rowwise=True, columnwise=True)
        dy8 = qg(dy)
        # dgrad partial sums [M_total, K] -> reduce-scatter to this rank's tokens, asynchronously
        dx_partial, *_ = general_gemm(w8, dy8, out_dtype=self.out_dtype, layout="NN", grad=True)
        dx_local, handle = reduce_scatter_along_first_dim(dx_partial, self.group, async_op=True)
        # wgrad needs X^T of the GATHERED input: rebuilt locally (ref distributed.py:1067-1075)
        if isinstance(x8, Float8Tensor) and (x8._transpose is None or x8._transpose_invalid):
            x8._create_transpose()
        dw, *_ = general_gemm(x8, dy8, out_dtype=self.out_dtype, layout="NT", grad=True)
        if handle is not None:
            handle.wait()
        st.update_bwd(self.group)
        return dx_local, dw

    def step(self, x_local, dy, is_first_microbatch=None):
        y, saved = self.forward(x_local, is_first_microbatch)
        dx, dw = self.backward(dy, saved)
        return y, dx, dw
