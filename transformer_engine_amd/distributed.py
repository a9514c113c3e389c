"""Tensor-parallel / sequence-parallel collectives for the FP8 Linear, on RCCL over xGMI.

Mirrors the reference's ``transformer_engine/pytorch/distributed.py``:
``gather_along_first_dim`` (:1665) / ``_all_gather_fp8`` (:986-1077) / ``_all_gather_mxfp8``
(:1524-1662), ``reduce_scatter_along_first_dim`` (:928-955), ``allreduce`` (:2051-2065), and
REPLACES ``common/comm_gemm_overlap`` (userbuffers: IPC buffers + hand-written ring kernels,
``comm_gemm_overlap.cpp:444-1390``) with plain RCCL collectives issued on PyTorch's
communication stream and consumed chunk by chunk by GEMMs on the compute stream.

How the overlap is cut (round 2; the per-source-rank ``broadcast`` pipeline of round 1 is kept only as
``gather_and_gemm_by_shard``):

  * AG + GEMM (``split_overlap_ag``, :1044).  The CONTRACTION dimension is cut, not the rows: chunk i
    is ``all_gather_into_tensor`` of the local FP8 bytes ``[m, K_i]`` - ONE full-width collective that
    uses all seven xGMI links of every GPU at once (a per-source broadcast or a ring is bound by one
    ≈153 GB/s link) - and lands as ``[tp*m, K_i]`` in natural token order.  GEMM i multiplies it with
    the matching K slice of the (cached, pre-sliced) FP8 weight and ACCUMULATES into the output, so it
    is a full-size persistent launch (all 256 CUs busy) that waits for chunk i only.  The transposes
    of the gathered chunks are the row blocks of X^T that wgrad needs: nothing is re-gathered.
  * GEMM + RS (``split_overlap_rs``, :611).  The OUTPUT columns are cut: GEMM j produces
    ``[M, N_j]`` with a row slice of the FP8 weight (contiguous, no copy) and its
    ``reduce_scatter_tensor`` runs while GEMM j+1 computes; the pieces are concatenated.
  * bulk overlap (:444): an independent collective (dgrad reduce-scatter) runs beside a whole
    GEMM (wgrad).

Every collective is issued with ``async_op=True`` and joined with ``work.wait()``: on RCCL that
stream-orders the consumer behind the collective without a host sync.  The byte-moving helpers take
plain tensors and callbacks so that the chunk bookkeeping is testable on CPU with gloo
(tests/test_distributed_cpu.py).  No multi-GPU hardware run exists yet (the build box has one GPU);
the 2-rank tests use gloo with both ranks on one device.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _dist_flags as _flags
from .recipe import DelayedScaling, Float8CurrentScaling, MXFP8BlockScaling, reduce_and_update_fp8_tensors
from .tensor import (DType, Float8CurrentScalingQuantizer, Float8Quantizer, Float8Tensor,
                     MXFP8Quantizer, MXFP8Tensor)


def get_distributed_world_size(group=None) -> int:
    if not dist.is_initialized():
        return 1
    return dist.get_world_size(group=group)


def _is_gloo(group) -> bool:
    return dist.is_initialized() and dist.get_backend(group) == "gloo"


def force_collectives(flag: bool = True) -> None:
    """Issue every collective even in a group of ONE rank (default: a single rank returns its input).  With a
    world-size-1 ``nccl`` group this runs the RCCL branch of every helper below - all_gather_into_tensor,
    reduce_scatter_tensor, all_reduce(MAX), ``work.wait()`` stream ordering beside the persistent GEMMs - on a
    one-GPU box (tests/test_rccl_single_gpu.py).  Also set by NVTE_AMD_FORCE_COLLECTIVES=1."""
    _flags.FORCE_COLLECTIVES = bool(flag)


def collectives_active(group=None) -> bool:
    """True when the helpers of this module communicate: more than one rank, or forced (see above)."""
    return _flags.collectives_active(group)


# ---------------------------------------------------------------------------------------
# plain-tensor collectives
# ---------------------------------------------------------------------------------------
class _Done:
    """Stand-in for a finished collective (gloo rehearsals run synchronously on the host)."""

    def wait(self):
        return True


def all_gather_rows(local: torch.Tensor, group=None, async_op: bool = False):
    """[m, ...] per rank -> [tp*m, ...] (rank-major).  One all_gather_into_tensor."""
    tp = get_distributed_world_size(group)
    if not collectives_active(group):
        return local, None
    if _is_gloo(group) and local.is_cuda:
        # rehearsal path (two ranks on one GPU): staged through the host and finished on return; gloo's
        # own device staging was seen to stall there
        src = local.detach().contiguous().view(torch.uint8).cpu() if local.dtype == torch.bfloat16 else local.detach().contiguous().cpu()
        gathered = torch.empty((src.shape[0] * tp,) + tuple(src.shape[1:]), dtype=src.dtype)
        dist.all_gather_into_tensor(gathered, src, group=group)
        out = gathered.to(local.device)
        if local.dtype == torch.bfloat16:
            out = out.view(torch.bfloat16)
        return out, (_Done() if async_op else None)
    out = torch.empty((local.shape[0] * tp,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    handle = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
    return out, handle


def reduce_scatter_along_first_dim(x: torch.Tensor, group=None, async_op: bool = False):
    """[tp*m, ...] partial sums -> [m, ...] (ref pytorch/distributed.py:928-955)."""
    tp = get_distributed_world_size(group)
    if not collectives_active(group):
        return x, None
    assert x.shape[0] % tp == 0, "first dim must divide by the tensor-parallel size"
    if _is_gloo(group):
        # gloo (CPU rehearsals / single-GPU tests) has no reduce_scatter: all-reduce and slice
        # (on the host: gloo's own CUDA staging was seen to stall when two ranks share one GPU)
        full = x.detach().to("cpu", copy=True).contiguous()
        if full.dtype in (torch.bfloat16, torch.float16):
            full = full.float()
        dist.all_reduce(full, group=group)
        m = x.shape[0] // tp
        r = dist.get_rank(group)
        return full[r * m:(r + 1) * m].to(device=x.device, dtype=x.dtype).contiguous(), (_Done() if async_op else None)
    out = torch.empty((x.shape[0] // tp,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    handle = dist.reduce_scatter_tensor(out, x.contiguous(), group=group, async_op=async_op)
    return out, handle


def allreduce(x: torch.Tensor, group=None, async_op: bool = False):
    """ref pytorch/distributed.py:2051-2065"""
    if not collectives_active(group):
        return x, None
    handle = dist.all_reduce(x, group=group, async_op=async_op)
    return x, handle


def all_reduce_max_(t: torch.Tensor, group=None) -> None:
    """In-place MAX all-reduce of an amax buffer (gloo rehearsals reduce on the host)."""
    if not collectives_active(group):
        return
    if _is_gloo(group) and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)


def split_points(total: int, chunks: int, align: int = 128) -> List[Tuple[int, int]]:
    """Cut [0, total) into at most `chunks` pieces whose boundaries are multiples of `align`
    (FP8 GEMM operands need 16-byte rows, MXFP8 whole 32-element blocks, the GEMM K tile is 128)."""
    units = max(total // align, 1)
    chunks = max(1, min(chunks, units))
    base, extra = divmod(units, chunks)
    pts, start = [], 0
    for i in range(chunks):
        n = (base + (1 if i < extra else 0)) * align
        end = total if i == chunks - 1 else start + n
        pts.append((start, end))
        start = end
    return pts


def gather_and_gemm_kchunked(local: torch.Tensor, gemm: Callable[[torch.Tensor, int, int, int], None],
                             group=None, chunks: int = 2, align: int = 128) -> List[torch.Tensor]:
    """AG + GEMM overlap, cut along the contraction dimension.

    ``local`` is this rank's ``[m, K]`` byte matrix.  For chunk i = columns [k0, k1) the gathered
    ``[tp*m, k1-k0]`` buffer (natural row order) is handed to ``gemm(buf, i, k0, k1)`` as soon as ITS
    all-gather has landed; all collectives are issued up front (they run back to back on the
    communication stream, each using every link).  Returns the gathered chunks."""
    active = collectives_active(group)
    m, K = local.shape
    pts = split_points(K, chunks if active else 1, align)
    bufs, works = [], []
    for k0, k1 in pts:
        # the collective's input must be contiguous: one strided copy of the LOCAL shard's chunk (1/tp of the
        # gathered bytes; the gathered chunks themselves land contiguous and are never copied)
        piece = local if (k0 == 0 and k1 == K) else local[:, k0:k1].contiguous()
        if not active:
            bufs.append(piece); works.append(None)
            continue
        out, work = all_gather_rows(piece, group, async_op=True)
        works.append(work)
        bufs.append(out)
    for i, (k0, k1) in enumerate(pts):
        if works[i] is not None:
            works[i].wait()              # stream-orders GEMM i behind all-gather i only
        gemm(bufs[i], i, k0, k1)
    return bufs


def gemm_and_reduce_scatter_nchunked(n_cols: int, gemm: Callable[[int, int], torch.Tensor], group=None,
                                     chunks: int = 2, align: int = 128) -> torch.Tensor:
    """GEMM + RS overlap, cut along the output columns.  ``gemm(c0, c1)`` returns the partial sums
    ``[M, c1-c0]`` (contiguous); its reduce-scatter runs while the next piece is computed.  Returns
    this rank's ``[M/tp, n_cols]``."""
    pts = split_points(n_cols, chunks if collectives_active(group) else 1, align)
    outs, works = [], []
    for c0, c1 in pts:
        part = gemm(c0, c1)
        o, w = reduce_scatter_along_first_dim(part, group, async_op=True)
        outs.append(o); works.append(w)
    for w in works:
        if w is not None:
            w.wait()
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)


def gather_and_gemm_by_shard(local: torch.Tensor, gemm: Callable[[torch.Tensor, int], None],
                             group=None, gathered: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Round-1 schedule, kept for comparison: one ``broadcast`` per source rank, GEMM on shard r behind
    broadcast r.  tp collectives that each use one link; superseded by gather_and_gemm_kchunked."""
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
    """Round-1 schedule, kept for comparison: GEMM per destination row block + ``reduce`` to that rank."""
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
    """All-gather the ROWWISE FP8 bytes; scale_inv must be identical on every rank (delayed scaling: the
    amax history is max-reduced over the group; current scaling: the quantizer all-reduces the amax
    before the scale is computed - configure_amax_reduction below) - ref pytorch/distributed.py:1056-1075."""
    data, _ = all_gather_rows(q._data.view(-1, q.shape[-1]), group)
    out = Float8Tensor(shape=(data.shape[0], q.shape[-1]), dtype=q.dtype, data=data,
                       fp8_scale_inv=q._scale_inv, fp8_dtype=q._fp8_dtype, quantizer=q._quantizer)
    if need_transpose:
        out._create_transpose()
    return out


def _all_gather_mxfp8(q: MXFP8Tensor, group=None) -> MXFP8Tensor:
    """Gather rowwise data + rowwise e8m0 scales and - when present - the columnwise pair, compact scales, padding
    stripped and re-added (ref pytorch/distributed.py:1614-1659).  A rank's rows are a whole number of 32-row
    blocks.  Reference layout (payload [m, K], scales [pad4(m/32), pad128(K)]): the gathered pair is the ranks'
    pairs one below the other - two more contiguous all-gathers, nothing is permuted; the GEMMs read it K-major.
    Transposed layout (amd_ext.h: payload [K, m], scales [pad128(K), pad4(m/32)]): the gathered payload is the
    per-rank payloads side by side, [K, tp*m], which costs a permuting copy after the gather."""
    assert not q._with_gemm_swizzled_scales, "all-gather needs compact (unswizzled) scales"
    tp = get_distributed_world_size(group)
    K = q.shape[-1]
    rows = q.numel() // K
    rd = rs = cd = cs = None
    if q._rowwise_data is not None:
        rd, _ = all_gather_rows(q._rowwise_data.view(rows, K), group)
        rs, _ = all_gather_rows(q._rowwise_scale_inv[:rows].contiguous(), group)
        pad = (-rs.shape[0]) % 128
        if pad:
            rs = torch.cat([rs, torch.zeros(pad, rs.shape[1], dtype=rs.dtype, device=rs.device)])
    transposed = q._columnwise_transposed
    if q._columnwise_data is not None and rows % 32 != 0:
        raise ValueError("MXFP8 sequence-parallel shards need a multiple of 32 rows per rank")
    if q._columnwise_data is not None and not transposed:
        nb = rows // 32
        cd, _ = all_gather_rows(q._columnwise_data.view(rows, K), group)                       # [tp*m, K]
        cs, _ = all_gather_rows(q._columnwise_scale_inv[:nb].contiguous(), group)              # [tp*nb, pad128(K)]
        pad = (-cs.shape[0]) % 4
        if pad:
            cs = torch.cat([cs, torch.zeros(pad, cs.shape[1], dtype=cs.dtype, device=cs.device)])
    elif q._columnwise_data is not None:
        nb = rows // 32
        cdl, _ = all_gather_rows(q._columnwise_data.view(1, K, rows), group)           # [tp, K, m]
        cd = cdl.permute(1, 0, 2).reshape(K, tp * rows).contiguous()
        csl, _ = all_gather_rows(q._columnwise_scale_inv[:K, :nb].contiguous().view(1, K, nb), group)
        cs_c = csl.permute(1, 0, 2).reshape(K, tp * nb)
        cs = torch.zeros((K + 127) // 128 * 128, (tp * nb + 3) // 4 * 4, dtype=cs_c.dtype, device=cs_c.device)
        cs[:K, :tp * nb] = cs_c
    return MXFP8Tensor(shape=(rows * tp, K), dtype=q.dtype, rowwise_data=rd, rowwise_scale_inv=rs,
                       columnwise_data=cd, columnwise_scale_inv=cs, fp8_dtype=q._fp8_dtype,
                       quantizer=q._quantizer, columnwise_transposed=cd is not None and transposed)


def _mx_k_slice(w8: MXFP8Tensor, k0: int, k1: int) -> MXFP8Tensor:
    """Rowwise MXFP8 operand restricted to the contraction range [k0, k1) (multiples of 128): a copy of the
    payload columns and of the matching scales (compact: columns k/32; GEMM-swizzled: whole 128-row x 4-block
    tiles, 512 bytes each, tile (row tile, k tile) at index row_tile * K/128 + k_tile)."""
    K = w8.shape[-1]
    rows = w8.numel() // K
    assert k0 % 128 == 0 and (k1 % 128 == 0 or k1 == K) and K % 128 == 0
    data = w8._rowwise_data.view(rows, K)[:, k0:k1].contiguous()
    sc = w8._rowwise_scale_inv
    if w8._with_gemm_swizzled_scales:
        rt = sc.numel() // ((K // 128) * 512)
        sc = sc.view(rt, K // 128, 512)[:, k0 // 128:k1 // 128].contiguous().view(rt * 128, (k1 - k0) // 32)
    else:
        sc = sc[:, k0 // 32:k1 // 32].contiguous()
    return MXFP8Tensor(shape=(rows, k1 - k0), dtype=w8.dtype, rowwise_data=data, rowwise_scale_inv=sc,
                       columnwise_data=None, columnwise_scale_inv=None, fp8_dtype=w8._fp8_dtype,
                       with_gemm_swizzled_scales=w8._with_gemm_swizzled_scales)


def _mx_row_slice(w8: MXFP8Tensor, r0: int, r1: int) -> MXFP8Tensor:
    """Rows [r0, r1) (multiples of 128) of a rowwise MXFP8 operand: views, no copy (a compact scale matrix is
    row-major, a swizzled one is stored by 128-row tiles)."""
    K = w8.shape[-1]
    rows = w8.numel() // K
    assert r0 % 128 == 0 and (r1 % 128 == 0 or r1 == rows)
    data = w8._rowwise_data.view(rows, K)[r0:r1]
    sc = w8._rowwise_scale_inv
    stride = sc.numel() // ((rows + 127) // 128 * 128)
    sc2 = sc.view(-1, stride)[r0:(r1 + 127) // 128 * 128]
    return MXFP8Tensor(shape=(r1 - r0, K), dtype=w8.dtype, rowwise_data=data, rowwise_scale_inv=sc2,
                       columnwise_data=None, columnwise_scale_inv=None, fp8_dtype=w8._fp8_dtype,
                       with_gemm_swizzled_scales=w8._with_gemm_swizzled_scales)


def _dgrad_operand_rows(w8):
    """For dgrad = dY W cut along its output columns: a function (c0, c1) -> the weight restricted to input features
    [c0, c1), built from ROW blocks of the K-contiguous copy the NN GEMM reads (per-tensor FP8: the transpose [K, N];
    MXFP8: the transposed columnwise pair) - views only.  None when the weight carries no such copy."""
    if isinstance(w8, Float8Tensor):
        if w8._transpose is None or w8._transpose_invalid:
            return None
        N = w8.shape[0]
        return lambda c0, c1: Float8Tensor(shape=(N, c1 - c0), dtype=w8.dtype, data=None, fp8_scale_inv=w8._scale_inv,
                                           fp8_dtype=w8._fp8_dtype, data_transpose=w8._transpose[c0:c1])
    if isinstance(w8, MXFP8Tensor) and w8._columnwise_transposed and w8._columnwise_data is not None:
        K = w8.shape[-1]
        N = w8.numel() // K
        sc = w8._columnwise_scale_inv
        stride = sc.numel() // ((K + 127) // 128 * 128)

        def rows(c0, c1):
            assert c0 % 128 == 0 and (c1 % 128 == 0 or c1 == K)
            return MXFP8Tensor(shape=(N, c1 - c0), dtype=w8.dtype, rowwise_data=None, rowwise_scale_inv=None,
                               columnwise_data=w8._columnwise_data.view(K, N)[c0:c1],
                               columnwise_scale_inv=sc.view(-1, stride)[c0:(c1 + 127) // 128 * 128],
                               fp8_dtype=w8._fp8_dtype, with_gemm_swizzled_scales=w8._with_gemm_swizzled_scales,
                               columnwise_transposed=True)
        return rows
    return None


def gather_along_first_dim(inp, group=None, quantizer=None, need_transpose: bool = False):
    """ref pytorch/distributed.py:1665: quantize locally (rowwise only for per-tensor FP8: a gathered
    transpose is not the transpose of the gather, linear.py:361-363), then all-gather bytes."""
    if isinstance(inp, torch.Tensor):
        if quantizer is None:
            out, _ = all_gather_rows(inp, group)
            return out
        if not isinstance(quantizer, MXFP8Quantizer):
            quantizer.set_usage(rowwise=True, columnwise=False)
        inp = quantizer(inp)
    if isinstance(inp, Float8Tensor):
        return _all_gather_fp8(inp, group, need_transpose)
    if isinstance(inp, MXFP8Tensor):
        return _all_gather_mxfp8(inp, group)
    raise TypeError(type(inp))


def _keep_weight_transposed(state) -> None:
    from .linear import FWD_WEIGHT
    qw = state.fwd_q[FWD_WEIGHT]
    if isinstance(qw, MXFP8Quantizer):
        qw.columnwise_transposed = True


def configure_amax_reduction(quantizer, group) -> None:
    """A tensor that is sharded over `group` and gathered in FP8 must be quantized with ONE scale on
    every rank.  Current scaling: the amax is MAX-all-reduced over the group before the scale is
    computed (ref module/linear.py:319-322, csrc/quantizer.cpp:994-999).  Delayed scaling gets the same
    from the recipe step's packed amax all-reduce; MXFP8 scales are per block and need nothing."""
    if isinstance(quantizer, Float8CurrentScalingQuantizer) and collectives_active(group):
        quantizer.with_amax_reduction = True
        quantizer.amax_reduction_group = group


# ---------------------------------------------------------------------------------------
# column-parallel + sequence-parallel Linear (the C4 pattern, ref module/linear.py:314-549,
# :760-1261 with sequence_parallel=True, parallel_mode="column")
# ---------------------------------------------------------------------------------------
class ColumnParallelLinearSP:
    """Weight shard [N/tp, K]; activations sharded on tokens.  forward: FP8 all-gather of X
    overlapped with the K-chunked GEMMs; backward: dgrad partial sums reduce-scattered to the
    token shard while wgrad runs; amax all-reduce (MAX) once per fwd and once per bwd."""

    def __init__(self, in_features: int, out_features: int, tp: int, rank: int, recipe=None,
                 device=None, group=None, params_dtype=torch.bfloat16, overlap: bool = True, chunks: int = 2,
                 activation_transposes: bool = False):
        from .linear import FWD_INPUT, LinearFP8State
        assert out_features % tp == 0
        # per-tensor FP8: dY is cast row-wise only (3 B/element) and wgrad reads it K-major (bit-equal);
        # True restores the cast_transpose of rounds 1-2.  The cached weight keeps its transpose (dgrad's output-column
        # pieces are row blocks of it).  MXFP8 always carries its columnwise pair.
        self.activation_transposes = activation_transposes
        self.K, self.N_local, self.tp, self.rank, self.group = in_features, out_features // tp, tp, rank, group
        self.device = device or torch.device("cuda")
        g = torch.Generator(device="cpu").manual_seed(4321 + rank)
        self.weight = (torch.randn(self.N_local, in_features, generator=g) * 0.02).to(params_dtype).to(self.device)
        # the gathered operand keeps compact MXFP8 scales (ref module/linear.py:2069)
        # (MXFP8: the gathered operand's columnwise pair travels in the reference layout, _all_gather_mxfp8 - plain
        # row gathers; the cached weight shard keeps the transposed one, dgrad's output-column pieces are row blocks of it)
        self.state = LinearFP8State(recipe or DelayedScaling(), device=self.device, compact_mx_scales=("input",), fp8_transposes="all",
                                    mx_columnwise="reference")
        _keep_weight_transposed(self.state)
        self.overlap = overlap
        self.chunks = chunks
        self.out_dtype = params_dtype
        # the input is sharded on tokens and gathered in FP8: one scale over the whole group
        configure_amax_reduction(self.state.fwd_q[FWD_INPUT], group)
        self._w_chunks = None      # K slices of the cached FP8 weight (bytes), for the chunked forward

    # -- pieces shared with the LayerNorm variant -----------------------------------------
    def _cached_weight(self, is_first_microbatch):
        from .linear import FWD_WEIGHT
        st = self.state
        qw = st.fwd_q[FWD_WEIGHT]
        if st.weight_cache is None or is_first_microbatch is None or is_first_microbatch:
            qw.set_usage(rowwise=True, columnwise=True)
            st.weight_cache = qw(self.weight)
            self._w_chunks = None
        return st.weight_cache

    def _weight_k_chunks(self, w8, pts):
        """K slices of the cached FP8 weight, rebuilt only when the weight is re-quantized.  MXFP8: the payload
        columns [k0, k1) and the matching scale columns (compact) or whole 128 x 4 scale tiles (swizzled)."""
        if self._w_chunks is None or self._w_chunks[0] != pts:
            if isinstance(w8, MXFP8Tensor):
                views = [_mx_k_slice(w8, k0, k1) for k0, k1 in pts]
            else:
                views = [Float8Tensor(shape=(self.N_local, k1 - k0), dtype=w8.dtype,
                                      data=w8._data[:, k0:k1].contiguous(), fp8_scale_inv=w8._scale_inv,
                                      fp8_dtype=w8._fp8_dtype) for k0, k1 in pts]
            self._w_chunks = (pts, views)
        return self._w_chunks[1]

    def _quantize_input(self, x_local):
        from .linear import FWD_INPUT
        qi = self.state.fwd_q[FWD_INPUT]
        if isinstance(qi, MXFP8Quantizer):
            qi.set_usage(rowwise=True, columnwise=True)      # the transposed columnwise pair is gathered too
        else:
            qi.set_usage(rowwise=True, columnwise=False)
        return qi(x_local)

    def _gather_and_gemm(self, x8_local, w8, m):
        """Returns (y, x8 for backward)."""
        from .gemm import general_gemm
        tp = self.tp
        pts = tuple(split_points(self.K, self.chunks)) if (self.overlap and collectives_active(self.group)) else ()
        if len(pts) > 1:
            # The chunk GEMMs accumulate in FP32 and the sum is rounded to the output dtype ONCE (a bf16
            # accumulator would round after every chunk): the result then differs from the unsplit
            # GEMM only by the fp32 rounding of the per-chunk partial sums.
            y_acc = torch.empty(m * tp, self.N_local, dtype=torch.float32, device=self.device)
            w_chunks = self._weight_k_chunks(w8, pts)
            if isinstance(x8_local, Float8Tensor):
                def chunk_gemm(buf, i, k0, k1):
                    xs = Float8Tensor(shape=(m * tp, k1 - k0), dtype=self.out_dtype, data=buf,
                                      fp8_scale_inv=x8_local._scale_inv, fp8_dtype=x8_local._fp8_dtype)
                    general_gemm(w_chunks[i], xs, layout="TN", out=y_acc, accumulate=i > 0)

                bufs = gather_and_gemm_kchunked(x8_local._data.view(m, self.K), chunk_gemm, self.group, self.chunks)
                # backward only needs X^T: the transposes of the gathered chunks are its row blocks
                x8 = _GatheredChunks(bufs, pts, x8_local)
            else:
                x8 = self._gather_and_gemm_mx(x8_local, w_chunks, pts, m, y_acc)
            y = y_acc.to(self.out_dtype)
        else:
            y = torch.empty(m * tp, self.N_local, dtype=self.out_dtype, device=self.device)
            x8 = gather_along_first_dim(x8_local, self.group)
            general_gemm(w8, x8, layout="TN", out=y)
        return y, x8

    def _gather_and_gemm_mx(self, x8_local: MXFP8Tensor, w_chunks, pts, m, y_acc):
        """The K-chunked schedule for MXFP8 (the reference's split_overlap_ag is recipe-agnostic,
        comm_gemm_overlap.cpp:1044-1216): chunk i = payload columns [k0, k1) and their e8m0 scale columns, two
        collectives per chunk that both land contiguous (nothing gathered is copied); the transposed
        columnwise pair that only backward reads is gathered behind the chunk GEMMs."""
        from .gemm import general_gemm
        tp = self.tp
        assert not x8_local._with_gemm_swizzled_scales, "the gathered operand keeps compact scales"
        data = x8_local._rowwise_data.view(m, self.K)
        sc = x8_local._rowwise_scale_inv
        gathered, works = [], []
        for k0, k1 in pts:
            d_out, d_work = all_gather_rows(data[:, k0:k1].contiguous(), self.group, async_op=True)
            s_out, s_work = all_gather_rows(sc[:m, k0 // 32:k1 // 32].contiguous(), self.group, async_op=True)
            gathered.append((d_out, s_out)); works.append((d_work, s_work))
        rows_pad = (-(m * tp)) % 128
        for i, (k0, k1) in enumerate(pts):
            for w in works[i]:
                if w is not None:
                    w.wait()
            xs_data, xs_sc = gathered[i]
            if rows_pad:
                xs_sc = torch.cat([xs_sc, torch.zeros(rows_pad, xs_sc.shape[1], dtype=xs_sc.dtype, device=xs_sc.device)])
            xs = MXFP8Tensor(shape=(m * tp, k1 - k0), dtype=self.out_dtype, rowwise_data=xs_data,
                             rowwise_scale_inv=xs_sc, columnwise_data=None, columnwise_scale_inv=None,
                             fp8_dtype=x8_local._fp8_dtype)
            general_gemm(w_chunks[i], xs, layout="TN", out=y_acc, accumulate=i > 0)
        col_only = MXFP8Tensor(shape=x8_local.shape, dtype=x8_local.dtype, rowwise_data=None, rowwise_scale_inv=None,
                               columnwise_data=x8_local._columnwise_data,
                               columnwise_scale_inv=x8_local._columnwise_scale_inv, fp8_dtype=x8_local._fp8_dtype,
                               quantizer=x8_local._quantizer, columnwise_transposed=x8_local._columnwise_transposed)
        return _all_gather_mxfp8(col_only, self.group)

    def forward(self, x_local: torch.Tensor, is_first_microbatch=None):
        st = self.state
        w8 = self._cached_weight(is_first_microbatch)
        x8_local = self._quantize_input(x_local)
        y, x8 = self._gather_and_gemm(x8_local, w8, x_local.shape[0])
        st.update_fwd(self.group)
        return y, (x8, w8)

    def _backward_gemms(self, dy: torch.Tensor, saved):
        """Returns (dx_local [m, K] after reduce-scatter, dw)."""
        from .gemm import general_gemm
        from .linear import BWD_GRAD_OUTPUT
        x8, w8 = saved
        st = self.state
        qg = st.bwd_q[BWD_GRAD_OUTPUT]
        qg.set_usage(rowwise=True, columnwise=isinstance(qg, MXFP8Quantizer) or self.activation_transposes)
        dy8 = qg(dy)
        # dgrad partial sums [M_total, K] -> reduce-scatter to this rank's tokens, asynchronously.  With the overlap on,
        # dgrad is cut along its OUTPUT columns (row blocks of the K-contiguous W^T copy: views, no copy): piece j's
        # reduce-scatter starts under dgrad piece j+1 instead of only under wgrad - at tp = 8 the whole reduce-scatter
        # (bf16, 8192 tp x K x 2 B x (tp-1)/tp) is longer than the wgrad GEMM it used to hide under (section 4).
        handle = None
        w8t = _dgrad_operand_rows(w8) if (self.overlap and collectives_active(self.group)) else None
        if w8t is not None:
            def cols_gemm(c0, c1):
                out, *_ = general_gemm(w8t(c0, c1), dy8, out_dtype=self.out_dtype, layout="NN", grad=True)
                return out
            dx_local = gemm_and_reduce_scatter_nchunked(self.K, cols_gemm, self.group, self.chunks)
        else:
            dx_partial, *_ = general_gemm(w8, dy8, out_dtype=self.out_dtype, layout="NN", grad=True)
            dx_local, handle = reduce_scatter_along_first_dim(dx_partial, self.group, async_op=True)
        # wgrad contracts over the rows of the GATHERED input.  The reference rebuilds X^T locally after the gather
        # (distributed.py:1067-1075); here only the K-chunked pipeline does (its chunks are column slices, whose
        # transposes are the row blocks of X^T) - a gathered row-wise tensor is read K-major by the GEMM as it is
        # (ds_read_b64_tr_b8, DESIGN 3.6c: bit-equal, one pass over the gathered bytes fewer)
        if isinstance(x8, _GatheredChunks):
            x8 = x8.as_transposed_tensor()
        dw, *_ = general_gemm(x8, dy8, out_dtype=self.out_dtype, layout="NT", grad=True)
        if handle is not None:
            handle.wait()
        return dx_local, dw

    def backward(self, dy: torch.Tensor, saved):
        dx_local, dw = self._backward_gemms(dy, saved)
        self.state.update_bwd(self.group)
        return dx_local, dw

    def step(self, x_local, dy, is_first_microbatch=None):
        y, saved = self.forward(x_local, is_first_microbatch)
        dx, dw = self.backward(dy, saved)
        return y, dx, dw


class _GatheredChunks:
    """The K-chunked all-gather result kept for backward: wgrad reads X^T [K, M], whose row block
    [k0, k1) is the byte transpose of gathered chunk i."""

    def __init__(self, bufs: Sequence[torch.Tensor], pts, like: Float8Tensor):
        self.bufs, self.pts, self.like = list(bufs), list(pts), like

    def rowwise_bytes(self) -> torch.Tensor:
        return self.bufs[0] if len(self.bufs) == 1 else torch.cat(self.bufs, dim=1)

    def as_transposed_tensor(self) -> Float8Tensor:
        from ._lib import api
        from .tensor import TensorWrapper, current_stream_ptr
        M = self.bufs[0].shape[0]
        K = self.pts[-1][1]
        xt = torch.empty((K, M), dtype=torch.uint8, device=self.bufs[0].device)
        for buf, (k0, k1) in zip(self.bufs, self.pts):
            wi = TensorWrapper().set_rowwise_data(buf, self.like._fp8_dtype)
            wo = TensorWrapper().set_rowwise_data(xt[k0:k1], self.like._fp8_dtype)
            api.nvte_transpose(wi.handle, wo.handle, current_stream_ptr())
            wi.destroy(); wo.destroy()
        return Float8Tensor(shape=(M, K), dtype=self.like.dtype, data=None, fp8_scale_inv=self.like._scale_inv,
                            fp8_dtype=self.like._fp8_dtype, data_transpose=xt)


class ColumnParallelLayerNormLinearSP(ColumnParallelLinearSP):
    """BASELINE configs[3]: LayerNormLinear, column-parallel + sequence-parallel (ref
    pytorch/module/layernorm_linear.py with parallel_mode="column", sequence_parallel=True).  Each rank
    normalizes ITS tokens and writes the GEMM's FP8 operand directly (norm + quantize fused for delayed
    scaling / MXFP8), the FP8 bytes are all-gathered, the GEMM runs on the full token set; backward
    reduce-scatters dgrad to the local tokens and runs the norm backward there.  dgamma / dbeta are
    the LOCAL tokens' sums: with sequence parallelism the framework all-reduces them over the TP group
    (Megatron's `sequence_parallel` parameter attribute); `allreduce_norm_grads=True` does it here."""

    def __init__(self, in_features: int, out_features: int, tp: int, rank: int, recipe=None, device=None,
                 group=None, params_dtype=torch.bfloat16, overlap: bool = True, chunks: int = 2,
                 normalization: str = "RMSNorm", eps: float = 1e-5, zero_centered_gamma: bool = False,
                 allreduce_norm_grads: bool = False, activation_transposes: bool = False):
        super().__init__(in_features, out_features, tp, rank, recipe, device, group, params_dtype, overlap, chunks,
                         activation_transposes)
        if normalization not in ("LayerNorm", "RMSNorm"):
            raise ValueError(f"unsupported normalization {normalization}")
        self.normalization, self.eps, self.zero_centered_gamma = normalization, eps, zero_centered_gamma
        self.allreduce_norm_grads = allreduce_norm_grads
        g0 = torch.zeros if zero_centered_gamma else torch.ones
        self.layer_norm_weight = g0(in_features, dtype=params_dtype, device=self.device)
        self.layer_norm_bias = (torch.zeros(in_features, dtype=params_dtype, device=self.device)
                                if normalization == "LayerNorm" else None)

    def forward(self, x_local: torch.Tensor, is_first_microbatch=None):
        from .linear import FWD_INPUT
        from .normalization import layernorm_fwd, rmsnorm_fwd
        st = self.state
        w8 = self._cached_weight(is_first_microbatch)
        qi = st.fwd_q[FWD_INPUT]
        if isinstance(qi, MXFP8Quantizer):
            qi.set_usage(rowwise=True, columnwise=True)
        else:
            qi.set_usage(rowwise=True, columnwise=False)
        x2d = x_local.reshape(-1, x_local.shape[-1])
        if self.normalization == "LayerNorm":
            ln8, mu, rsigma = layernorm_fwd(x2d, self.layer_norm_weight, self.layer_norm_bias, self.eps,
                                            quantizer=qi, zero_centered_gamma=self.zero_centered_gamma)
        else:
            ln8, mu, rsigma = rmsnorm_fwd(x2d, self.layer_norm_weight, self.eps, quantizer=qi,
                                          zero_centered_gamma=self.zero_centered_gamma)
        y, x8 = self._gather_and_gemm(ln8, w8, x2d.shape[0])
        st.update_fwd(self.group)
        return y, (x8, w8, x2d, mu, rsigma)

    def backward(self, dy: torch.Tensor, saved):
        from .normalization import layernorm_bwd, rmsnorm_bwd
        x8, w8, x2d, mu, rsigma = saved
        dln_local, dw = self._backward_gemms(dy, (x8, w8))
        dln_local = dln_local.to(self.layer_norm_weight.dtype)
        if self.normalization == "LayerNorm":
            dx, dgamma, dbeta = layernorm_bwd(dln_local, x2d, mu, rsigma, self.layer_norm_weight,
                                              zero_centered_gamma=self.zero_centered_gamma)
        else:
            dx, dgamma = rmsnorm_bwd(dln_local, x2d, rsigma, self.layer_norm_weight,
                                     zero_centered_gamma=self.zero_centered_gamma)
            dbeta = None
        if self.allreduce_norm_grads and collectives_active(self.group):
            for t in (dgamma, dbeta):
                if t is not None:
                    _all_reduce_sum_(t, self.group)
        self.state.update_bwd(self.group)
        return dx, dw, dgamma, dbeta

    def step(self, x_local, dy, is_first_microbatch=None):
        y, saved = self.forward(x_local, is_first_microbatch)
        dx, dw, dgamma, dbeta = self.backward(dy, saved)
        return y, dx, dw, dgamma, dbeta


def _all_reduce_sum_(t: torch.Tensor, group) -> None:
    if _is_gloo(group) and t.is_cuda:
        h = t.float().cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h.to(t.dtype))
    else:
        dist.all_reduce(t, group=group)


# ---------------------------------------------------------------------------------------
# row-parallel + sequence-parallel Linear (ref module/linear.py:539-549 forward reduce-scatter,
# module/base.py:1680-1684 FP8 all-gather of dY in backward)
# ---------------------------------------------------------------------------------------
class RowParallelLinearSP:
    """Weight shard [N, K/tp].  forward: the input holds ALL tokens but only this rank's K/tp features
    (the output of a column-parallel layer); Y partial sums [M, N] are reduce-scattered to the token
    shard, the GEMM cut along N so that piece j's reduce-scatter runs under GEMM j+1.  backward: dY
    [M/tp, N] is quantized with a group-wide scale, all-gathered in FP8 (transpose rebuilt locally),
    dgrad and wgrad run on the full token set without further communication."""

    def __init__(self, in_features: int, out_features: int, tp: int, rank: int, recipe=None, device=None,
                 group=None, params_dtype=torch.bfloat16, overlap: bool = True, chunks: int = 2,
                 activation_transposes: bool = False):
        from .linear import BWD_GRAD_OUTPUT, LinearFP8State
        assert in_features % tp == 0
        # per-tensor FP8: X is cast row-wise only and wgrad reads X and the gathered dY K-major (see the column layer)
        self.activation_transposes = activation_transposes
        self.K_local, self.N, self.tp, self.rank, self.group = in_features // tp, out_features, tp, rank, group
        self.device = device or torch.device("cuda")
        g = torch.Generator(device="cpu").manual_seed(8765 + rank)
        self.weight = (torch.randn(out_features, self.K_local, generator=g) * 0.02).to(params_dtype).to(self.device)
        # the gathered operand keeps compact MXFP8 scales (ref module/linear.py:2077)
        self.state = LinearFP8State(recipe or DelayedScaling(), device=self.device, compact_mx_scales=("grad_output",), fp8_transposes="all",
                                    mx_columnwise="reference")
        self.overlap, self.chunks, self.out_dtype = overlap, chunks, params_dtype
        # dY is sharded on tokens and gathered in FP8: one scale over the whole group
        configure_amax_reduction(self.state.bwd_q[BWD_GRAD_OUTPUT], group)

    def forward(self, x: torch.Tensor, is_first_microbatch=None):
        from .gemm import general_gemm
        from .linear import FWD_INPUT, FWD_WEIGHT
        st = self.state
        qi, qw = st.fwd_q[FWD_INPUT], st.fwd_q[FWD_WEIGHT]
        if st.weight_cache is None or is_first_microbatch is None or is_first_microbatch:
            qw.set_usage(rowwise=True, columnwise=True)
            st.weight_cache = qw(self.weight)
        w8 = st.weight_cache
        qi.set_usage(rowwise=True, columnwise=isinstance(qi, MXFP8Quantizer) or self.activation_transposes)
        x8 = qi(x)
        M = x.shape[0]
        if self.overlap and collectives_active(self.group):
            def cols_gemm(c0, c1):
                if isinstance(w8, MXFP8Tensor):
                    wj = _mx_row_slice(w8, c0, c1)
                else:
                    wj = Float8Tensor(shape=(c1 - c0, self.K_local), dtype=w8.dtype, data=w8._data[c0:c1],
                                      fp8_scale_inv=w8._scale_inv, fp8_dtype=w8._fp8_dtype)
                out, *_ = general_gemm(wj, x8, out_dtype=self.out_dtype, layout="TN")
                return out
            y_local = gemm_and_reduce_scatter_nchunked(self.N, cols_gemm, self.group, self.chunks)
        else:
            y_partial, *_ = general_gemm(w8, x8, out_dtype=self.out_dtype, layout="TN")
            y_local, _ = reduce_scatter_along_first_dim(y_partial, self.group)
        st.update_fwd(self.group)
        return y_local, (x8, w8)

    def backward(self, dy_local: torch.Tensor, saved):
        from .gemm import general_gemm
        from .linear import BWD_GRAD_OUTPUT
        x8, w8 = saved
        st = self.state
        qg = st.bwd_q[BWD_GRAD_OUTPUT]
        # ref base.py:1680-1684 gathers dY and rebuilds its transpose for wgrad; per-tensor FP8: wgrad reads the gathered
        # row-wise bytes K-major instead (no transpose pass over [M_total, N]); MXFP8 gathers its columnwise pair
        dy8 = gather_along_first_dim(dy_local.reshape(-1, dy_local.shape[-1]), self.group, quantizer=qg,
                                     need_transpose=False)
        dx, *_ = general_gemm(w8, dy8, out_dtype=self.out_dtype, layout="NN", grad=True)
        dw, *_ = general_gemm(x8, dy8, out_dtype=self.out_dtype, layout="NT", grad=True)
        st.update_bwd(self.group)
        return dx, dw

    def step(self, x, dy_local, is_first_microbatch=None):
        y, saved = self.forward(x, is_first_microbatch)
        dx, dw = self.backward(dy_local, saved)
        return y, dx, dw
