#!/usr/bin/env python
"""Headline benchmark: FP8 Linear fwd+bwd TFLOP/s (+ cast_transpose GB/s) on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--no-graph] [--no-cpu-baseline]

N = 1 : BASELINE.json configs[1] - Llama-3-8B MLP Linear 4096 -> 14336, 8192 tokens, FP8 delayed
        scaling (HYBRID: E4M3 forward, E5M2 backward), forward + backward.
        One "step" = quantize X (row+col), GEMM fprop, recipe step (fwd), quantize dY (row+col),
        GEMM dgrad, GEMM wgrad, recipe step (bwd).  The weight quantize is amortised exactly as
        te.Linear does with is_first_microbatch=False (cached FP8 weight); its stand-alone cost is
        reported in "extra".
N > 1 : (default, --parallelism dp) every rank runs that same step on its own 8192 tokens with a
        full copy of the weight; the path's one real exchange at this level is the delayed-scaling
        amax all-reduce (MAX over the FP8 group, ref quantization.py:666-674), once per forward
        and once per backward, on RCCL.  Per-GPU work fixed: "scaling": "weak".
        --parallelism tp_sp : the Linear tensor-parallel + sequence-parallel over N ranks
        (column-parallel, FP8 all-gather of the input, reduce-scatter of dgrad, amax all-reduce),
        tokens grow with N (8192 per rank), also weak scaling (BASELINE configs[3] style).

Prints ONE JSON line on rank 0.  Inputs are resident in HBM before the timed region.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP8_TFLOPS = 5000.0   # dense FP8 MFMA, MI355X_MICROARCH.md "Peak FP8 MFMA ~5 PF dense"
PEAK_HBM_GBPS = 8000.0     # HBM3E spec, MI355X_MICROARCH.md
# newest first: the PMC passes (profiles/collect.sh) of the kernels this tree ships; each file records the
# commit and date it was taken at under "_meta"
TRAFFIC_FILES = ("round3_traffic.json", "round2_traffic.json", "round1_traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tokens", type=int, default=8192)
    ap.add_argument("--in-features", type=int, default=4096)
    ap.add_argument("--out-features", type=int, default=14336)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--parallelism", choices=["dp", "tp_sp", "ep"], default="dp",
                    help="dp: N replicas of the configs[1] step (default); tp_sp: sharded layers (configs[3] pattern); "
                         "ep: configs[4], 8 experts spread over the ranks, no collective inside the path")
    ap.add_argument("--tp-layer", choices=["column", "row", "ln_column"], default="column",
                    help="tp_sp: column-parallel Linear (default), row-parallel Linear, or column-parallel LayerNormLinear (configs[3])")
    ap.add_argument("--tp-chunks", type=int, default=2, help="tp_sp: pieces the AG+GEMM / GEMM+RS pipelines are cut into")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo only to rehearse on one device)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--gemm-cus", type=int, default=0,
                    help="cap the persistent GEMM grid at this many CUs (NVTE_EXT_MARGIN_SM of the reference: the rest is "
                         "left to the RCCL kernels of an overlapped step); 0 = all")
    ap.add_argument("--fp8-transposes", default="none", choices=["none", "input", "grad_output", "all"],
                    help="which per-tensor FP8 tensors keep a transposed copy for the backward GEMMs (LinearFP8State): 'none' = "
                         "the library default (row-wise payloads read K-major by dgrad / wgrad), 'all' = the data flow of "
                         "rounds 1-2 (cast_transpose for X and dY); outputs are bit-equal")
    ap.add_argument("--rehearse-multi", action="store_true",
                    help="N = 1 only: take the N > 1 code paths (two graphs + side-stream amax all-reduce for dp, the sharded "
                         "layers for tp_sp) over a world-size-1 RCCL group with the collectives forced on - what a one-GPU box "
                         "can execute of the multi-GPU run; the numbers it prints are not a scaling result")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grouped", action="store_true", help="skip the configs[4]-shaped GroupedLinear side measurement")
    return ap.parse_args()


def load_traffic():
    """Fabric-side bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 on
    gfx950 + WRITE_SIZE, see profiles/README.md).  Counters cannot be read from inside this
    process, so the figures come from the separate profiled run of the same kernels and shapes."""
    for name in TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as fh:
                table = json.load(fh)
            table["_file"] = "profiles/" + name
            return table
    return {}


def traffic_of(table, role):
    rec = table.get(role)
    return rec.get("traffic_bytes_per_launch") if rec else None


def time_kernel(fn, iters, warm=3):
    """Average device time (ms) of fn() measured with events on torch's current stream - the
    stream every nvte_* call in fn is enqueued on."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    gc.collect()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def graph_of(step):
    """The step replayed from a HIP graph (as the headline step is timed); the eager callable if capture fails."""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        torch.cuda.synchronize()
        return g.replay, True
    except Exception as exc:
        print(f"[bench] graph capture unavailable for a side measurement ({type(exc).__name__}: {exc})", file=sys.stderr)
        torch.cuda.synchronize()
        return step, False


def grouped_c5(dev, iters):
    """Side measurement (not `value`): BASELINE configs[4]-shaped GroupedLinear step on ONE GPU -
    8 experts 4096->14336, 16384 routed tokens (8192 tokens x top-2, fixed-seed ragged splits padded
    to 16), FP8 current scaling, fwd+bwd on the grouped path with device-side splits."""
    import numpy as np
    from transformer_engine_amd.grouped_linear import (GroupedLinearFP8State, grouped_linear_backward,
                                                       grouped_linear_forward)
    from transformer_engine_amd.recipe import Float8CurrentScaling
    G, K, N, routed = 8, 4096, 14336, 16384
    counts = np.bincount(np.random.default_rng(1234).integers(0, G, size=routed), minlength=G)
    m_splits = [int((c + 15) // 16 * 16) for c in counts]
    M = sum(m_splits)
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(G * N, K, device=dev, dtype=torch.bfloat16) * 0.02
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    fd = torch.tensor(m_splits, dtype=torch.int64, device=dev)
    st = GroupedLinearFP8State(Float8CurrentScaling(), G, dev)
    grads = [torch.zeros(N, K, device=dev, dtype=torch.bfloat16) for _ in range(G)]
    grouped_linear_forward(x, w, fd, st, is_first_microbatch=True)     # weights quantized once, cached

    def step():
        y, saved = grouped_linear_forward(x, w, fd, st, is_first_microbatch=False)
        grouped_linear_backward(dy, saved, fd, st, wgrad_out=grads, accumulate_wgrad=False)

    for _ in range(3):
        step()
    ms_eager = time_kernel(step, iters)
    run, graphed = graph_of(step)
    ms_graph = time_kernel(run, iters) if graphed else None
    ms = min(ms_eager, ms_graph) if graphed else ms_eager      # replay is not always faster: its missing gaps cost clock
    return {"workload": "8 experts 4096->14336, 16384 routed tokens, current scaling, fwd+bwd, 1 GPU",
            "m_splits": m_splits, "ms_per_step": ms, "tflops": 6.0 * M * N * K / (ms * 1e-3) / 1e12,
            "ms_per_step_hip_graph": ms_graph, "ms_per_step_eager": ms_eager}


def mxfp8_c3(dev, iters):
    """Side measurement (not `value`): BASELINE configs[2]-shaped MXFP8 Linear step on ONE GPU - 8192 tokens,
    8192 -> 8192, MXFP8 block scaling (scales written GEMM-swizzled, the module default), fwd+bwd, weight cached."""
    from transformer_engine_amd import linear as L
    from transformer_engine_amd.recipe import MXFP8BlockScaling
    M = K = N = 8192
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(N, K, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
    dy = (torch.randn(M, N, device=dev, dtype=torch.float32) * 1e-2).to(torch.bfloat16)
    st = L.LinearFP8State(MXFP8BlockScaling(), device=dev)
    L.linear_forward(x, w, None, st, is_first_microbatch=True)

    def step():
        y, saved = L.linear_forward(x, w, None, st, is_first_microbatch=False)
        L.linear_backward(dy, saved, st)

    for _ in range(5):      # allocator and clocks settled before the timed loop
        step()
    ms_eager = time_kernel(step, iters)
    run, graphed = graph_of(step)
    ms_graph = time_kernel(run, iters) if graphed else None
    ms = min(ms_eager, ms_graph) if graphed else ms_eager
    return {"workload": "Linear 8192->8192, 8192 tokens, MXFP8 block scaling (columnwise copies in the reference's own "
                        "layout, read K-major by dgrad / wgrad), fwd+bwd, 1 GPU",
            "ms_per_step": ms, "tflops": 6.0 * M * N * K / (ms * 1e-3) / 1e12, "ms_per_step_hip_graph": ms_graph,
            "ms_per_step_eager": ms_eager}


def cpu_baseline(tokens=4096, K=4096, N=14336):
    """The reference's CPU path (tests/cpp host references, restated in oracle/) on a bounded
    sample of the SAME workload: the configs[1] Linear (4096 -> 14336, full-size weight) forward +
    backward on `tokens` of its 8192 tokens: three cast_transpose+amax (serial loops, as the
    reference's compute_ref) and three FP32 GEMMs of the dequantized operands (OpenMP over the
    host cores the process may use)."""
    import numpy as np
    import oracle
    M = tokens
    rng = np.random.default_rng(1234)
    mk = lambda s, sc: oracle.float_to_bf16_bits(rng.standard_normal(s, dtype=np.float32) * np.float32(sc))
    x, w, dy = mk((M, K), 1.0), mk((N, K), 0.02), mk((M, N), 0.01)
    t0 = time.perf_counter()
    qx, qxt, ax, sx = oracle.cast_transpose(x, 64.0, oracle.E4M3)
    qw, qwt, aw, sw = oracle.cast_transpose(w, 2048.0, oracle.E4M3)
    qdy, qdyt, ady, sdy = oracle.cast_transpose(dy, 65536.0, oracle.E5M2)
    t1 = time.perf_counter()
    fx, fw, fdy = (oracle.dequant(qx, 0, sx), oracle.dequant(qw, 0, sw), oracle.dequant(qdy, 1, sdy))
    fwt, fdyt, fxt = (oracle.dequant(qwt, 0, sw), oracle.dequant(qdyt, 1, sdy), oracle.dequant(qxt, 0, sx))
    t2 = time.perf_counter()
    oracle.gemm_tn(fx, fw, fast=True)       # fprop  Y  = X W^T
    oracle.gemm_tn(fdy, fwt, fast=True)     # dgrad  dX = dY W
    oracle.gemm_tn(fdyt, fxt, fast=True)    # wgrad  dW = dY^T X
    t3 = time.perf_counter()
    flops = 6.0 * M * N * K
    total = (t1 - t0) + (t3 - t2)
    elems = M * K + N * K + M * N
    return {
        "value": flops / total / 1e12, "unit": "TFLOP/s", "cores": oracle.num_threads(), "kind": "port",
        "sample": f"Linear {K}->{N} fwd+bwd on {M} tokens (configs[1]: 4096 of its 8192 tokens, full-size weight): "
                  "3x cast_transpose+amax (serial reference loops, 1 core) + 3x dequantized-FP32 GEMM "
                  "(OpenMP, `cores` threads); dequantize-to-FP32 staging excluded",
        "cast_transpose_GBps": elems * 4 / (t1 - t0) / 1e9, "cast_transpose_cores": 1,
        "gemm_GFLOPs": flops / (t3 - t2) / 1e9, "seconds": total,
    }


def main():
    args = parse()
    # No collector pauses inside any timed loop: the eager paths (tp_sp, the N > 1 data-parallel segments, the
    # instrumented steps) were seen to take one ~40 ms host stall every few dozen steps - the size of a full Python
    # garbage collection.  Collections happen explicitly, in front of the timed loops.
    gc.disable()
    if os.environ.get("BENCH_DEBUG_HANG"):   # rehearsal aid: dump all stacks and exit if stuck
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_DEBUG_HANG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import transformer_engine_amd as te
    from transformer_engine_amd import linear as L
    from transformer_engine_amd.recipe import DelayedScaling

    dist = None
    multi = world > 1 or args.rehearse_multi      # which code paths run (world stays the number of ranks)
    if multi:
        import torch.distributed as dist
        if world == 1:
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(port))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        if world == 1:
            from transformer_engine_amd import distributed as _D
            _D.force_collectives(True)

    M, K, N = args.tokens, args.in_features, args.out_features
    torch.manual_seed(1234 + rank)
    recipe = DelayedScaling()   # HYBRID, history 1024, "max", margin 0 (reference defaults)
    if args.gemm_cus > 0:
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        os.environ["NVTE_EXT_MARGIN_SM"] = str(max(cus - args.gemm_cus, 0))

    ep_info = None
    if args.parallelism == "ep":
        # BASELINE configs[4]: Mixtral 8x7B GroupedLinear, expert-parallel.  The 8 experts are spread over the ranks
        # (8 / N each); a rank sees the routed tokens of ITS experts (fixed-seed ragged splits of 8192 tokens x top-2,
        # padded to 16 rows) already dispatched - the all-to-all belongs to the MoE layer around GroupedLinear, nothing
        # inside this path communicates.  Total work is fixed as N grows: "strong" scaling.
        import numpy as np
        from transformer_engine_amd.grouped_linear import (GroupedLinearFP8State, grouped_linear_backward,
                                                           grouped_linear_forward)
        from transformer_engine_amd.recipe import Float8CurrentScaling
        E, routed = 8, 16384
        if E % world != 0:
            raise SystemExit("--parallelism ep needs a rank count that divides the 8 experts")
        counts = np.bincount(np.random.default_rng(1234).integers(0, E, size=routed), minlength=E)
        all_splits = [int((c + 15) // 16 * 16) for c in counts]
        G = E // world
        m_splits = all_splits[rank * G:(rank + 1) * G]
        Mr = sum(m_splits)
        x = torch.randn(Mr, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(G * N, K, device=dev, dtype=torch.bfloat16) * 0.02
        dy = torch.randn(Mr, N, device=dev, dtype=torch.bfloat16)
        fd = torch.tensor(m_splits, dtype=torch.int64, device=dev)
        gst = GroupedLinearFP8State(Float8CurrentScaling(), G, dev)
        grads = [torch.zeros(N, K, device=dev, dtype=torch.bfloat16) for _ in range(G)]

        def step(first=False):
            y, saved = grouped_linear_forward(x, w, fd, gst, is_first_microbatch=first)
            return grouped_linear_backward(dy, saved, fd, gst, wgrad_out=grads, accumulate_wgrad=False)

        flops_per_step = 6.0 * sum(all_splits) * N * K          # whole job
        parallelism = f"ep{world} ({G} experts per rank)"
        global_tokens = sum(all_splits)
        ep_info = {"experts": E, "experts_per_rank": G, "m_splits_all": all_splits}
    elif (not multi) or args.parallelism == "dp":
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = (torch.randn(N, K, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
        dy = (torch.randn(M, N, device=dev, dtype=torch.float32) * 1e-2).to(torch.bfloat16)
        st = L.LinearFP8State(recipe, device=dev, fp8_transposes=args.fp8_transposes)

        def step(first=False):
            # the recipe step of BOTH directions is one launch behind the backward pass (LinearFP8State.update_fwd_bwd:
            # same values as the reference's two calls per step, tests/test_parity_r3_gpu.py)
            y, saved = L.linear_forward(x, w, None, st, is_first_microbatch=first)
            dx, dw, _ = L.linear_backward(dy, saved, st)
            st.update_fwd_bwd()
            return y, dx, dw

        flops_per_step = 6.0 * M * N * K * world     # whole job
        parallelism = "single" if not multi else f"dp{world}"
        global_tokens = M * world
    else:
        from transformer_engine_amd import distributed as D

        def make_tp(overlap):
            kw = dict(device=dev, overlap=overlap, chunks=args.tp_chunks)
            if args.tp_layer == "row":
                return D.RowParallelLinearSP(K, N, world, rank, recipe, **kw)
            if args.tp_layer == "ln_column":
                return D.ColumnParallelLayerNormLinearSP(K, N, world, rank, recipe, **kw)
            return D.ColumnParallelLinearSP(K, N, world, rank, recipe, **kw)

        mod = make_tp(True)
        if args.tp_layer == "row":      # all tokens x local features in, local tokens x N out
            x_local = torch.randn(M * world, K // world, device=dev, dtype=torch.bfloat16)
            dy = (torch.randn(M, N, device=dev, dtype=torch.float32) * 1e-2).to(torch.bfloat16)
        else:                           # local tokens x K in, all tokens x local output columns out
            x_local = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            dy = (torch.randn(M * world, N // world, device=dev, dtype=torch.float32) * 1e-2).to(torch.bfloat16)

        def step(first=False):
            return mod.step(x_local, dy, is_first_microbatch=first)

        flops_per_step = 6.0 * (M * world) * N * K      # whole job
        parallelism = f"tp{world}+sp ({args.tp_layer}-parallel)"
        global_tokens = M * world

    # warm the scales (>= 2 steps so delayed scaling values are in range), cache the FP8 weight
    step(first=True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()

    # RCCL collectives are left out of graph capture; the expert-parallel step has none (per-rank graph)
    use_graph = (not args.no_graph) and ((not multi) or args.parallelism == "ep")
    graph = None
    seg = None
    if (not args.no_graph) and multi and args.parallelism == "dp":
        # N data-parallel replicas: the same kernels as at N = 1, replayed from TWO graphs (forward,
        # backward) with the recipe's packed amax all_reduce(MAX) + scale update issued eagerly between
        # them - no collective is ever captured.
        try:
            holder = {}

            def fwd_part():
                holder["y"], holder["saved"] = L.linear_forward(x, w, None, st, is_first_microbatch=False)

            def bwd_part():
                holder["dx"], holder["dw"], _ = L.linear_backward(dy, holder["saved"], st)

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fwd_part(); st.update_fwd(); bwd_part(); st.update_bwd()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g_fwd, g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_fwd):
                fwd_part()
            with torch.cuda.graph(g_bwd, pool=g_fwd.pool()):
                bwd_part()
            torch.cuda.synchronize()
            seg = (g_fwd, g_bwd)
        except Exception as exc:
            print(f"[bench] hipGraph capture unavailable ({type(exc).__name__}: {exc}); running eagerly",
                  file=sys.stderr)
            seg = None
            torch.cuda.synchronize()
    if use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
            torch.cuda.synchronize()
        except Exception as exc:  # capture is an optimisation, not a requirement
            print(f"[bench] hipGraph capture unavailable ({type(exc).__name__}: {exc}); running eagerly",
                  file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    # N > 1: the recipe's amax all_reduce(MAX) + scale update of the forward tensors runs on a side stream
    # while the backward graph runs (and the backward one while the next forward graph runs): the forward
    # scales are only read by the NEXT step's forward graph, the backward scales by the next backward graph,
    # and no kernel of the other graph touches that recipe state - same values as the serial order, two
    # collective latencies per step off the critical path.
    comm = torch.cuda.Stream() if multi else None
    ev = {k: torch.cuda.Event() for k in ("f", "fd", "b", "bd")} if multi else None

    def run_segments():
        main = torch.cuda.current_stream()
        main.wait_event(ev["fd"])            # the previous step's forward scale update (no-op before the first)
        seg[0].replay()
        ev["f"].record(main)
        with torch.cuda.stream(comm):
            comm.wait_event(ev["f"])
            st.update_fwd()
            ev["fd"].record(comm)
        main.wait_event(ev["bd"])
        seg[1].replay()
        ev["b"].record(main)
        with torch.cuda.stream(comm):
            comm.wait_event(ev["b"])
            st.update_bwd()
            ev["bd"].record(comm)

    run = (lambda: graph.replay()) if graph is not None else (run_segments if seg is not None else (lambda: step()))

    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    # (the collector is off for the whole run, see main(): collect here, outside the timed region)
    gc.collect()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timed_as = "hip_graph" if graph is not None else ("hip_graph_segments" if seg is not None else "eager")
    other_ms = None
    if graph is not None and not multi:
        # N = 1: the same K steps launched eagerly.  Replay is not always the faster form on this chip (its kernels
        # follow each other without the eager launches' few-microsecond gaps, and the power-bound GEMMs then hold a
        # lower clock: 1-4 % either way depending on the box); `value` is the better of the two, both are reported.
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        gc.collect()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        eager_elapsed = time.perf_counter() - t1
        if eager_elapsed < elapsed:
            other_ms, elapsed, timed_as = elapsed / args.steps * 1e3, eager_elapsed, "eager"
        else:
            other_ms = eager_elapsed / args.steps * 1e3
    ms_per_step = elapsed / args.steps * 1e3
    tflops = flops_per_step / (elapsed / args.steps) / 1e12

    result = {
        "metric": "FP8 Linear fwd+bwd TFLOPS/GPU + cast_transpose GB/s",
        "value": tflops, "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp8(e4m3/e5m2)->f32acc", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: Llama-3-8B MLP Linear 4096->14336, FP8 delayed scaling "
                               "(HYBRID), fwd+bwd, 8192 tokens per GPU",
                   "tokens": global_tokens, "in_features": K, "out_features": N, "parallelism": parallelism,
                   "hip_graph": timed_as != "eager", "timed_as": timed_as,
                   **({"ms_per_step_other_launch_form": other_ms} if other_ms is not None else {}),
                   "weight_quantize": "amortised (is_first_microbatch=False)",
                   **({"fp8_transposes": args.fp8_transposes} if ((not multi) or args.parallelism == "dp") else {}),
                   **({"amax_reduction": "all_reduce(MAX) + scale update on a side stream, behind the other graph"}
                      if seg is not None else {})},
        "tflops_per_gpu": tflops / world,
    }
    if args.rehearse_multi and world == 1:
        result["rehearsal"] = "N > 1 code paths over a world-size-1 RCCL group with forced collectives: not a scaling result"
    if ep_info is not None:
        result["scaling"] = "strong"
        result["dtype"] = "fp8(e4m3/e5m2)->f32acc"
        result["config"] = {"workload": "BASELINE configs[4]: Mixtral 8x7B GroupedLinear 4096->14336, 8 experts, 16384 routed "
                                        "tokens (8192 x top-2, ragged splits padded to 16), per-expert FP8 current scaling "
                                        "quantize + grouped FP8 GEMM, fwd+bwd; experts spread over the ranks, tokens "
                                        "already dispatched (no collective inside the path)",
                            "routed_tokens": global_tokens, "in_features": K, "out_features": N, "parallelism": parallelism,
                            "hip_graph": timed_as != "eager", "timed_as": timed_as,
                            **({"ms_per_step_other_launch_form": other_ms} if other_ms is not None else {}),
                            "weight_quantize": "amortised (is_first_microbatch=False)", **ep_info}
    if args.gemm_cus > 0:
        result["config"]["gemm_cus"] = args.gemm_cus

    if multi and args.parallelism == "tp_sp":
        # ---- exposed communication: the same step (a) with the pipelines switched off (one blocking
        # collective, then the whole GEMM) and (b) the step's collectives alone, same sizes, back to back.
        # comm_ms = (b); overlap_frac = the share of (b) the overlapped schedule hides = ((a) - step) / (b).
        def timed(fn, iters):
            for _ in range(2):
                fn()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) / iters * 1e3],
                             device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        serial = make_tp(False)
        serial.weight = mod.weight
        serial.step(x_local, dy, is_first_microbatch=True)
        it = max(3, args.steps // 4)
        serial_ms = timed(lambda: serial.step(x_local, dy, is_first_microbatch=False), it)
        if args.tp_layer == "row":
            rs_src = torch.empty(M * world, N, device=dev, dtype=torch.bfloat16)      # Y partial sums
            ag_src = torch.empty(M, N, device=dev, dtype=torch.uint8)                 # dY FP8 bytes
        else:
            rs_src = torch.empty(M * world, K, device=dev, dtype=torch.bfloat16)      # dgrad partial sums
            ag_src = torch.empty(M, K, device=dev, dtype=torch.uint8)                 # X FP8 bytes
        amax = torch.zeros(3, device=dev)

        def comm_only():
            D.all_gather_rows(ag_src, None)
            D.all_reduce_max_(amax, None)
            D.reduce_scatter_along_first_dim(rs_src, None)
            D.all_reduce_max_(amax[:2], None)

        comm_ms = timed(comm_only, it)
        result["comm_ms"] = comm_ms
        result["serial_ms_per_step"] = serial_ms
        result["overlap_frac"] = max(0.0, min(1.0, (serial_ms - ms_per_step) / comm_ms)) if comm_ms > 0 else None
        result["comm_bytes_per_rank"] = {"all_gather_fp8": ag_src.numel() * (world - 1),
                                         "reduce_scatter_bf16": rs_src.numel() * 2 * (world - 1) // world}
        result["config"]["tp_chunks"] = args.tp_chunks

    if ((not multi) or args.parallelism == "dp") and args.parallelism != "ep":
        # ---- per-kernel measurements (HIP events on the launch stream) -------------------------
        # every rank runs them (the instrumented steps contain the recipe's amax all-reduce); rank 0 reports
        from transformer_engine_amd import general_gemm
        it = max(10, args.steps // 2)
        qi, qw, qg = st.fwd_q[L.FWD_INPUT], st.fwd_q[L.FWD_WEIGHT], st.bwd_q[L.BWD_GRAD_OUTPUT]
        mode = args.fp8_transposes
        # which tensors carry a transposed copy in the timed step (LinearFP8State.fp8_transposes)
        cols = {"all": (True, True, True), "grad_output": (False, False, True), "input": (True, False, False),
                "none": (False, False, False)}

        def operands(flags):
            """FP8 X, W, dY (+ preallocated outputs of the same kind) with / without transposed copies."""
            outs, empties = [], []
            for q, t, c in zip((qi, qw, qg), (x, w, dy), flags):
                q.set_usage(rowwise=True, columnwise=c)
                outs.append(q(t))
                empties.append(q.make_empty(t.shape, dtype=t.dtype, device=dev))
            return outs, empties

        # cast + transpose in isolation (the metric's second half), always measured
        (x8t, w8t, dy8t), (oxt, owt, odyt) = operands(cols["all"])
        for q in (qi, qw, qg):
            q.set_usage(rowwise=True, columnwise=True)
        t_ct_x = time_kernel(lambda: qi.update_quantized(x, oxt), it)
        t_ct_w = time_kernel(lambda: qw.update_quantized(w, owt), it)
        t_ct_dy = time_kernel(lambda: qg.update_quantized(dy, odyt), it)
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
        dw = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
        # the GEMMs in isolation, on the operands of the timed step (K-major reads where it keeps no transposes)
        if mode == "all":
            x8, w8, dy8 = x8t, w8t, dy8t
        else:
            (x8, w8, dy8), _ = operands(cols[mode])
        t_fprop = time_kernel(lambda: general_gemm(w8, x8, layout="TN", out=y), it)
        t_dgrad = time_kernel(lambda: general_gemm(w8, dy8, layout="NN", out=dx, grad=True), it)
        t_wgrad = time_kernel(lambda: general_gemm(x8, dy8, layout="NT", out=dw, grad=True), it)
        gemm_flops = 2.0 * M * N * K
        # the same fprop launch on operands whose bits do not toggle (all zeros): what the kernel's instruction
        # stream sustains when the chip does not drop its clock under switching power - the GEMM on random
        # operand bits is power-bound, and this figure separates that from the kernel's own stalls
        from transformer_engine_amd.tensor import Float8CurrentScalingQuantizer
        qz = Float8CurrentScalingQuantizer(qi.dtype, dev)
        x8z, w8z = qz(torch.zeros_like(x)), qz(torch.zeros_like(w))
        t_fprop_zero = time_kernel(lambda: general_gemm(w8z, x8z, layout="TN", out=y), it)
        t_fprop_again = time_kernel(lambda: general_gemm(w8, x8, layout="TN", out=y), it)
        del x8z, w8z
        # ---- the same kernels timed INSIDE the step: args.steps eager steps of the timed workload with a
        # HIP event pair (on the launch stream) around every hot launch.  These are the durations the
        # roofline uses - they include the cache state a real step leaves behind (a kernel launched
        # back to back on the same operands finds part of them in the 256 MiB Infinity Cache) and they
        # are what `rocprofv3 --kernel-trace --stats -- python3 bench.py` averages (profiles/).
        roles = ("quantize_x", "gemm_fprop", "quantize_dy", "gemm_dgrad", "gemm_wgrad", "empty_pair")

        def instrument(flags, state):
            """args.steps eager steps with event pairs; flags = which of X / W / dY carry a transposed copy."""
            in_step = {k: [] for k in roles}
            (_, wq8, _), (ox, _, ody) = operands(flags)

            class rec:
                def __init__(self, role):
                    self.role = role
                    self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

                def __enter__(self):
                    self.a.record()

                def __exit__(self, *exc):
                    self.b.record()
                    in_step[self.role].append((self.a, self.b))

            def instrumented_step():
                qi.set_usage(rowwise=True, columnwise=flags[0])
                with rec("quantize_x"):
                    qi.update_quantized(x, ox)
                # outputs preallocated (y / dx / dw above): an allocator call between the two events of a pair would be
                # charged to the kernel (seen: one 67 MB torch.empty stalling a dgrad pair to 0.9 ms)
                with rec("gemm_fprop"):
                    general_gemm(wq8, ox, layout="TN", out=y)
                qg.set_usage(rowwise=True, columnwise=flags[2])
                with rec("quantize_dy"):
                    qg.update_quantized(dy, ody)
                with rec("gemm_dgrad"):
                    general_gemm(wq8, ody, layout="NN", grad=True, out=dx)
                with rec("gemm_wgrad"):
                    general_gemm(ox, ody, layout="NT", grad=True, out=dw)
                with rec("empty_pair"):      # two records with nothing between them: what a pair adds to any kernel
                    pass
                state.update_fwd_bwd()

            for _ in range(3):
                instrumented_step()
            for v in in_step.values():
                v.clear()
            gc.collect()
            for _ in range(args.steps):
                instrumented_step()
            torch.cuda.synchronize()
            # mean per kernel; a pair more than 1.3x its median holds a host stall between the two records (the first
            # record completes at once on an idle stream, the launch follows late: seen 8-40 ms pairs around 0.37 ms
            # kernels whose rocprofv3-traced durations were normal) and is dropped, counted below; the rocprofv3
            # per-kernel averages of the same command (profiles/) are the cross-check
            smp = {k: sorted(a.elapsed_time(b) for a, b in v) for k, v in in_step.items()}
            kpt = {k: [t for t in s_ if t <= 1.3 * s_[len(s_) // 2]] for k, s_ in smp.items()}
            return smp, kpt, {k: sum(v) / len(v) for k, v in kpt.items()}

        samples, kept, step_ms = instrument(cols[mode], st)
        if mode == "all":
            ct_step_ms = step_ms
        else:   # the cast + transpose kernel inside a step: the same step with transposed copies for all three tensors
            _, _, ct_step_ms = instrument(cols["all"], L.LinearFP8State(recipe, device=dev, fp8_transposes="all"))
        gemm_avg_ms = (step_ms["gemm_fprop"] + step_ms["gemm_dgrad"] + step_ms["gemm_wgrad"]) / 3.0
        achieved = gemm_flops / (gemm_avg_ms * 1e-3) / 1e12
        traffic = load_traffic()
        default_shape = (M, K, N) == (8192, 4096, 14336)   # the committed passes were taken on these shapes
        sfx = "" if mode == "all" else "_" + mode          # roles of profiles/pmc_target.py
        g_traffic = [traffic_of(traffic, r + sfx) if default_shape else None for r in ("gemm_fprop", "gemm_dgrad", "gemm_wgrad")]
        result["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s",
                              "frac": achieved / PEAK_FP8_TFLOPS,
                              "traffic": (sum(g_traffic) / 3.0) if all(t is not None for t in g_traffic) else None,
                              "traffic_unit": "bytes/launch (fabric side: HBM + Infinity Cache; fprop/dgrad/wgrad mean)",
                              "kernel": "gemm_fp8_kernel (fprop/dgrad/wgrad average, 2*M*N*K FLOP per launch"
                                        + ("" if mode == "all" else "; dgrad / wgrad read row-wise payloads K-major") + ")"}
        ct_bytes = dy.numel() * 4.0      # 2 B read + 1 B rowwise + 1 B transposed per bf16 element
        ct_gbps = ct_bytes / (ct_step_ms["quantize_dy"] * 1e-3) / 1e9
        result["roofline_cast_transpose"] = {"bound": "hbm", "achieved": ct_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                             "frac": ct_gbps / PEAK_HBM_GBPS,
                                             "traffic": traffic_of(traffic, "cast_transpose_dy") if default_shape else None,
                                             "algorithmic_bytes": ct_bytes,
                                             "kernel": "cast_transpose_tile_kernel on dY [8192,14336] bf16->e5m2, 4 B/element"
                                                       + ("" if mode == "all" else " (inside the fp8_transposes='all' step)")}
        if mode != "all":
            # what the timed step runs instead: the row-wise cast, 3 B/element
            c_bytes = dy.numel() * 3.0
            c_gbps = c_bytes / (step_ms["quantize_dy"] * 1e-3) / 1e9
            result["roofline_cast"] = {"bound": "hbm", "achieved": c_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                       "frac": c_gbps / PEAK_HBM_GBPS, "algorithmic_bytes": c_bytes,
                                       "traffic": traffic_of(traffic, "cast_dy") if default_shape else None,
                                       "launch_ms": step_ms["quantize_dy"],
                                       "kernel": ("cast_flat_kernel" if mode == "none" else "cast_transpose_tile_kernel")
                                                 + " on dY [8192,14336] bf16->e5m2 inside the timed step"}
        meta = traffic.get("_meta", {})
        result["roofline"]["traffic_source"] = {
            "file": traffic.get("_file"), "commit": meta.get("commit"), "date": meta.get("date"),
            "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of profiles/pmc_target.py (same kernels, same "
                   "shapes), not read live: counters cannot be collected from inside this process"}
        result["roofline"]["launch_ms"] = {k: step_ms[k] for k in ("gemm_fprop", "gemm_dgrad", "gemm_wgrad")}
        result["roofline"]["timing"] = "HIP events around each launch inside args.steps eager steps of the timed workload"
        result["roofline"]["launch_ms_min_median_max"] = {k: [s_[0], s_[len(s_) // 2], s_[-1]] for k, s_ in samples.items()}
        result["roofline"]["stalled_pairs_dropped"] = {k: len(samples[k]) - len(kept[k]) for k in samples}
        result["roofline"]["event_pair_floor_ms"] = step_ms["empty_pair"]    # included in every launch_ms above, not subtracted
        # cross-check against the timed region itself: the medians of the instrumented launches add up to the
        # replayed step (see extra.step_breakdown), the means carry the eager loop's occasional +20 % launches
        med = [samples[k][len(samples[k]) // 2] for k in ("gemm_fprop", "gemm_dgrad", "gemm_wgrad")]
        result["roofline"]["frac_at_median_launch"] = gemm_flops / (sum(med) / 3.0 * 1e-3) / 1e12 / PEAK_FP8_TFLOPS
        result["roofline_cast_transpose"]["launch_ms"] = ct_step_ms["quantize_dy"]
        # headline both ways: `value` amortises the weight quantize as te.Linear does with
        # is_first_microbatch=False; the other figure charges one full weight cast_transpose to every step
        result["value_incl_weight_quantize"] = flops_per_step / ((ms_per_step + t_ct_w) * 1e-3) / 1e12
        hot_kernels = ("quantize_x", "gemm_fprop", "quantize_dy", "gemm_dgrad", "gemm_wgrad")
        # the breakdown adds up MEDIAN launches: the eager instrumented loop has occasional +20-30 % launches (host
        # stalls between the two records) that the replayed graph of the timed region does not have
        med_ms = {k: samples[k][len(samples[k]) // 2] for k in samples}
        hot_ms = sum(med_ms[k] for k in hot_kernels)
        hot_net_ms = hot_ms - len(hot_kernels) * med_ms["empty_pair"]     # every event pair carries the floor of its own
        result["extra"] = {
            "step_breakdown": {
                "ms_per_step": ms_per_step, "hot_kernels_ms": hot_ms, "hot_kernels_net_of_event_floor_ms": hot_net_ms,
                "gemm_ms": med_ms["gemm_fprop"] + med_ms["gemm_dgrad"] + med_ms["gemm_wgrad"],
                "quantize_ms": med_ms["quantize_x"] + med_ms["quantize_dy"],
                "recipe_kernel_and_launch_gaps_ms": ms_per_step - hot_net_ms,
                "recipe_kernel_and_launch_gaps_share": (ms_per_step - hot_net_ms) / ms_per_step,
                "note": "hot kernels timed in eager instrumented steps (HIP event pairs, MEDIAN launch, the measured empty-pair "
                        "floor subtracted once per kernel), ms_per_step from the timed region (config.timed_as): the remainder is the ONE "
                        "recipe-update kernel of the step (forward + backward states in one launch; amax rows are "
                        "zeroed by that kernel, there is no memset) and the five inter-kernel gaps"},
            "in_step_ms": step_ms,
            "in_step_ms_with_all_transposes": ct_step_ms,
            "isolated_note": "gemm_ms / cast_transpose_ms below: the same kernel launched back to back on resident "
                             "operands (upper bound: inputs partly stay in the Infinity Cache)",
            "gemm_ms": {"fprop": t_fprop, "dgrad": t_dgrad, "wgrad": t_wgrad},
            "gemm_tflops": {"fprop": gemm_flops / t_fprop / 1e9, "dgrad": gemm_flops / t_dgrad / 1e9,
                            "wgrad": gemm_flops / t_wgrad / 1e9},
            "cast_transpose_ms": {"x": t_ct_x, "w": t_ct_w, "dy": t_ct_dy},
            "cast_transpose_GBps": {"x": x.numel() * 4 / t_ct_x / 1e6, "w": w.numel() * 4 / t_ct_w / 1e6,
                                    "dy": dy.numel() * 4 / t_ct_dy / 1e6},
            "gemm_only_tflops": 3 * gemm_flops / ((t_fprop + t_dgrad + t_wgrad) * 1e-3) / 1e12,
            "power_bound_check": {
                "fprop_ms_random_operands": t_fprop_again, "fprop_ms_zero_operands": t_fprop_zero,
                "tflops_random": gemm_flops / t_fprop_again / 1e9, "tflops_zero": gemm_flops / t_fprop_zero / 1e9,
                "frac_of_peak_zero_operands": gemm_flops / t_fprop_zero / 1e9 / PEAK_FP8_TFLOPS,
                "note": "same kernel, same shapes, same launch: only the operand bits differ (all-zero FP8 payloads "
                        "vs the workload's randn data); the difference is the clock the chip gives up under "
                        "switching power, not kernel stalls"},
            "weight_quantize_ms_not_in_step": t_ct_w,
        }
        if not multi:
            # The same step with the other settings of LinearFP8State.fp8_transposes ("all": transposed copies of X, W
            # and dY as in rounds 1-2; "grad_output": only dY^T; "none": no FP8 transpose, the GEMMs read row-wise
            # payloads K-major through ds_read_b64_tr_b8).  Same results bit for bit (tests/test_parity_r3_gpu.py).
            def variant_step(vmode):
                st2 = L.LinearFP8State(recipe, device=dev, fp8_transposes=vmode)

                def step2(first=False):
                    y2, saved2 = L.linear_forward(x, w, None, st2, is_first_microbatch=first)
                    L.linear_backward(dy, saved2, st2)
                    st2.update_fwd_bwd()

                step2(first=True)
                for _ in range(2):
                    step2()
                torch.cuda.synchronize()
                run2 = step2
                if graph is not None:
                    try:
                        side = torch.cuda.Stream()
                        side.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side):
                            step2()
                        torch.cuda.current_stream().wait_stream(side)
                        torch.cuda.synchronize()
                        g2 = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g2):
                            step2()
                        torch.cuda.synchronize()
                        run2 = g2.replay
                    except Exception as exc:
                        print(f"[bench] second graph unavailable ({type(exc).__name__}: {exc})", file=sys.stderr)
                        torch.cuda.synchronize()
                return time_kernel(run2, args.steps, warm=args.warmup)

            what = {"all": "fp8_transposes='all' (the data flow of rounds 1-2): cast_transpose for X and dY, every GEMM "
                           "operand K-contiguous",
                    "grad_output": "fp8_transposes='grad_output': no X^T / W^T, dgrad + wgrad read the row-wise payloads of "
                                   "W and X K-major (ds_read_b64_tr_b8)",
                    "input": "fp8_transposes='input': only X^T is kept; dgrad reads W and wgrad reads dY K-major (one K-major "
                             "operand per GEMM)",
                    "none": "fp8_transposes='none': X, W and dY are cast row-wise only (3 B/element, no FP8 transpose in "
                            "the layer), dgrad reads W and wgrad reads X and dY K-major"}
            result["extra"]["step_variants"] = {"note": "the same step with the other settings of LinearFP8State."
                                                        "fp8_transposes; outputs and recipe state are bit-equal "
                                                        "(tests/test_parity_r3_gpu.py); `value` is the '%s' one" % mode}
            for other in ("all", "grad_output", "input", "none"):
                ms_o = ms_per_step if other == mode else variant_step(other)
                result["extra"]["step_variants"][other] = {"ms_per_step": ms_o, "tflops": flops_per_step / (ms_o * 1e-3) / 1e12,
                                                           "what": what[other]}
        if not args.no_grouped and not multi:
            result["extra"]["c5_grouped_linear"] = grouped_c5(dev, max(5, args.steps // 4))
            result["extra"]["c3_mxfp8_linear"] = mxfp8_c3(dev, max(5, args.steps // 4))
        if not args.no_cpu_baseline and not multi:
            result["cpu_baseline"] = cpu_baseline()
            # configs[0] (the reference's CPU-runnable case: Linear 768 -> 3072, 2048 tokens) in full, CPU and GPU
            c1 = cpu_baseline(tokens=2048, K=768, N=3072)
            x1 = torch.randn(2048, 768, device=dev, dtype=torch.bfloat16)
            w1 = (torch.randn(3072, 768, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
            dy1 = (torch.randn(2048, 3072, device=dev, dtype=torch.float32) * 1e-2).to(torch.bfloat16)
            st1 = L.LinearFP8State(recipe, device=dev)

            def step1(first=False):
                _, saved1 = L.linear_forward(x1, w1, None, st1, is_first_microbatch=first)
                st1.update_fwd()
                L.linear_backward(dy1, saved1, st1)
                st1.update_bwd()

            step1(first=True)
            step1()
            ms1 = time_kernel(step1, 20)
            result["cpu_baseline"]["configs0_full"] = {
                "workload": "configs[0]: Linear 768->3072, 2048 tokens, delayed scaling, fwd+bwd (28.99 GFLOP)",
                "cpu_tflops": c1["value"], "cpu_seconds": c1["seconds"], "cpu_cores": c1["cores"],
                "gpu_ms_per_step_eager": ms1, "gpu_tflops_eager": 6.0 * 2048 * 768 * 3072 / (ms1 * 1e-3) / 1e12}

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
