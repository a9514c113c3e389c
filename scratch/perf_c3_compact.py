import sys, os; sys.path.insert(0, '.')
import torch, bench
from transformer_engine_amd import linear as L, gemm as G
from transformer_engine_amd.recipe import MXFP8BlockScaling
dev = torch.device("cuda:0")
M = K = N = 8192
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
dy = (torch.randn(M, N, device=dev) * 0.01).to(torch.bfloat16)
for rnd in range(2):
    for first in (1 << 37, 1 << 62):
        G._SWIZZLE_FIRST_MACS = first
        for compact in ((), ("input", "weight", "grad_output"), ("input",)):
            st = L.LinearFP8State(MXFP8BlockScaling(), device=dev, compact_mx_scales=compact)
            L.linear_forward(x, w, None, st, is_first_microbatch=True)
            def step():
                y, saved = L.linear_forward(x, w, None, st, is_first_microbatch=False)
                L.linear_backward(dy, saved, st)
            for _ in range(5): step()
            ms = bench.time_kernel(step, 20)
            print("swizzle-first" if first == 1 << 37 else "direct       ", compact, round(ms, 3), flush=True)
