"""C5-shaped GroupedLinear step under MXFP8 (members padded to 128 rows) by columnwise layout."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, bench
from transformer_engine_amd.grouped_linear import GroupedLinearFP8State, grouped_linear_backward, grouped_linear_forward
from transformer_engine_amd.recipe import MXFP8BlockScaling
dev = torch.device("cuda:0")
G, K, N, routed = 8, 4096, 14336, 16384
counts = np.bincount(np.random.default_rng(1234).integers(0, G, size=routed), minlength=G)
m_splits = [int((c + 127) // 128 * 128) for c in counts]
M = sum(m_splits)
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
w = torch.randn(G * N, K, device=dev, dtype=torch.bfloat16) * 0.02
dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
fd = torch.tensor(m_splits, dtype=torch.int64, device=dev)
for rnd in range(2):
    for layout in ("transposed", "reference"):
        st = GroupedLinearFP8State(MXFP8BlockScaling(), G, dev, mx_columnwise=layout)
        grads = [torch.zeros(N, K, device=dev, dtype=torch.bfloat16) for _ in range(G)]
        grouped_linear_forward(x, w, fd, st, is_first_microbatch=True)
        def step():
            y, saved = grouped_linear_forward(x, w, fd, st, is_first_microbatch=False)
            grouped_linear_backward(dy, saved, fd, st, wgrad_out=grads, accumulate_wgrad=False)
        for _ in range(3): step()
        ms = bench.time_kernel(step, 20)
        print(layout, round(ms, 3), "ms", round(6.0 * M * N * K / ms / 1e9, 1), "TFLOP/s", flush=True)
