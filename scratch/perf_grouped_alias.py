"""Is the C5 grouped fprop slowed by the 8 x 58 MB weight stream?  A discrete A list of 8 DISTINCT weights vs the SAME
weight 8 times (then the weights sit in the Infinity Cache after the first member)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, bench
from transformer_engine_amd.grouped import general_grouped_gemm_for_grouped_tensor, group_quantize, GroupedTensor, splits_to_offsets
from transformer_engine_amd.tensor import DType, Float8Quantizer, Float8CurrentScalingQuantizer
dev = torch.device("cuda:0")
G, K, N, routed = 8, 4096, 14336, 16384
counts = np.bincount(np.random.default_rng(1234).integers(0, G, size=routed), minlength=G)
m_splits = [int((c + 15) // 16 * 16) for c in counts]
M = sum(m_splits)
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
fd = torch.tensor(m_splits, dtype=torch.int64, device=dev)
qx = Float8CurrentScalingQuantizer(DType.kFloat8E4M3, dev, columnwise=False)
x8 = group_quantize(x, qx, G, first_dims=fd)
def q8(t):
    return Float8Quantizer(torch.full((1,), 16.0, device=dev), torch.zeros(1, device=dev), DType.kFloat8E4M3, columnwise=False)(t)
ws = [q8(torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02) for _ in range(G)]
out = GroupedTensor(num_tensors=G, logical_shape=(M, N), dtype=torch.bfloat16,
                    rowwise_data=torch.empty(M * N, dtype=torch.bfloat16, device=dev), first_dims=fd,
                    tensor_offsets=splits_to_offsets(fd, N))
for rnd in range(2):
    for name, lst in (("distinct", ws), ("aliased", [ws[0]] * G)):
        f = lambda: general_grouped_gemm_for_grouped_tensor(lst, x8, out, layout="TN")
        for _ in range(3): f()
        ms = bench.time_kernel(f, 20)
        print(name, round(ms * 1e3, 1), "us", round(2.0 * M * N * K / ms / 1e9, 1), "TFLOP/s", flush=True)
