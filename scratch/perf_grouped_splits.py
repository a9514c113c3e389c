"""Grouped fprop / dgrad / wgrad (C5 shape, grouped tensors, per-tensor current scaling, no transposes) by split pattern."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, bench
from transformer_engine_amd.grouped import general_grouped_gemm_for_grouped_tensor, group_quantize, GroupedTensor, splits_to_offsets
from transformer_engine_amd.tensor import DType, Float8CurrentScalingQuantizer
dev = torch.device("cuda:0")
G, K, N = 8, 4096, 14336
counts = np.bincount(np.random.default_rng(1234).integers(0, G, size=16384), minlength=G)
cases = {"one16384": [16384] + [0] * 7, "two8192_first": [8192, 8192] + [0] * 6, "two8192_spread": [8192, 0, 0, 0, 8192, 0, 0, 0], "four4096": [4096] * 4 + [0] * 4, "c5": [int((c + 15) // 16 * 16) for c in counts], "uniform2048": [2048] * 8, "uniform2064": [2064] * 8,
         }
w = torch.randn(G * N, K, device=dev, dtype=torch.bfloat16) * 0.02
w8 = group_quantize(w, Float8CurrentScalingQuantizer(DType.kFloat8E4M3, dev, columnwise=False), G)
def hp(fd, rows, cols):
    return GroupedTensor(num_tensors=G, logical_shape=(rows, cols), dtype=torch.bfloat16,
                         rowwise_data=torch.empty(rows * cols, dtype=torch.bfloat16, device=dev), first_dims=fd,
                         tensor_offsets=splits_to_offsets(fd, cols))
from transformer_engine_amd.gemm import general_gemm
from transformer_engine_amd.tensor import Float8Quantizer
def q8(t, dt=DType.kFloat8E4M3):
    return Float8CurrentScalingQuantizer(dt, dev, columnwise=False)(t)
for Md in (8192, 16384):
    xd, wd = q8(torch.randn(Md, K, device=dev, dtype=torch.bfloat16)), q8(torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02)
    dyd = q8(torch.randn(Md, N, device=dev, dtype=torch.bfloat16), DType.kFloat8E5M2)
    yd, dxd, dwd = (torch.empty(Md, N, device=dev, dtype=torch.bfloat16), torch.empty(Md, K, device=dev, dtype=torch.bfloat16),
                    torch.empty(N, K, device=dev, dtype=torch.bfloat16))
    fns = {"fprop": lambda: general_gemm(wd, xd, layout="TN", out=yd), "dgrad": lambda: general_gemm(wd, dyd, layout="NN", out=dxd),
           "wgrad": lambda: general_gemm(xd, dyd, layout="NT", out=dwd)}
    res = {}
    for k, f in fns.items():
        for _ in range(3): f()
        ms = bench.time_kernel(f, 15)
        res[k] = (round(ms * 1e3, 1), round(2.0 * Md * N * K / ms / 1e9))
    print("dense", Md, res, flush=True)
for name, ms_ in cases.items():
    M = sum(ms_)
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    fd = torch.tensor(ms_, dtype=torch.int64, device=dev)
    x8 = group_quantize(x, Float8CurrentScalingQuantizer(DType.kFloat8E4M3, dev, columnwise=False), G, first_dims=fd)
    dy8 = group_quantize(dy, Float8CurrentScalingQuantizer(DType.kFloat8E5M2, dev, columnwise=False), G, first_dims=fd)
    y, dx = hp(fd, M, N), hp(fd, M, K)
    dws = [torch.empty(N, K, device=dev, dtype=torch.bfloat16) for _ in range(G)]
    fns = {"fprop": lambda: general_grouped_gemm_for_grouped_tensor(w8, x8, y, layout="TN"),
           "dgrad": lambda: general_grouped_gemm_for_grouped_tensor(w8, dy8, dx, layout="NN"),
           "wgrad": lambda: general_grouped_gemm_for_grouped_tensor(x8, dy8, dws, layout="NT")}
    if name not in ("c5", "uniform2048"): fns.pop("wgrad")
    res = {}
    for k, f in fns.items():
        for _ in range(3): f()
        ms = bench.time_kernel(f, 15)
        res[k] = (round(ms * 1e3, 1), round(2.0 * M * N * K / ms / 1e9))
    print(name, M, res, flush=True)
