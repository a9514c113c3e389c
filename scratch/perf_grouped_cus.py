"""C5 grouped GEMMs by workgroup cap (sm_count): does balancing the rounds beat a short tail round?"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, bench
from transformer_engine_amd.grouped import general_grouped_gemm_for_grouped_tensor, group_quantize, GroupedTensor, splits_to_offsets
from transformer_engine_amd.tensor import DType, Float8CurrentScalingQuantizer
dev = torch.device("cuda:0")
G, K, N = 8, 4096, 14336
counts = np.bincount(np.random.default_rng(1234).integers(0, G, size=16384), minlength=G)
ms_ = [int((c + 15) // 16 * 16) for c in counts]
M = sum(ms_)
w = torch.randn(G * N, K, device=dev, dtype=torch.bfloat16) * 0.02
w8 = group_quantize(w, Float8CurrentScalingQuantizer(DType.kFloat8E4M3, dev, columnwise=False), G)
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
fd = torch.tensor(ms_, dtype=torch.int64, device=dev)
x8 = group_quantize(x, Float8CurrentScalingQuantizer(DType.kFloat8E4M3, dev, columnwise=False), G, first_dims=fd)
dy8 = group_quantize(dy, Float8CurrentScalingQuantizer(DType.kFloat8E5M2, dev, columnwise=False), G, first_dims=fd)
def hp(rows, cols):
    return GroupedTensor(num_tensors=G, logical_shape=(rows, cols), dtype=torch.bfloat16,
                         rowwise_data=torch.empty(rows * cols, dtype=torch.bfloat16, device=dev), first_dims=fd,
                         tensor_offsets=splits_to_offsets(fd, cols))
y, dx = hp(M, N), hp(M, K)
dws = [torch.empty(N, K, device=dev, dtype=torch.bfloat16) for _ in range(G)]
for rnd in range(2):
    for cus in (0, 248, 240, 232, 224, 216, 208, 192):
        fns = {"fprop": lambda: general_grouped_gemm_for_grouped_tensor(w8, x8, y, layout="TN", sm_count=cus),
               "dgrad": lambda: general_grouped_gemm_for_grouped_tensor(w8, dy8, dx, layout="NN", sm_count=cus),
               "wgrad": lambda: general_grouped_gemm_for_grouped_tensor(x8, dy8, dws, layout="NT", sm_count=cus)}
        res = {}
        for k, f in fns.items():
            for _ in range(2): f()
            res[k] = round(bench.time_kernel(f, 10) * 1e3, 1)
        print(cus, res, flush=True)
