import sys, os; sys.path.insert(0, '.')
import numpy as np, torch, bench
from transformer_engine_amd.grouped import group_quantize
from transformer_engine_amd.tensor import DType, Float8CurrentScalingQuantizer
dev = torch.device("cuda:0")
G, K, N, routed = 8, 4096, 14336, 16384
counts = np.bincount(np.random.default_rng(1234).integers(0, G, size=routed), minlength=G)
m_splits = [int((c + 15) // 16 * 16) for c in counts]
M = sum(m_splits)
fd = torch.tensor(m_splits, dtype=torch.int64, device=dev)
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
qc = Float8CurrentScalingQuantizer(DType.kFloat8E4M3, dev, columnwise=False)
for nm, t_ in (("X", x), ("dY", dy)):
    t = bench.time_kernel(lambda: group_quantize(t_, qc, G, first_dims=fd), 30)
    print(f"{os.environ.get('TE_AMD_LIBRARY','main')[-40:]} cur {nm:2s} {t*1e3:7.1f} us (zero+amax+scale+cast, 5 B/elem) {t_.numel()*5/t/1e6:6.0f} GB/s", flush=True)
