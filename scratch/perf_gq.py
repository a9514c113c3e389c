"""Grouped quantize rates on the C5-shaped operands: MXFP8 (row + transposed columnwise, swizzled) and FP8 current
scaling (amax + scale + cast + transposes), 8 ragged members."""
import sys, os; sys.path.insert(0, '.')
import numpy as np, torch, bench
from transformer_engine_amd.grouped import group_quantize
from transformer_engine_amd.tensor import DType, MXFP8Quantizer, Float8CurrentScalingQuantizer
dev = torch.device("cuda:0")
print("per_cu", os.environ.get("NVTE_AMD_GROUP_WGS_PER_CU", "default"), flush=True)
G, K, N, routed = 8, 4096, 14336, 16384
counts = np.bincount(np.random.default_rng(1234).integers(0, G, size=routed), minlength=G)
for pad in (128, 16):
    m_splits = [int((c + pad - 1) // pad * pad) for c in counts]
    M = sum(m_splits)
    fd = torch.tensor(m_splits, dtype=torch.int64, device=dev)
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    if pad == 128:
        q = MXFP8Quantizer(DType.kFloat8E4M3, columnwise_transposed=True); q.optimize_for_gemm = True
        for nm, t_ in (("X", x), ("dY", dy)):
            t = bench.time_kernel(lambda: group_quantize(t_, q, G, first_dims=fd), 30)
            print(f"mx   pad{pad} {nm:2s} {t*1e3:7.1f} us {t_.numel()*4.0625/t/1e6:6.0f} GB/s", flush=True)
    qc = Float8CurrentScalingQuantizer(DType.kFloat8E4M3, dev)
    for nm, t_ in (("X", x), ("dY", dy)):
        t = bench.time_kernel(lambda: group_quantize(t_, qc, G, first_dims=fd), 30)
        print(f"cur  pad{pad} {nm:2s} {t*1e3:7.1f} us (amax+scale+cast, 6 B/elem) {t_.numel()*6/t/1e6:6.0f} GB/s", flush=True)
