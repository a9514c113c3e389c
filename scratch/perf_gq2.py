import sys; sys.path.insert(0, '.')
import torch, bench
from transformer_engine_amd.grouped import group_quantize
from transformer_engine_amd.tensor import DType, MXFP8Quantizer
dev = torch.device("cuda:0")
M, N = 16384, 14336
dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
q = MXFP8Quantizer(DType.kFloat8E4M3, columnwise_transposed=True); q.optimize_for_gemm = True
fd = torch.tensor([2048] * 8, dtype=torch.int64, device=dev)
for rep in range(2):
    for nm, fn in (("G=8 uniform", lambda: group_quantize(dy, q, 8)), ("G=8 first_dims", lambda: group_quantize(dy, q, 8, first_dims=fd)),
                   ("G=4 uniform", lambda: group_quantize(dy, q, 4)), ("G=2 uniform", lambda: group_quantize(dy, q, 2))):
        t = bench.time_kernel(fn, 10)
        print(f"{nm}: {t*1e3:.1f} us", flush=True)
