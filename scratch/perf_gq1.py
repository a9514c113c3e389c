import sys; sys.path.insert(0, '.')
import torch, bench
from transformer_engine_amd.grouped import group_quantize
from transformer_engine_amd.tensor import DType, MXFP8Quantizer
dev = torch.device("cuda:0")
M, N = 16384, 14336
dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
for swz in (True, False):
    for tcol in (True, False):
        q = MXFP8Quantizer(DType.kFloat8E4M3, columnwise_transposed=tcol); q.optimize_for_gemm = swz
        out = q.make_empty(dy.shape, dtype=dy.dtype, device=dev)
        t0 = bench.time_kernel(lambda: q.update_quantized(dy, out), 20)
        res = {}
        for G in (1, 8):
            res[G] = bench.time_kernel(lambda: group_quantize(dy, q, G), 20)
        fd = torch.tensor([2048] * 8, dtype=torch.int64, device=dev)
        tv = bench.time_kernel(lambda: group_quantize(dy, q, 8, first_dims=fd), 20)
        print(f"swz={swz} tcol={tcol}: dense {t0*1e3:.1f} us, grouped G=1 {res[1]*1e3:.1f}, G=8 uniform {res[8]*1e3:.1f}, G=8 first_dims {tv*1e3:.1f}", flush=True)
