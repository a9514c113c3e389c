import sys; sys.path.insert(0, '.')
import torch, bench
from transformer_engine_amd import general_gemm
from transformer_engine_amd.tensor import DType, MXFP8Quantizer
dev = torch.device("cuda:0")
M = K = N = 8192
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
dy = (torch.randn(M, N, device=dev) * 0.01).to(torch.bfloat16)
import itertools
for swz, tr in itertools.product((False, True), (True, False)):
    print(f"--- scales {'swizzled' if swz else 'compact'}, columnwise {'transposed (amd_ext)' if tr else 'reference (K-major reads)'}", flush=True)
    def mk(dt):
        q = MXFP8Quantizer(dt, columnwise_transposed=tr); q.optimize_for_gemm = swz; return q
    qx, qw, qg = mk(DType.kFloat8E4M3), mk(DType.kFloat8E4M3), mk(DType.kFloat8E5M2)
    for nm, q, t_ in (("qx", qx, x), ("qdy", qg, dy)):
        t = bench.time_kernel(lambda: q(t_), 20)
        print(f"swz={swz} quantize {nm}: {t*1e3:.1f} us", flush=True)
    x8, w8, dy8 = qx(x), qw(w), qg(dy)
    for nm, fn in (("fprop", lambda: general_gemm(w8, x8, out_dtype=torch.bfloat16, layout="TN")),
                   ("dgrad", lambda: general_gemm(w8, dy8, out_dtype=torch.bfloat16, layout="NN", grad=True)),
                   ("wgrad", lambda: general_gemm(x8, dy8, out_dtype=torch.bfloat16, layout="NT", grad=True))):
        t = bench.time_kernel(fn, 20)
        print(f"swz={swz} {nm}: {t*1e3:.1f} us = {2.0*M*N*K/t/1e9:.0f} TF", flush=True)
