import torch
from transformer_engine_amd.gemm import general_gemm
from transformer_engine_amd.tensor import DType, Float8Quantizer
dev="cuda"
def q8(rows, cols):
    x = torch.randn(rows, cols, device=dev, dtype=torch.bfloat16)
    q = Float8Quantizer(torch.ones(1, device=dev), torch.zeros(1, device=dev), DType.kFloat8E4M3, columnwise=False)
    return q(x)
def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for n, m, k in [(8192, 14336, 4096), (16384, 14336, 4096), (16384, 4096, 14336), (14336, 4096, 16384)]:
    A, B = q8(m, k), q8(n, k)
    out = torch.empty(n, m, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: general_gemm(A, B, out=out, layout="TN"))
    print(f"n={n} m={m} k={k}: {t:7.1f} us {2*n*m*k/t/1e6:6.0f} TF  out {n*m*2/1e6:.0f} MB")
