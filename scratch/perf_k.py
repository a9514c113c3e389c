import sys; sys.path.insert(0, '.')
import os, torch
from transformer_engine_amd.tensor import *
from transformer_engine_amd.gemm import general_gemm
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
def q8(x, dt=DType.kFloat8E4M3):
    q = Float8Quantizer(torch.full((1,), 64.0, device="cuda"), torch.zeros(1, device="cuda"), dt)
    return q(x)
res = []
for (n, m, k) in [(8192, 14336, 4096), (8192, 14336, 4096 + 128), (8192, 14336, 4096 - 128), (8192, 14336, 4096 + 256), (8192, 14336, 8192), (8192, 14336, 8192 + 128), (8192, 4096, 14336), (8192, 4096, 14336 + 128)]:
    b = q8(torch.randn(n, k, device="cuda", dtype=torch.bfloat16))
    a = q8(torch.randn(m, k, device="cuda", dtype=torch.bfloat16))
    t = timeit(lambda: general_gemm(a, b, out_dtype=torch.bfloat16, layout="TN"))
    res.append(f"[{n}x{m}x{k}] {t*1e3:.0f} us {2*n*m*k/t/1e9:.0f} TF")
print("DBG", os.environ.get("NVTE_AMD_GEMM_DBG", "0"), " | ".join(res))
